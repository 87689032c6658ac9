// thx_fft8.h -- hand-written FFT passes of the gridding-weight iteration (Reconstructor::reconstruct's balancing loop,
// src/Reconstructor.cpp:1379-1551) for power-of-two grids of 64 ... 1024 points per axis (R 8^NS, R = 1, 2, 4;
// 2048 = 4 x 8^3 is instantiable but not enabled: untested).
//
// Why: rocFFT's strided passes over the [P][P][P/2+1] grid run at 2.25 TB/s (0.48 ms per pass at P = 512); a pass that
// stages 512 points x TX adjacent columns in LDS and does the three radix-8 stages there runs at 4.3 TB/s (0.25 ms,
// tools/fft_pass_probe.hip).  Owning the passes also lets the loop's elementwise steps ride along for free:
//   k_fft_x_conv    inverse x transform -> x 1/size x tabulated kernel / nf (convoluteC's real-space stage) -> forward x
//                   transform, one read and one write of C instead of five sweeps over C and the real grid;
//   k_fft_z_update  forward z transform -> W /= max(|C|, 1e-6), checkC maximum, C = T W -> inverse z transform of the
//                   next round, one read and one write of C instead of three.
// Per round: 4 sweeps of C (0.54 GB each way) + W, T once, against 12 sweeps before.
//
// Decomposition (N = 8^3; N = 8^2 drops the middle stage, R = 2, 4 add a radix-R pre-stage, see fftN): n = 64 n2 + 8 n1 + n0, k = k0 + 8 k1 + 64 k2, w = exp(DIR 2 pi i / N):
//   A[k0; n1, n0]   = sum_n2 w8^(n2 k0) x[n2, n1, n0]                      thread t = 8 n1 + n0 holds x[t + 64 n2]
//   B[k0, k1; n0]   = sum_n1 w8^(n1 k1) w^(8 n1 k0) A[k0; n1, n0]          thread t = n0 + 8 k0
//   X[k0, k1, k2]   = sum_n0 w8^(n0 k2) w^(n0 (k0 + 8 k1)) B[k0, k1; n0]   thread t = k0 + 8 k1 ends with X[t + 64 k2]
// so a thread ends a transform holding exactly the elements it would load to start the next one: forward and inverse
// transforms chain through registers.  LDS tile: element-major, one padding row per 64 elements (keeps the stride-64
// reads of the last stage off a single bank group).
#pragma once
#include "thx_common.h"

namespace thx {

__device__ __forceinline__ float2 f8_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 f8_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
template <int DIR> __device__ __forceinline__ float2 f8_mul_i(float2 a) { return DIR > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }
template <int DIR> __device__ __forceinline__ float2 f8_tw(float2 w) { return DIR > 0 ? make_float2(w.x, -w.y) : w; }  // table holds exp(-i)

// y[k] = sum_n v[n] exp(DIR 2 pi i n k / 8), in place, natural order (decimation in frequency)
template <int DIR>
__device__ __forceinline__ void fft8(float2 v[8])
{
    const float h = 0.70710678118654752f;
    const float2 s0 = f8_add(v[0], v[4]), s1 = f8_add(v[1], v[5]), s2 = f8_add(v[2], v[6]), s3 = f8_add(v[3], v[7]);
    float2 d0 = f8_sub(v[0], v[4]), d1 = f8_sub(v[1], v[5]), d2 = f8_sub(v[2], v[6]), d3 = f8_sub(v[3], v[7]);
    {
        const float2 t1 = f8_mul_i<DIR>(d1);
        d1 = make_float2((d1.x + t1.x) * h, (d1.y + t1.y) * h);   // d1 (1 + DIR i) / sqrt 2
        d2 = f8_mul_i<DIR>(d2);
        const float2 t3 = f8_mul_i<DIR>(d3);
        d3 = make_float2((t3.x - d3.x) * h, (t3.y - d3.y) * h);   // d3 (-1 + DIR i) / sqrt 2
    }
    {
        const float2 e0 = f8_add(s0, s2), e1 = f8_sub(s0, s2), o0 = f8_add(s1, s3), o1 = f8_mul_i<DIR>(f8_sub(s1, s3));
        v[0] = f8_add(e0, o0); v[4] = f8_sub(e0, o0); v[2] = f8_add(e1, o1); v[6] = f8_sub(e1, o1);
    }
    {
        const float2 e0 = f8_add(d0, d2), e1 = f8_sub(d0, d2), o0 = f8_add(d1, d3), o1 = f8_mul_i<DIR>(f8_sub(d1, d3));
        v[1] = f8_add(e0, o0); v[5] = f8_sub(e0, o0); v[3] = f8_add(e1, o1); v[7] = f8_sub(e1, o1);
    }
}

// LDS slot of element e of column c: element-major with `pitch` slots per element, one padding row per 64 elements
__device__ __forceinline__ int f8_slot(int e, int c, int pitch) { return (e + (e >> 6)) * pitch + c; }
// Grid sizes: N = R * 8^NS with R in {1, 2, 4}: 64, 128, 256 (NS = 2) and 512, 1024, 2048 (NS = 3)
template <int NS> __host__ __device__ constexpr int f8_m() { return NS == 3 ? 512 : 64; }
template <int NS, int R> __host__ __device__ constexpr int f8_n() { return R * f8_m<NS>(); }
template <int NS, int R> __host__ __device__ constexpr int f8_rows() { return f8_n<NS, R>() + (f8_n<NS, R>() >> 6); }

// The radix-8 stages of an M = 8^NS point transform whose first butterfly's operands are in registers:
// in v[n] = x[t + (M/8) n], out v[k] = X[t + (M/8) k], t < M/8.  The sequence lives at elements [eoff, eoff + M) of the
// tile s; sTw holds exp(-2 pi i m / (TWS M)), so w_M^j = sTw[TWS j].  All threads of the workgroup must call it.
template <int NS, int DIR, int TWS>
__device__ __forceinline__ void fft8n(float2 v[8], int t, int c, int pitch, float2* s, const float2* sTw, int eoff)
{
    fft8<DIR>(v);
    if (NS == 3) {
        {
            const int n1 = t >> 3;
#pragma unroll
            for (int k0 = 0; k0 < 8; k0++)
                s[f8_slot(eoff + k0 * 64 + t, c, pitch)] = cmul(v[k0], f8_tw<DIR>(sTw[((8 * n1 * k0) & 511) * TWS]));
        }
        __syncthreads();
        {
            const int n0 = t & 7, k0 = t >> 3;
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) v[n1] = s[f8_slot(eoff + k0 * 64 + n1 * 8 + n0, c, pitch)];
            fft8<DIR>(v);
#pragma unroll
            for (int k1 = 0; k1 < 8; k1++)
                s[f8_slot(eoff + k0 * 64 + k1 * 8 + n0, c, pitch)] = cmul(v[k1], f8_tw<DIR>(sTw[(n0 * (k0 + 8 * k1)) * TWS]));
        }
        __syncthreads();
        {
            const int k0 = t & 7, k1 = t >> 3;
#pragma unroll
            for (int n0 = 0; n0 < 8; n0++) v[n0] = s[f8_slot(eoff + k0 * 64 + k1 * 8 + n0, c, pitch)];
            fft8<DIR>(v);
        }
    } else {   // M = 64: t = n0, then t = k0
#pragma unroll
        for (int k0 = 0; k0 < 8; k0++) s[f8_slot(eoff + k0 * 8 + t, c, pitch)] = cmul(v[k0], f8_tw<DIR>(sTw[(t * k0) * TWS]));
        __syncthreads();
#pragma unroll
        for (int n0 = 0; n0 < 8; n0++) v[n0] = s[f8_slot(eoff + t * 8 + n0, c, pitch)];
        fft8<DIR>(v);
    }
    __syncthreads();   // the tile may be rewritten by the caller (next transform, staging for the store)
}

// Element held in register q of thread tau (tau < N/8) BEFORE a transform, and in register k AFTER it.
//   N = R M:  X[k0 + R k1] = sum_b w_M^(b k1) [ w_N^(b k0) sum_a w_R^(a k0) x[M a + b] ],  a, k0 < R,  b, k1 < M
// before: q = a + R i  ->  element M a + (tau + (N/8) i)          (the R-point butterflies of 8/R values of b)
// after : thread tau = (k0, t), k0 = tau / (M/8)  ->  element k0 + R (t + (M/8) k)
// For R = 1 both are tau + (N/8) q: transforms chain through registers; otherwise through one pass over the tile.
template <int NS, int R> __device__ __forceinline__ int f8_epre(int tau, int q)
{
    return f8_m<NS>() * (q % R) + tau + (f8_n<NS, R>() / 8) * (q / R);
}
template <int NS, int R> __device__ __forceinline__ int f8_eout(int tau, int k)
{
    constexpr int M8 = f8_m<NS>() / 8;
    return (tau / M8) + R * ((tau % M8) + M8 * k);
}

// N = R 8^NS point transform: in v[q] = x[f8_epre(tau, q)], out v[k] = X[f8_eout(tau, k)].  s: tile of f8_rows<NS, R>() rows,
// sTw: exp(-2 pi i m / N), m < N.
template <int NS, int R, int DIR>
__device__ __forceinline__ void fftN(float2 v[8], int tau, int c, int pitch, float2* s, const float2* sTw)
{
    constexpr int M = f8_m<NS>(), M8 = M / 8, T = f8_n<NS, R>() / 8;
    if (R == 1) {
        fft8n<NS, DIR, 1>(v, tau, c, pitch, s, sTw, 0);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8 / R; i++) {
        float2* u = v + R * i;
        if (R == 2) {
            const float2 a0 = u[0], a1 = u[1];
            u[0] = f8_add(a0, a1); u[1] = f8_sub(a0, a1);
        } else {
            const float2 e0 = f8_add(u[0], u[2]), e1 = f8_sub(u[0], u[2]), o0 = f8_add(u[1], u[3]), o1 = f8_mul_i<DIR>(f8_sub(u[1], u[3]));
            u[0] = f8_add(e0, o0); u[2] = f8_sub(e0, o0); u[1] = f8_add(e1, o1); u[3] = f8_sub(e1, o1);
        }
        const int b = tau + T * i;
#pragma unroll
        for (int k0 = 0; k0 < R; k0++) s[f8_slot(k0 * M + b, c, pitch)] = cmul(u[k0], f8_tw<DIR>(sTw[b * k0]));
    }
    __syncthreads();
    const int k0 = tau / M8, t = tau % M8;
#pragma unroll
    for (int n = 0; n < 8; n++) v[n] = s[f8_slot(k0 * M + t + M8 * n, c, pitch)];
    // the first stage below writes exactly the slots this thread has just read: no barrier needed in between
    fft8n<NS, DIR, R>(v, t, c, pitch, s, sTw, k0 * M);
}

// registers after one transform -> registers before the next one (a no-op for R = 1)
template <int NS, int R>
__device__ __forceinline__ void f8_rearrange(float2 v[8], int tau, int c, int pitch, float2* s)
{
    if (R == 1) return;
#pragma unroll
    for (int k = 0; k < 8; k++) s[f8_slot(f8_eout<NS, R>(tau, k), c, pitch)] = v[k];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = s[f8_slot(f8_epre<NS, R>(tau, q), c, pitch)];
    __syncthreads();
}

// Workgroups are dealt to the 8 XCDs round-robin in launch order, and each XCD has its own L2.  Adjacent x tiles share
// 128-byte lines (64-byte segments; rows of 257 / 264 values are not line-aligned), so the linear workgroup id is
// remapped to give every XCD a CONTIGUOUS run of tiles: a line is then fetched into one L2 instead of two.
__device__ __forceinline__ void f8_xcd_tile(int& bx, int& by)
{
    const unsigned nx = gridDim.x, nT = gridDim.x * gridDim.y;
    unsigned L = blockIdx.x + nx * blockIdx.y;
    if ((nT & 7) == 0) L = (L & 7) * (nT >> 3) + (L >> 3);
    bx = (int)(L % nx);
    by = (int)(L / nx);
}

// ---------------------------------------------------------------------------------------------
// Strided pass, in place: data[b strideB + e strideE + x], e < N, x = blockIdx.x TX + c < nx.  grid (ceil(nx / TX), nBatch).
// ---------------------------------------------------------------------------------------------
template <int NS, int R, int TX, int DIR>
__global__ __launch_bounds__((f8_n<NS, R>() / 8) * TX) void k_fft_strided(float2* __restrict__ data, long strideE, long strideB,
                                                                        int nx, const float2* __restrict__ tw,
                                                                        const int* __restrict__ stop)
{
    constexpr int N = f8_n<NS, R>(), NT8 = N / 8;
    if (stop && *stop) return;   // the gridding loop's stop rule has fired (k_reco_stop_rule): the queued rounds fall through
    extern __shared__ float2 f8_lds[];
    float2* sTw = f8_lds + f8_rows<NS, R>() * TX;
    const int c = threadIdx.x % TX, t = threadIdx.x / TX;
    int bx, by;
    f8_xcd_tile(bx, by);
    const int x = bx * TX + c;
    const bool ok = x < nx;
    float2* base = data + (long)by * strideB + x;
    for (int i = threadIdx.x; i < N; i += NT8 * TX) sTw[i] = tw[i];
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = ok ? base[(long)f8_epre<NS, R>(t, q) * strideE] : make_float2(0.f, 0.f);
    __syncthreads();
    fftN<NS, R, DIR>(v, t, c, TX, f8_lds, sTw);
    if (ok) {
#pragma unroll
        for (int k = 0; k < 8; k++) base[(long)f8_eout<NS, R>(t, k) * strideE] = v[k];
    }
}

// ---------------------------------------------------------------------------------------------
// x direction, fused: per row of the half-complex grid (P/2+1 values, Hermitian), inverse transform -> real row ->
// x 1/size x kernel(|x|^2 / (N pf)^2) / nf (k_convolute_rl's arithmetic, thx_reco.hip) -> forward transform -> same row.
// One workgroup = one canonical pair 0 <= j <= k <= P/2 and the up to 8 rows (+-j, +-k) / (+-k, +-j) that share the
// P/2+1 tabulated kernel values (looked up once, kept in LDS).  Two real rows a, b ride in ONE complex transform:
// Z = A + i B (A, B Hermitian-extended) -> z = a + i b -> multiply -> Z' -> A' = (Z'[k] + conj Z'[P-k]) / 2,
// B' = (Z'[k] - conj Z'[P-k]) / 2i; the imaginary parts of A[0], A[P/2] are dropped on load as a c2r transform ignores
// them.  grid (P/2+1, P/2+1), (P/8) x 4 threads.
// ---------------------------------------------------------------------------------------------
template <int NS, int R>
__global__ __launch_bounds__((f8_n<NS, R>() / 8) * 4) void k_fft_x_conv(float2* __restrict__ C, int ncp, int NP,
                                                                      const float* __restrict__ tab, int tabN, float nf,
                                                                      float rnf, float rs, const float2* __restrict__ tw,
                                                                      const int* __restrict__ stop)
{
    constexpr int P = f8_n<NS, R>(), NT8 = P / 8, h = P / 2, NCOL = 4, PITCH = NCOL + 1, NTHR = NT8 * NCOL;
    if (stop && *stop) return;
    extern __shared__ float2 f8_lds[];
    float2* sTw = f8_lds + f8_rows<NS, R>() * PITCH;
    float* sval = reinterpret_cast<float*>(sTw + P);   // [h + 1]
    __shared__ int rowOff[8];
    __shared__ int sNRows;
    const int j = blockIdx.x, k = blockIdx.y;
    if (j > k) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += NTHR) sTw[i] = tw[i];
    {
        const int qjk = j * j + k * k;
        const float inp2f = (float)(1.0 / (double)pow2f_((float)NP));
        const float s = 1.0f / (float)tabN;
        for (int i = tid; i <= h; i += NTHR) {
            const float xq = (float)(i * i + qjk) * inp2f;
            const int idx = (int)rintf(div_by_const(xq - 0.0f, s, rs));
            sval[i] = tab[idx < tabN ? idx : tabN];
        }
    }
    if (tid == 0) {
        int n = 0;
        for (int v = 0; v < 8; v++) {
            const int swap = v >> 2, sj = (v >> 1) & 1, sk = v & 1;
            if ((swap && j == k) || (sj && j == 0) || (sk && k == 0)) continue;
            const int ja = sj ? -j : j, kb = sk ? -k : k;
            const int a = swap ? kb : ja, b = swap ? ja : kb;   // a = row (j') index, b = slice (k') index
            if (a == h || b == h) continue;                     // +P/2 is not a stored index, -P/2 is
            rowOff[n++] = (b < 0 ? b + P : b) * P + (a < 0 ? a + P : a);
        }
        sNRows = n;
    }
    __syncthreads();
    const int nRows = sNRows, nPairs = (nRows + 1) >> 1;
    // coalesced loads of the stored half rows; Z = A + i B with the Hermitian extension of both
    for (int idx = tid; idx < nPairs * (h + 1); idx += NTHR) {
        const int p = idx / (h + 1), i = idx - p * (h + 1);
        float2 A = C[(size_t)rowOff[2 * p] * ncp + i];
        float2 B = (2 * p + 1 < nRows) ? C[(size_t)rowOff[2 * p + 1] * ncp + i] : make_float2(0.f, 0.f);
        if (i == 0 || i == h) { A.y = 0.f; B.y = 0.f; }
        f8_lds[f8_slot(i, p, PITCH)] = make_float2(A.x - B.y, A.y + B.x);
        if (i > 0 && i < h) f8_lds[f8_slot(P - i, p, PITCH)] = make_float2(A.x + B.y, B.x - A.y);
    }
    __syncthreads();
    const int c = tid & (NCOL - 1), t = tid / NCOL;
    const bool live = c < nPairs;   // idle columns still take part in the barriers
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = live ? f8_lds[f8_slot(f8_epre<NS, R>(t, q), c, PITCH)] : make_float2(0.f, 0.f);
    __syncthreads();
    fftN<NS, R, 1>(v, t, c, PITCH, f8_lds, sTw);
    {
        const float rn = (float)(1.0 / ((double)P * P * P));   // P^3 is a power of two: exact
#pragma unroll
        for (int n = 0; n < 8; n++) {
            const int iw = f8_eout<NS, R>(t, n);
            const float sv = sval[iw >= h ? P - iw : iw];
            v[n] = make_float2(div_by_const((v[n].x * rn) * sv, nf, rnf), div_by_const((v[n].y * rn) * sv, nf, rnf));
        }
    }
    f8_rearrange<NS, R>(v, t, c, PITCH, f8_lds);
    fftN<NS, R, -1>(v, t, c, PITCH, f8_lds, sTw);
#pragma unroll
    for (int n = 0; n < 8; n++) f8_lds[f8_slot(f8_eout<NS, R>(t, n), c, PITCH)] = v[n];
    __syncthreads();
    for (int idx = tid; idx < nPairs * (h + 1); idx += NTHR) {
        const int p = idx / (h + 1), i = idx - p * (h + 1);
        const float2 U = f8_lds[f8_slot(i, p, PITCH)];
        const float2 Vr = f8_lds[f8_slot((P - i) & (P - 1), p, PITCH)];   // conj(V) = (Vr.x, -Vr.y)
        C[(size_t)rowOff[2 * p] * ncp + i] = make_float2(0.5f * (U.x + Vr.x), 0.5f * (U.y - Vr.y));
        if (2 * p + 1 < nRows) C[(size_t)rowOff[2 * p + 1] * ncp + i] = make_float2(0.5f * (U.y + Vr.y), 0.5f * (Vr.x - U.x));
    }
}

// ---------------------------------------------------------------------------------------------
// z direction, fused with the weight update.  Column (ky = blockIdx.y, x = blockIdx.x TX + c), all kz.
//   FIRST:  C = T W                                   -> inverse z transform        (before round 0)
//   else :  forward z transform -> W /= max(|C|, 1e-6) inside the sphere, checkC max |(|C| - 1)|, C = T W -> inverse z
// W, T rows are P/2+1 long, C rows ncp.  diffBits: float bits of the running maximum (>= 0: uint order == float order).
// ---------------------------------------------------------------------------------------------
// TILED: W and T are read from / written to the z pass's own layout [ky][x tile][kz][TX] (k_tile_real below), where the 4-byte
// values one workgroup touches are one contiguous run of P TX floats; in the volume's layout they are P segments of TX floats
// (32 bytes at P = 1024) a whole z plane apart, and such a pass runs at the fabric's REQUEST rate, not at its bandwidth
// (P = 1024: 335 M requests, 60 % of them for W and T, in 7.9 ms = 43 G/s).
// WPS: waves per SIMD the register allocation aims at (HIP's second launch-bound argument).  A 1024-thread workgroup is 4 waves per
// SIMD: 8 lets two of them share a CU but caps the kernel at 64 VGPRs -- the P = 1024 instance then spills 46 dwords per thread
// to scratch -- 4 gives it 128 VGPRs and one workgroup per CU.
template <int NS, int R, int TX, bool FIRST, bool TILED, int WPS>
__global__ __launch_bounds__((f8_n<NS, R>() / 8) * TX, WPS) void k_fft_z_update(float2* __restrict__ C, float* __restrict__ W,
                                                                         const float* __restrict__ T, int ncp, int r2i,
                                                                         unsigned* __restrict__ diffBits,
                                                                         const float2* __restrict__ tw, const int* __restrict__ stop)
{
    constexpr int P = f8_n<NS, R>(), NT8 = P / 8, nc = P / 2 + 1, NTHR = NT8 * TX;
    if (stop && *stop) return;
    extern __shared__ float2 f8_lds[];
    float2* sTw = f8_lds + f8_rows<NS, R>() * TX;
    __shared__ float sred[16];
    const int c = threadIdx.x % TX, t = threadIdx.x / TX;
    int bx, jw;
    f8_xcd_tile(bx, jw);
    const int x = bx * TX + c;
    const bool ok = x < nc;
    for (int i = threadIdx.x; i < P; i += NTHR) sTw[i] = tw[i];
    float2 v[8];
    float d = 0.f;
    const long strideE = (long)P * ncp;
    float2* base = C + (long)jw * ncp + x;
    if (!FIRST) {
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = ok ? base[(long)f8_epre<NS, R>(t, q) * strideE] : make_float2(0.f, 0.f);
        __syncthreads();
        fftN<NS, R, -1>(v, t, c, TX, f8_lds, sTw);
    } else {
        __syncthreads();
    }
    {
        const int j = jw >= P / 2 ? jw - P : jw;
        const double qij = (double)x * x + (double)j * j;
        const double r2 = (double)pow2f_((float)r2i);
#pragma unroll
        for (int n = 0; n < 8; n++) {
            const int kw = f8_eout<NS, R>(t, n);
            const int k = kw >= P / 2 ? kw - P : kw;
            float2 o = make_float2(0.f, 0.f);
            if (ok) {
                const size_t e = TILED ? ((((size_t)jw * gridDim.x + bx) * P + kw) * TX + c) : (((size_t)kw * P + jw) * nc + x);
                float w = W[e];
                if (!FIRST && (qij + (double)k * k < r2)) {
                    const float a = ts_hypot(v[n].x, v[n].y);
                    w = w / (a > (float)1e-6 ? a : (float)1e-6);
                    W[e] = w;
                    d = fmaxf(d, fabsf(a - 1));
                }
                o = make_float2(T[e] * w, 0.0f * w);
            }
            v[n] = o;
        }
    }
    f8_rearrange<NS, R>(v, t, c, TX, f8_lds);
    fftN<NS, R, 1>(v, t, c, TX, f8_lds, sTw);
    if (ok) {
#pragma unroll
        for (int n = 0; n < 8; n++) base[(long)f8_eout<NS, R>(t, n) * strideE] = v[n];
    }
    if (!FIRST) {
        d = wave_max(d);
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (NTHR + 63) / 64; w++) d = fmaxf(d, sred[w]);
            const unsigned bitsd = __float_as_uint(d);
            if (bitsd > __hip_atomic_load(diffBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(diffBits, bitsd);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same pass with SIXTEEN points per thread (round 5; R = 2 grids, i.e. P = 1024): a physical thread T < P / 16 plays the two
// threads tau = T and tau = T + P / 16 of the eight-point form, stage by stage, with the barriers shared -- the same loads, the same
// LDS traffic, the same arithmetic in the same order (bit-identical results), but HALF the threads per workgroup: two workgroups
// of 512 threads fit a CU (2 x 75 KB of LDS, 128 VGPRs each) where the 1 024-thread form runs alone, so one workgroup's loads
// and stores overlap the other's transforms.  The 1 024-thread form alternates between memory and arithmetic with nothing to
// cover either (7.0 ms per round at 1 024^3 = 2.2 TB/s against 4.9 TB/s for the y passes).
// ---------------------------------------------------------------------------------------------
template <int NS, int DIR, int TWS>
__device__ __forceinline__ void fft8n_x2(float2 v[2][8], int t, int c, int pitch, float2* s, const float2* sTw, const int eoff[2])
{
    static_assert(NS == 3, "two-thread form: 512-point sub-transforms only");
#pragma unroll
    for (int u = 0; u < 2; u++) {
        fft8<DIR>(v[u]);
        const int n1 = t >> 3;
#pragma unroll
        for (int k0 = 0; k0 < 8; k0++)
            s[f8_slot(eoff[u] + k0 * 64 + t, c, pitch)] = cmul(v[u][k0], f8_tw<DIR>(sTw[((8 * n1 * k0) & 511) * TWS]));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int n0 = t & 7, k0 = t >> 3;
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) v[u][n1] = s[f8_slot(eoff[u] + k0 * 64 + n1 * 8 + n0, c, pitch)];
        fft8<DIR>(v[u]);
#pragma unroll
        for (int k1 = 0; k1 < 8; k1++)
            s[f8_slot(eoff[u] + k0 * 64 + k1 * 8 + n0, c, pitch)] = cmul(v[u][k1], f8_tw<DIR>(sTw[(n0 * (k0 + 8 * k1)) * TWS]));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int k0 = t & 7, k1 = t >> 3;
#pragma unroll
        for (int n0 = 0; n0 < 8; n0++) v[u][n0] = s[f8_slot(eoff[u] + k0 * 64 + k1 * 8 + n0, c, pitch)];
        fft8<DIR>(v[u]);
    }
    __syncthreads();
}

// fftN<3, 2, DIR> for the two threads tau = T and T + 64 a physical thread T < 64 plays (M = 512, M / 8 = 64, N / 8 = 128)
template <int DIR>
__device__ __forceinline__ void fftN_x2(float2 v[2][8], int T, int c, int pitch, float2* s, const float2* sTw)
{
    constexpr int M = 512, TT = 128;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int tau = T + 64 * u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float2* w = v[u] + 2 * i;
            const float2 a0 = w[0], a1 = w[1];
            w[0] = f8_add(a0, a1); w[1] = f8_sub(a0, a1);
            const int b = tau + TT * i;
#pragma unroll
            for (int k0 = 0; k0 < 2; k0++) s[f8_slot(k0 * M + b, c, pitch)] = cmul(w[k0], f8_tw<DIR>(sTw[b * k0]));
        }
    }
    __syncthreads();
    // tau = T: k0 = 0, t = T; tau = T + 64: k0 = 1, t = T
    const int eoff[2] = {0, M};
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int n = 0; n < 8; n++) v[u][n] = s[f8_slot(eoff[u] + T + 64 * n, c, pitch)];
    fft8n_x2<3, DIR, 2>(v, T, c, pitch, s, sTw, eoff);
}

__device__ __forceinline__ void f8_rearrange_x2(float2 v[2][8], int T, int c, int pitch, float2* s)
{
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int k = 0; k < 8; k++) s[f8_slot(f8_eout<3, 2>(T + 64 * u, k), c, pitch)] = v[u][k];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int q = 0; q < 8; q++) v[u][q] = s[f8_slot(f8_epre<3, 2>(T + 64 * u, q), c, pitch)];
    __syncthreads();
}

template <int TX, bool FIRST, bool TILED>
__global__ __launch_bounds__(64 * TX, 4) void k_fft_z_update_x2(float2* __restrict__ C, float* __restrict__ W, const float* __restrict__ T,
                                                               int ncp, int r2i, unsigned* __restrict__ diffBits,
                                                               const float2* __restrict__ tw, const int* __restrict__ stop)
{
    constexpr int P = 1024, nc = P / 2 + 1, NTHR = 64 * TX;
    if (stop && *stop) return;
    extern __shared__ float2 f8_lds[];
    float2* sTw = f8_lds + f8_rows<3, 2>() * TX;
    __shared__ float sred[16];
    const int c = threadIdx.x % TX, t = threadIdx.x / TX;   // t < 64
    int bx, jw;
    f8_xcd_tile(bx, jw);
    const int x = bx * TX + c;
    const bool ok = x < nc;
    for (int i = threadIdx.x; i < P; i += NTHR) sTw[i] = tw[i];
    float2 v[2][8];
    float d = 0.f;
    const long strideE = (long)P * ncp;
    float2* base = C + (long)jw * ncp + x;
    if (!FIRST) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int q = 0; q < 8; q++) v[u][q] = ok ? base[(long)f8_epre<3, 2>(t + 64 * u, q) * strideE] : make_float2(0.f, 0.f);
        __syncthreads();
        fftN_x2<-1>(v, t, c, TX, f8_lds, sTw);
    } else {
        __syncthreads();
    }
    {
        const int j = jw >= P / 2 ? jw - P : jw;
        const double qij = (double)x * x + (double)j * j;
        const double r2 = (double)pow2f_((float)r2i);
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int n = 0; n < 8; n++) {
                const int kw = f8_eout<3, 2>(t + 64 * u, n);
                const int k = kw >= P / 2 ? kw - P : kw;
                float2 o = make_float2(0.f, 0.f);
                if (ok) {
                    const size_t e = TILED ? ((((size_t)jw * gridDim.x + bx) * P + kw) * TX + c) : (((size_t)kw * P + jw) * nc + x);
                    float w = W[e];
                    if (!FIRST && (qij + (double)k * k < r2)) {
                        const float a = ts_hypot(v[u][n].x, v[u][n].y);
                        w = w / (a > (float)1e-6 ? a : (float)1e-6);
                        W[e] = w;
                        d = fmaxf(d, fabsf(a - 1));
                    }
                    o = make_float2(T[e] * w, 0.0f * w);
                }
                v[u][n] = o;
            }
    }
    f8_rearrange_x2(v, t, c, TX, f8_lds);
    fftN_x2<1>(v, t, c, TX, f8_lds, sTw);
    if (ok) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int n = 0; n < 8; n++) base[(long)f8_eout<3, 2>(t + 64 * u, n) * strideE] = v[u][n];
    }
    if (!FIRST) {
        d = wave_max(d);
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (NTHR + 63) / 64; w++) d = fmaxf(d, sred[w]);
            const unsigned bitsd = __float_as_uint(d);
            if (bitsd > __hip_atomic_load(diffBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(diffBits, bitsd);
        }
    }
}

// natural [kz][ky][nc] <-> tiled [ky][x tile][kz][TX] (columns beyond nc are zero), 16 z planes of one ky row per workgroup
// through LDS so that both sides move whole lines.  grid (P / 16, P), 256 threads, LDS 16 nTx TX floats.
template <int TX>
__global__ __launch_bounds__(256) void k_tile_real(float* __restrict__ dst, const float* __restrict__ src, int P, int nc, int nTx, int toTiled)
{
    constexpr int KC = 16;
    extern __shared__ float f8_tile[];
    const int jw = blockIdx.y, k0 = blockIdx.x * KC, Wd = nTx * TX;
    if (toTiled) {
        for (int i = threadIdx.x; i < KC * Wd; i += 256) {
            const int kk = i / Wd, x = i - kk * Wd;
            f8_tile[i] = x < nc ? src[((size_t)(k0 + kk) * P + jw) * nc + x] : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nTx * KC * TX; i += 256) {
            const int bx = i / (KC * TX), r = i - bx * (KC * TX), kk = r / TX, c = r - kk * TX;
            dst[(((size_t)jw * nTx + bx) * P + k0 + kk) * TX + c] = f8_tile[kk * Wd + bx * TX + c];
        }
    } else {
        for (int i = threadIdx.x; i < nTx * KC * TX; i += 256) {
            const int bx = i / (KC * TX), r = i - bx * (KC * TX), kk = r / TX, c = r - kk * TX;
            f8_tile[kk * Wd + bx * TX + c] = src[(((size_t)jw * nTx + bx) * P + k0 + kk) * TX + c];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < KC * Wd; i += 256) {
            const int kk = i / Wd, x = i - kk * Wd;
            if (x < nc) dst[((size_t)(k0 + kk) * P + jw) * nc + x] = f8_tile[i];
        }
    }
}

}  // namespace thx
