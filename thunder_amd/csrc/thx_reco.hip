// thx_reco.hip -- Reconstructor state and the reconstruct() pipeline on the device:
// Wiener term, iterative gridding weights (hipFFT/rocFFT 3-D c2r/r2c), F*W, inverse FFT, crop, TIK correction;
// Projector::setProjectee; FSC.  Reference behaviour: src/Reconstructor.cpp:43-160,1129-1831,2563-2674,
// src/Projector.cpp:123-148,524-606, src/Functions/Spectrum.cpp:302-337, src/FFT.cpp:176-232.  gfx950 only.
#include <hipfft/hipfft.h>
#include <math.h>

#include <map>
#include <tuple>
#include <mutex>
#include <utility>
#include <vector>

#include <type_traits>

#include "thx_common.h"

namespace thx {

#define THX_FFT_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipfftResult _r = (expr);                                                            \
        if (_r != HIPFFT_SUCCESS) {                                                          \
            thx::set_error("%s failed: hipfftResult %d (%s:%d)", #expr, (int)_r, __FILE__, __LINE__); \
            return 1000 + (int)_r;                                                           \
        }                                                                                    \
    } while (0)

// ---- host: MKB_RL / MKB_RL_R2 (src/Functions/Functions.cpp:143-214, FUNCTIONS_MKB_ORDER_0) ----
static double bessel_I0(double x)
{
    double q = x * x / 4.0, term = 1.0, sum = 1.0;
    for (int k = 1; k < 500; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-17) break;
    }
    return sum;
}
static double bessel_I15(double v) { return sqrt(2.0 / (M_PI * v)) * (cosh(v) - sinh(v) / v); }
static double bessel_J15(double v) { return sqrt(2.0 / (M_PI * v)) * (sin(v) / v - cos(v)); }

static float mkb_rl(float r, float a, float alpha)
{
    float u = (float)(2 * M_PI * a * r);
    float v = (u <= alpha) ? sqrtf(pow2f_(alpha) - pow2f_(u)) : sqrtf(pow2f_(u) - pow2f_(alpha));
    float I0a = (float)bessel_I0((double)alpha);
    float w = (float)(pow(2 * M_PI, 1.5) * pow3f_(a) / I0a / pow((double)v, 1.5));
    return (u <= alpha) ? w * (float)bessel_I15((double)v) : w * (float)bessel_J15((double)v);
}
static float mkb_rl_r2(float r2, float a, float alpha)
{
    float u2 = pow2f_((float)(2 * M_PI * a)) * r2;
    float v = (u2 <= pow2f_(alpha)) ? sqrtf(pow2f_(alpha) - u2) : sqrtf(u2 - pow2f_(alpha));
    float I0a = (float)bessel_I0((double)alpha);
    float w = (float)(pow(2 * M_PI, 1.5) * pow3f_(a) / I0a / pow((double)v, 1.5));
    return (u2 <= pow2f_(alpha)) ? w * (float)bessel_I15((double)v) : w * (float)bessel_J15((double)v);
}

constexpr int kTabN = 100000;  // _kernelRL.init(..., 0, 1, 1e5), src/Reconstructor.cpp:77-86

// ---- device kernels ----
__device__ __forceinline__ void unpack_half(size_t e, int P, int& i, int& j, int& k)
{
    // the half grid has < 2^32 voxels up to P = 2048: 32-bit div/mod (64-bit ones cost ~10x more on the VALU)
    const unsigned nc = P / 2 + 1, e32 = (unsigned)e;
    const unsigned row = e32 / nc;
    i = (int)(e32 - row * nc);
    const unsigned kw = row / (unsigned)P, jw = row - kw * (unsigned)P;
    j = (int)jw >= P / 2 ? (int)jw - P : (int)jw;
    k = (int)kw >= P / 2 ? (int)kw - P : (int)kw;
}

// [MAP] T /= FSC'(shell), src/Reconstructor.cpp:1242-1270
__global__ __launch_bounds__(256) void k_wiener_T(float* __restrict__ T, int P, int pf, int maxRadius,
                                                  const float* __restrict__ FSC, int nFSC, int joinHalf, int wienerF)
{
    const size_t n = (size_t)P * P * (P / 2 + 1);
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int i, j, k;
    unpack_half(e, P, i, j, k);
    const double q = (double)i * i + (double)j * j + (double)k * k;
    if ((q >= pow2f_((float)(wienerF * pf))) && (q < pow2f_((float)(maxRadius * pf)))) {
        const int u = (int)rint(gsl_hypot3_((double)i, (double)j, (double)k));
        float f = (u / pf >= nFSC) ? 0.f : FSC[u / pf];
        const float lo = (float)1e-3, hi = (float)(1 - 1e-3);
        const float mn = hi < f ? hi : f;
        f = lo > mn ? lo : mn;
        if (joinHalf) f = sqrtf(2 * f / (1 + f));
        T[e] = T[e] / f;
    }
}

// W = 1 in the sphere else 0 (:1299-1304); T = max(T, 1e-25) (:1322-1324)
__global__ __launch_bounds__(256) void k_initW_floorT(float* __restrict__ W, float* __restrict__ T, int P, int pf,
                                                      int maxRadius)
{
    const size_t n = (size_t)P * P * (P / 2 + 1);
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int i, j, k;
    unpack_half(e, P, i, j, k);
    const double q = (double)i * i + (double)j * j + (double)k * k;
    W[e] = (q < pow2f_((float)(maxRadius * pf))) ? 1.0f : 0.0f;
    const float t = T[e];
    T[e] = t > (float)1e-25 ? t : (float)1e-25;
}

// Row length of the library-internal complex work grid C: rocFFT's strided passes over a [P][P][P/2+1] grid run at
// 1.8 TB/s because rows of 257 complex values start on odd 8-byte boundaries; with the rows padded to a multiple of 8
// elements (64 B) the same transforms take 1.05 ms instead of 1.44 ms at P = 512 (tools/probes/fft_layout_probe.hip).
__host__ __device__ __forceinline__ int padded_nc(int P) { return ((P / 2 + 1) + 7) & ~7; }

// correctly rounded a / b from the correctly rounded reciprocal rb = RN(1 / b) (Markstein): q0 = a rb, r = a - q0 b
// (exact, fma), q = q0 + r rb.  Replaces the ~12-instruction IEEE division sequence for divisors that are launch constants.
__device__ __forceinline__ float div_by_const(float a, float b, float rb)
{
    const float q0 = a * rb;
    const float rem = fmaf(-q0, b, a);
    return fmaf(rem, rb, q0);
}

// convoluteC real-space stage (:2635-2652) fused with bwExecutePlan's 1/size scaling (src/FFT.cpp:355-367).
// The tabulated kernel value depends only on q = i^2 + j^2 + k^2, and a lookup per voxel is bound by L2->L1 line traffic
// (each 4-byte gather from the 400 KB table pulls a whole line: 635 us at P = 512).  One workgroup therefore takes a
// canonical pair 0 <= j <= k <= P/2, looks the P/2+1 values of its rows up ONCE into LDS, and streams every row that
// shares them -- (+-j, +-k) and (+-k, +-j), up to 8 rows, each using a value for +i and -i: 16x fewer gathers.
// POW2: P^3 and (N*pf)^2 are powers of two, so v / size and q / (N*pf)^2 are exact float scalings (q < 2^24 is an exact
// float) and the reference's double-precision detours produce the same bits as the float operations used here.
constexpr int kMaxHalfP = 1024;  // P <= 2048
template <bool POW2>
__global__ __launch_bounds__(256) void k_convolute_rl(float* __restrict__ rl, int P, int NP, const float* __restrict__ tab,
                                                      float nf, float rnf, float rs, int applyScale)
{
    __shared__ float sval[kMaxHalfP + 1];
    const int h = P / 2;
    const int j = blockIdx.x, k = blockIdx.y;
    if (j > k) return;
    const size_t n = (size_t)P * P * P;
    const int qjk = j * j + k * k;
    const double np2 = (double)pow2f_((float)NP);
    const float inp2f = (float)(1.0 / np2);
    const double rn = 1.0 / (double)n;
    const float rnf32 = (float)rn;
    const float s = 1.0f / kTabN;  // _s = (_b - _a) / _n in RFLOAT, src/TabFunction.cpp:34
    for (int i = threadIdx.x; i <= h; i += blockDim.x) {
        const int qi = i * i + qjk;
        const float x = POW2 ? (float)qi * inp2f : (float)((double)qi / np2);
        const int idx = (int)rintf(div_by_const(x - 0.0f, s, rs));
        sval[i] = tab[idx < kTabN ? idx : kTabN];
    }
    // the rows sharing these values (uniform across the workgroup); +P/2 is not a stored index, -P/2 is
    int rowOff[8], nRows = 0;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        const int swap = v >> 2, sj = (v >> 1) & 1, sk = v & 1;
        if ((swap && j == k) || (sj && j == 0) || (sk && k == 0)) continue;
        const int ja = sj ? -j : j, kb = sk ? -k : k;
        const int a = swap ? kb : ja, b = swap ? ja : kb;  // a = row (j') index, b = slice (k') index
        if (a == h || b == h) continue;
        rowOff[nRows++] = (b < 0 ? b + P : b) * P + (a < 0 ? a + P : a);
    }
    __syncthreads();
    const int per = P / 4;
    for (int item = threadIdx.x; item < nRows * per; item += blockDim.x) {
        const int rsel = item / per, i4 = item - rsel * per;
        int off = rowOff[0];
#pragma unroll
        for (int v = 1; v < 8; v++) off = (rsel == v) ? rowOff[v] : off;
        float4* row = reinterpret_cast<float4*>(rl + (size_t)off * P);
        float4 v4 = row[i4];
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int iw = 4 * i4 + c;
            const int ai = iw >= h ? P - iw : iw;  // |i|
            const float vv = !applyScale ? v[c] : (POW2 ? v[c] * rnf32 : (float)((double)v[c] * rn));
            v[c] = div_by_const(vv * sval[ai], nf, rnf);
        }
        row[i4] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__device__ __forceinline__ float ts_hypot(float x, float y)
{
    float xabs = fabsf(x), yabs = fabsf(y), mn, mx;
    if (xabs < yabs) { mn = xabs; mx = yabs; } else { mn = yabs; mx = xabs; }
    if (mn == 0) return mx;
    float u = mn / mx;
    return mx * sqrtf(1 + u * u);
}

// One sweep per balancing round over the half grid, row (jw, kw) per workgroup:
//   UPDATE: W /= max(|C|, 1e-6) in the sphere (:1487-1496) + checkC max (RECONSTRUCTOR_CHECK_C_MAX, :2563-2592), then
//   always: C = T * REAL(W) for the next round (:1389-1391) -- fused so that W is read once and the next round's
//   input is written while the row is in registers.  C rows are ncp long (padded_nc), T / W rows P/2+1.
template <bool UPDATE>
__global__ __launch_bounds__(256) void k_updateW_calcC(float* __restrict__ W, float2* __restrict__ C,
                                                       const float* __restrict__ T, int P, int ncp, int pf, int maxRadius,
                                                       unsigned* __restrict__ diffBits)
{
    __shared__ float sred[4];
    const int nc = P / 2 + 1;
    const int jw = blockIdx.x, kw = blockIdx.y;
    const int j = jw >= P / 2 ? jw - P : jw, k = kw >= P / 2 ? kw - P : kw;
    const double qjk = (double)j * j + (double)k * k;
    const double r2 = (double)pow2f_((float)(maxRadius * pf));
    const size_t base = ((size_t)kw * P + jw) * nc, baseC = ((size_t)kw * P + jw) * ncp;
    float d = 0.f;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) {
        float w = W[base + i];
        if (UPDATE && ((double)i * i + qjk < r2)) {
            const float2 c = C[baseC + i];
            const float a = ts_hypot(c.x, c.y);
            const float m = a > (float)1e-6 ? a : (float)1e-6;
            w = w / m;
            W[base + i] = w;
            d = fmaxf(d, fabsf(a - 1));
        }
        C[baseC + i] = make_float2(T[base + i] * w, 0.0f * w);
    }
    if (UPDATE) {
        d = wave_max(d);
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            d = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
            // d >= 0: uint order == float order.  One hot word retires only ~88 atomics/us, so a workgroup whose maximum
            // cannot raise the current value (plain L2 read) skips the atomic.
            const unsigned bitsd = __float_as_uint(d);
            if (bitsd > __hip_atomic_load(diffBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(diffBits, bitsd);
        }
    }
}

}  // namespace thx
#include "thx_fft8.h"
namespace thx {

// no grid correction: W = 1 / max(|T|, 1e-6) in the sphere (:1566-1578)
__global__ __launch_bounds__(256) void k_W_nogridcorr(float* __restrict__ W, const float* __restrict__ T, int P, int pf,
                                                      int maxRadius)
{
    const size_t n = (size_t)P * P * (P / 2 + 1);
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int i, j, k;
    unpack_half(e, P, i, j, k);
    const double q = (double)i * i + (double)j * j + (double)k * k;
    if (q < pow2f_((float)(maxRadius * pf))) {
        const float a = ts_hypot(T[e], 0.f);
        W[e] = (float)(1.0 / (double)(a > (float)1e-6 ? a : (float)1e-6));
    }
}

// padDst = F * W in the sphere, 0 elsewhere (:1678-1701).  pad grid PN = _N*_pf, F grid PF = _pf*_size.
__global__ __launch_bounds__(256) void k_FW(float2* __restrict__ pad, int PN, int ncp, const float2* __restrict__ F,
                                            const float* __restrict__ W, int PF, int pf, int maxRadius)
{
    const size_t n = (size_t)PN * PN * (PN / 2 + 1);
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int i, j, k;
    unpack_half(e, PN, i, j, k);
    const size_t eo = (size_t)((unsigned)e / (unsigned)(PN / 2 + 1)) * ncp + i;  // padded row of the work grid
    float2 o = make_float2(0.f, 0.f);
    const double q = (double)i * i + (double)j * j + (double)k * k;
    const int h = PF / 2;
    if ((q < pow2f_((float)(maxRadius * pf))) && i <= h && j >= -h && j < h && k >= -h && k < h) {
        const size_t f = ((size_t)(k >= 0 ? k : k + PF) * PF + (j >= 0 ? j : j + PF)) * (PF / 2 + 1) + i;
        const float2 a = F[f];
        const float b0 = W[f], b1 = 0.f;
        o = make_float2(a.x * b0 - a.y * b1, a.x * b1 + a.y * b0);
    }
    pad[eo] = o;
}

// fft.bw 1/size + VOL_EXTRACT_RL + TIK correction (:1716-1802)
__global__ __launch_bounds__(256) void k_extract_tik(float* __restrict__ dst, const float* __restrict__ pad, int P, int N,
                                                     int pf, int corr)
{
    const size_t n = (size_t)N * N * N;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int iw = (int)(e % N), jw = (int)((e / N) % N), kw = (int)(e / ((size_t)N * N));
    const int i = iw >= N / 2 ? iw - N : iw, j = jw >= N / 2 ? jw - N : jw, k = kw >= N / 2 ? kw - N : kw;
    const size_t ip = ((size_t)(k >= 0 ? k : k + P) * P + (j >= 0 ? j : j + P)) * P + (i >= 0 ? i : i + P);
    float v = (float)((double)pad[ip] * (1.0 / ((double)P * P * P)));
    if (corr) v = v / tik_rl((float)(gsl_hypot3_((double)i, (double)j, (double)k) / (pf * N)));
    dst[e] = v;
}

// VOL_PAD_RL + gridCorrection LINEAR branch (src/Projector.cpp:573-583): pad = src / TIK_RL(|x| / (pf * P))
__global__ __launch_bounds__(256) void k_pad_gridcorr(float* __restrict__ pad, const float* __restrict__ src, int N, int pf)
{
    const int P = N * pf;
    const size_t n = (size_t)P * P * P;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int iw = (int)(e % P), jw = (int)((e / P) % P), kw = (int)(e / ((size_t)P * P));
    const int i = iw >= P / 2 ? iw - P : iw, j = jw >= P / 2 ? jw - P : jw, k = kw >= P / 2 ? kw - P : kw;
    float v = 0.f;
    const int h = N / 2;
    if (i >= -h && i < h && j >= -h && j < h && k >= -h && k < h) {
        const size_t is = ((size_t)(k >= 0 ? k : k + N) * N + (j >= 0 ? j : j + N)) * N + (i >= 0 ? i : i + N);
        v = src[is] / tik_rl((float)(gsl_hypot3_((double)i, (double)j, (double)k) / (pf * P)));
    }
    pad[e] = v;
}

__global__ __launch_bounds__(256) void k_scale_rl(float* __restrict__ rl, size_t n)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) rl[e] = (float)((double)rl[e] * (1.0 / (double)n));
}

// FSC shell sums (src/Functions/Spectrum.cpp:302-337), DETERMINISTIC: no atomics anywhere.  A wave owns a fixed set of rows
// (j, k) of the half grid; along a row the shell index u = AROUND(NORM_3(i, j, k)) never decreases, so the lanes of a 64-element
// chunk that share a shell are contiguous and a segmented wave scan (fixed tree order) leaves each shell's chunk total in the
// last lane of its segment, which adds it to the wave's PRIVATE table in LDS (distinct shells: no conflict; chunks and rows
// in program order).  Every wave writes its table; k_fsc_reduce adds the tables in wave order.  (The first version summed
// floats with LDS atomics: the curve moved by an ulp from run to run, and the MAP reconstruction's stop rule -- DESIGN 3a --
// turned that ulp into another round count.)
constexpr int kFscBlocks = 256;
__global__ __launch_bounds__(256) void k_fsc_rows(double* __restrict__ part, int nShell, const float2* __restrict__ A,
                                                  const float2* __restrict__ B, int P)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* my = reinterpret_cast<double*>(smem_raw) + (size_t)wave * 3 * nShell;
    for (int q = lane; q < 3 * nShell; q += 64) my[q] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int nc = P / 2 + 1;
    const long rows = (long)P * P;
    const int gw = blockIdx.x * 4 + wave, nW = gridDim.x * 4;
    for (long row = gw; row < rows; row += nW) {
        const int jw = (int)(row % P), kw = (int)(row / P);
        const int j = jw >= P / 2 ? jw - P : jw, k = kw >= P / 2 ? kw - P : kw;
        for (int i0 = 0; i0 < nc; i0 += 64) {
            const int i = i0 + lane;
            int u = INT_MAX;
            double s0 = 0, s1 = 0, s2 = 0;
            if (i < nc) {
                const int uu = (int)rint(gsl_hypot3_((double)i, (double)j, (double)k));
                if (uu < nShell) {
                    u = uu;
                    const float2 a = A[(size_t)row * nc + i], b = B[(size_t)row * nc + i];
                    s0 = (double)(a.x * b.x + a.y * b.y);
                    s1 = (double)(a.x * a.x + a.y * a.y);
                    s2 = (double)(b.x * b.x + b.y * b.y);
                }
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int uo = __shfl_up(u, o, 64);
                const double t0 = __shfl_up(s0, o, 64), t1 = __shfl_up(s1, o, 64), t2 = __shfl_up(s2, o, 64);
                if (lane >= o && uo == u) { s0 += t0; s1 += t1; s2 += t2; }
            }
            const int un = __shfl_down(u, 1, 64);
            if (u != INT_MAX && (lane == 63 || un != u)) {
                my[u] += s0; my[nShell + u] += s1; my[2 * nShell + u] += s2;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int q = lane; q < 3 * nShell; q += 64) part[(size_t)gw * 3 * nShell + q] = my[q];
}

__global__ void k_fsc_reduce(double* __restrict__ acc, const double* __restrict__ part, int n3, int nW)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n3) return;
    double s = 0;
    for (int w = 0; w < nW; w++) s += part[(size_t)w * n3 + q];
    acc[q] = s;
}

__global__ void k_fsc_final(float* __restrict__ fsc, const double* __restrict__ acc, int nShell)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nShell) return;
    const float vS = (float)acc[u], vA = (float)acc[nShell + u], vB = (float)acc[2 * nShell + u];
    const float AB = sqrtf(vA * vB);
    fsc[u] = (AB == 0) ? 0.f : vS / AB;
}

// Staged form (ExposeWC, Interface.h:392-406; kernel_RecalculateW + kernel_CheckCMAX): W /= max(|C|, 1e-6) in the sphere
// and the checkC maximum, without producing the next round's C (the caller's ExposeC does).  Row (jw, kw) per workgroup.
__global__ __launch_bounds__(256) void k_recalcW_max(float* __restrict__ W, const float2* __restrict__ C, int P, int r2i,
                                                     unsigned* __restrict__ diffBits)
{
    __shared__ float sred[4];
    const int nc = P / 2 + 1;
    const int jw = blockIdx.x, kw = blockIdx.y;
    const int j = jw >= P / 2 ? jw - P : jw, k = kw >= P / 2 ? kw - P : kw;
    const double qjk = (double)j * j + (double)k * k;
    const double r2 = (double)pow2f_((float)r2i);
    const size_t base = ((size_t)kw * P + jw) * nc;
    float d = 0.f;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) {
        if ((double)i * i + qjk < r2) {
            const float2 c = C[base + i];
            const float a = ts_hypot(c.x, c.y);
            W[base + i] = W[base + i] / (a > (float)1e-6 ? a : (float)1e-6);
            d = fmaxf(d, fabsf(a - 1));
        }
    }
    d = wave_max(d);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(diffBits, __float_as_uint(fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]))));
}

// ExposeCorrF (Interface.h:492-501; kernel_CorrectF, RECONSTRUCTOR_TRILINEAR_KERNEL form): dst /= tik[|k|][|j|][|i|]
__global__ __launch_bounds__(256) void k_correctF(float* __restrict__ dst, const float* __restrict__ tik, int dim)
{
    const size_t n = (size_t)dim * dim * dim;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int i = (int)(e % dim), j = (int)((e / dim) % dim), k = (int)(e / ((size_t)dim * dim));
    if (i >= dim / 2) i = dim - i;
    if (j >= dim / 2) j = dim - j;
    if (k >= dim / 2) k = dim - k;
    const int h = dim / 2 + 1;
    dst[e] = dst[e] / tik[((size_t)k * h + j) * h + i];
}

static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

// The stop rule of the gridding loop (src/Reconstructor.cpp:1530-1551: DIFF_C_THRES 1e-2, DIFF_C_DECREASE_THRES 0.95,
// N_DIFF_C_NO_DECREASE 2, comparisons in double against RFLOAT operands) evaluated ON THE DEVICE after every round: the host
// queues all MAX_N_ITER_BALANCE rounds without reading anything back; once `done` is set the kernels of the remaining rounds
// return at entry, so W is exactly what the host-driven loop leaves.
struct RecoStop {
    int done, iters, nNoDec;
    float diffC, diffCPrev;
};

__global__ void k_reco_stop_init(RecoStop* s)
{
    s->done = 0; s->iters = 0; s->nNoDec = 0;
    s->diffC = 3.402823466e+38f; s->diffCPrev = 3.402823466e+38f;
}

__global__ void k_reco_stop_rule(RecoStop* s, const unsigned* diffBits, int m, int minIter)
{
    if (s->done) return;
    const float d = __uint_as_float(*diffBits);
    s->diffCPrev = s->diffC;
    s->diffC = d;
    s->iters = m + 1;
    if ((double)d > (double)s->diffCPrev * 0.95) s->nNoDec += 1; else s->nNoDec = 0;
    if (((double)d < 1e-2) || ((m >= minIter) && (s->nNoDec == 2))) s->done = 1;
}

}  // namespace thx

using namespace thx;

struct thx_reco {
    int size, N, pf, PF, PN;
    float a, alpha, nf;
    float* tab;       // device, kTabN + 1
    float* W;         // device, PF half grid (real)
    float2* C;        // device, max(PF, PN) half grid
    float* rl;        // device, max(PF, PN)^3
    unsigned* diff;   // device scalar
    int* stop;        // device RecoStop: the gridding loop's stop rule evaluated on the device (k_reco_stop_rule)
    int maxIter, minIter;   // MAX_N_ITER_BALANCE 30 / MIN_N_ITER_BALANCE 10 (include/Reconstructor.h); thx_reco_set_balance_rounds
    float* fscDev;    // device, up to 4096 shells
    float rnf, rs;    // RN(1 / nf), RN(1 / table step): launch constants of k_convolute_rl
    // r2cF / c2rF / c2rN work on the padded C grid (rows of padded_nc); r2cStd writes the standard [P][P][P/2+1] layout
    // (Projector volume)
    hipfftHandle r2cF, c2rF, c2rN, r2cStd;
    bool haveN;
    float2* tw;       // device, exp(-2 pi i m / PF), m < PF, when PF is a power of two in 64 .. 1024 (thx_fft8.h)
    int handNS;       // 0: rocFFT only; 2: PF = 64, 128, 256; 3: PF = 512, 1024
};

extern "C" {

static int reco_fill(thx_reco* r, int size, int N, int pf, float a, float alpha);

int thx_reco_create(thx_reco** out, int size, int N, int pf, float a, float alpha)
{
    THX_REQUIRE(out, "out is NULL");
    THX_REQUIRE(size > 0 && N >= size && pf >= 1 && (size % 2 == 0) && (N % 2 == 0), "bad size / N / pf");
    THX_REQUIRE((pf * size) % 4 == 0, "pf * size must be a multiple of 4");
    THX_REQUIRE(pf * size <= 2 * kMaxHalfP && pf * N <= 2 * kMaxHalfP, "grids above 2048 are not supported");
    thx_reco* r = new thx_reco();
    memset(r, 0, sizeof(*r));
    const int rc = reco_fill(r, size, N, pf, a, alpha);
    if (rc) { (void)thx_reco_destroy(r); return rc; }   // no leak of the partly built handle (buffers, plans)
    *out = r;
    return 0;
}

static int reco_fill(thx_reco* r, int size, int N, int pf, float a, float alpha)
{
    r->size = size; r->N = N; r->pf = pf; r->PF = pf * size; r->PN = pf * N; r->a = a; r->alpha = alpha;
    r->maxIter = 30; r->minIter = 10;
    r->nf = mkb_rl(0.f, a, alpha);  // nf = MKB_RL(0, _a, _alpha), src/Reconstructor.cpp:2600
    std::vector<float> tab(kTabN + 1);
    {
        const float ta = 0.f, tb = 1.f;
        const float s = (tb - ta) / kTabN;
        for (int i = 0; i <= kTabN; i++) tab[i] = mkb_rl_r2(ta + i * s, a, alpha);
    }
    const int PM = r->PN > r->PF ? r->PN : r->PF;
    const size_t nHalfF = (size_t)r->PF * r->PF * (r->PF / 2 + 1);
    const size_t nHalfM = (size_t)PM * PM * padded_nc(PM);
    r->rnf = (float)(1.0 / (double)r->nf);
    r->rs = (float)(1.0 / (double)(1.0f / kTabN));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(&r->tab), (kTabN + 1) * sizeof(float)));
    THX_CHECK(hipMemcpy(r->tab, tab.data(), (kTabN + 1) * sizeof(float), hipMemcpyHostToDevice));
    // (the hand-written loop keeps W tiled by z column while it runs: rows padded to a whole number of x tiles)
    size_t nW = nHalfF;
    {
        const int pfP = r->PF, nt8 = pfP / 8, txz = nt8 * 16 <= 1024 ? 16 : (nt8 * 8 <= 1024 ? 8 : 4);
        const size_t tiled = (size_t)pfP * pfP * (size_t)(((pfP / 2 + 1) + txz - 1) / txz) * txz;
        if (tiled > nW) nW = tiled;
    }
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(&r->W), nW * sizeof(float)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(&r->C), nHalfM * sizeof(float2)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(&r->rl), (size_t)PM * PM * PM * sizeof(float)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(&r->diff), sizeof(unsigned)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(&r->stop), sizeof(RecoStop)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(&r->fscDev), 4096 * sizeof(float)));
    auto plan_padded = [](hipfftHandle* h, int P, hipfftType type) -> hipfftResult {
        int n[3] = {P, P, P};
        int cE[3] = {P, P, padded_nc(P)}, rE[3] = {P, P, P};
        if (type == HIPFFT_C2R) return hipfftPlanMany(h, 3, n, cE, 1, P * P * padded_nc(P), rE, 1, P * P * P, type, 1);
        return hipfftPlanMany(h, 3, n, rE, 1, P * P * P, cE, 1, P * P * padded_nc(P), type, 1);
    };
    r->handNS = (r->PF == 512 || r->PF == 1024) ? 3 : ((r->PF == 64 || r->PF == 128 || r->PF == 256) ? 2 : 0);
    if (r->handNS) {
        std::vector<float2> tw(r->PF);
        for (int m = 0; m < r->PF; m++) {
            const double a = 2.0 * M_PI * m / (double)r->PF;
            tw[m] = make_float2((float)cos(a), (float)-sin(a));
        }
        THX_CHECK(hipMalloc(reinterpret_cast<void**>(&r->tw), r->PF * sizeof(float2)));
        THX_CHECK(hipMemcpy(r->tw, tw.data(), r->PF * sizeof(float2), hipMemcpyHostToDevice));
    }
    THX_FFT_CHECK(plan_padded(&r->r2cF, r->PF, HIPFFT_R2C));
    THX_FFT_CHECK(plan_padded(&r->c2rF, r->PF, HIPFFT_C2R));
    r->haveN = r->PN != r->PF;
    if (r->haveN) THX_FFT_CHECK(plan_padded(&r->c2rN, r->PN, HIPFFT_C2R));
    else r->c2rN = r->c2rF;
    THX_FFT_CHECK(hipfftPlan3d(&r->r2cStd, r->PN, r->PN, r->PN, HIPFFT_R2C));
    return 0;
}

// T = max(T, 1e-25) (src/Reconstructor.cpp:1322-1324) on its own: what a reconstruction leaves behind in the caller's T.  A rank
// that runs only the SECOND reconstruction of an iteration (MAP on) on its own copy of T applies it first, so that its T goes
// through exactly the modifications the reference's one T goes through (floor, Wiener term, floor).  r->W is overwritten.
int thx_reco_floor_T_dev(thx_reco* r, float* T, int maxRadius, void* stream)
{
    THX_REQUIRE(r && T, "NULL pointer");
    const size_t nHalfF = (size_t)r->PF * r->PF * (r->PF / 2 + 1);
    hipLaunchKernelGGL(k_initW_floorT, dim3(nblk(nHalfF)), dim3(256), 0, as_stream(stream), r->W, T, r->PF, r->pf, maxRadius);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_reco_set_balance_rounds(thx_reco* r, int maxIter, int minIter)
{
    THX_REQUIRE(r && maxIter >= 0 && maxIter <= 1000 && minIter >= 0, "bad arguments");
    r->maxIter = maxIter; r->minIter = minIter;
    return 0;
}

int thx_reco_destroy(thx_reco* r)
{
    if (!r) return 0;
    if (r->r2cF) (void)hipfftDestroy(r->r2cF);
    if (r->c2rF) (void)hipfftDestroy(r->c2rF);
    if (r->r2cStd) (void)hipfftDestroy(r->r2cStd);
    if (r->haveN && r->c2rN) (void)hipfftDestroy(r->c2rN);
    (void)hipFree(r->tw); (void)hipFree(r->tab); (void)hipFree(r->W); (void)hipFree(r->C); (void)hipFree(r->rl); (void)hipFree(r->diff); (void)hipFree(r->stop); (void)hipFree(r->fscDev);
    delete r;
    return 0;
}

}  // extern "C"

// The same iteration with the hand-written passes of thx_fft8.h (PF = R 8^NS: 64 ... 1024, power-of-two N pf; 2048 stays on rocFFT: untested): per round
// y inverse -> fused x (inverse, kernel multiply, forward) -> y forward -> fused z (forward, W update + checkC, C = T W,
// inverse of the next round).  C stays in its padded half-complex grid; the real grid is never materialised.
template <int NS, int R, bool TILED, int WPS>
static int balance_W_hand_t(thx_reco* r, const float* T, int maxRadius, int maxIter, int minIter, int* itersOut,
                            float* diffCOut, void* resultDev, hipStream_t st)
{
    constexpr int P = f8_n<NS, R>(), NT8 = P / 8, nc = P / 2 + 1;
    constexpr int TXZ = NT8 * 16 <= 1024 ? 16 : (NT8 * 8 <= 1024 ? 8 : 4), TXY = NT8 * 8 <= 1024 ? 8 : 4;
    const int pf = r->pf, ncp = padded_nc(P);
    const size_t ldsZ = (size_t)(f8_rows<NS, R>() * TXZ + P) * sizeof(float2);
    const size_t ldsY = (size_t)(f8_rows<NS, R>() * TXY + P) * sizeof(float2);
    const size_t ldsX = (size_t)(f8_rows<NS, R>() * 5 + P) * sizeof(float2) + (P / 2 + 1) * sizeof(float);
    constexpr int nTx = (nc + TXZ - 1) / TXZ;
    const size_t ldsT = (size_t)16 * nTx * TXZ * sizeof(float);
    // WPS = 16 selects the sixteen-points-per-thread form of the z pass (512-thread workgroups, two per CU; P = 1024 only)
    constexpr bool X2 = WPS == 16;
    static_assert(!X2 || (NS == 3 && R == 2), "the sixteen-points-per-thread z pass exists for P = 1024");
    auto zfn = [](bool first) -> const void* {
        if constexpr (X2)
            return first ? reinterpret_cast<const void*>(k_fft_z_update_x2<TXZ, true, TILED>) : reinterpret_cast<const void*>(k_fft_z_update_x2<TXZ, false, TILED>);
        else
            return first ? reinterpret_cast<const void*>(k_fft_z_update<NS, R, TXZ, true, TILED, WPS>)
                         : reinterpret_cast<const void*>(k_fft_z_update<NS, R, TXZ, false, TILED, WPS>);
    };
    static std::once_flag once;
    static hipError_t attrErr = hipSuccess;
    std::call_once(once, [&]() {
        const void* fn[6] = {zfn(true), zfn(false),
                             reinterpret_cast<const void*>(k_fft_strided<NS, R, TXY, 1>),
                             reinterpret_cast<const void*>(k_fft_strided<NS, R, TXY, -1>),
                             reinterpret_cast<const void*>(k_fft_x_conv<NS, R>),
                             reinterpret_cast<const void*>(k_tile_real<TXZ>)};
        const size_t sz[6] = {ldsZ, ldsZ, ldsY, ldsY, ldsX, ldsT};
        for (int i = 0; i < 6 && attrErr == hipSuccess; i++)
            attrErr = hipFuncSetAttribute(fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)sz[i]);
    });
    THX_CHECK(attrErr);
    const dim3 gZ(nTx, P), bZ(X2 ? 64 * TXZ : NT8 * TXZ), gY((nc + TXY - 1) / TXY, P), bY(NT8 * TXY);
    const int r2i = maxRadius * pf;
    int iters = 0, nNoDec = 0;
    float diffC = 3.402823466e+38f, diffCPrev = 3.402823466e+38f;
    // (An error return from inside the loop leaves r->W in the tiled layout.  That is harmless by construction, not by discipline:
    // every reconstruction starts with k_initW_floorT writing ALL of W in the natural layout before this function is entered, and the
    // only reader of W after this function -- k_FW, thx_ExposeWT_host's copy -- runs on the success path alone.)
    // TILED: W moves into the z pass's layout for the duration of the loop (through the real-space scratch, which the
    // hand-written loop does not use), T's tiled copy stays in that scratch; W comes back in the volume's layout at the end
    const size_t nTiled = (size_t)P * P * nTx * TXZ, nNat = (size_t)P * P * nc;
    float* Wz = r->W;
    const float* Tz = T;
    if (TILED) {
        const dim3 gT(P / 16, P);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_real<TXZ>), gT, dim3(256), ldsT, st, r->rl, r->W, P, nc, nTx, 1);
        THX_CHECK(hipMemcpyAsync(r->W, r->rl, nTiled * sizeof(float), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_real<TXZ>), gT, dim3(256), ldsT, st, r->rl, T, P, nc, nTx, 1);
        Tz = r->rl;
    }
    // stop rule: on the device (default; the host queues every round and reads nothing back) or, THX_RECO_STOP=host, one
    // 4-byte read-back per round as in rounds 1-4 (A/B; the two give the same W bit for bit)
    const bool devStop = !knobs().recoHostStop;
    RecoStop* stop = reinterpret_cast<RecoStop*>(r->stop);
    const int* flag = devStop ? &stop->done : nullptr;
    if (devStop) hipLaunchKernelGGL(k_reco_stop_init, dim3(1), dim3(1), 0, st, stop);
    auto launchZ = [&](auto firstTag, const int* flg) {
        constexpr bool FIRST = decltype(firstTag)::value;
        if constexpr (X2)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fft_z_update_x2<TXZ, FIRST, TILED>), gZ, bZ, ldsZ, st, r->C, Wz, Tz, ncp, r2i, r->diff, r->tw, flg);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fft_z_update<NS, R, TXZ, FIRST, TILED, WPS>), gZ, bZ, ldsZ, st, r->C, Wz, Tz, ncp, r2i, r->diff, r->tw, flg);
    };
    launchZ(std::true_type{}, (const int*)nullptr);
    for (int m = 0; m < maxIter; m++) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fft_strided<NS, R, TXY, 1>), gY, bY, ldsY, st, r->C, (long)ncp, (long)P * ncp, nc, r->tw, flag);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fft_x_conv<NS, R>), dim3(P / 2 + 1, P / 2 + 1), dim3(NT8 * 4), ldsX, st, r->C, ncp,
                           r->N * pf, r->tab, kTabN, r->nf, r->rnf, r->rs, r->tw, flag);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fft_strided<NS, R, TXY, -1>), gY, bY, ldsY, st, r->C, (long)ncp, (long)P * ncp, nc, r->tw, flag);
        THX_CHECK(hipMemsetAsync(r->diff, 0, sizeof(unsigned), st));
        launchZ(std::false_type{}, flag);
        if (devStop) {
            hipLaunchKernelGGL(k_reco_stop_rule, dim3(1), dim3(1), 0, st, stop, r->diff, m, minIter);
            // a queued round that falls through still costs its four launches (~ 60 us of empty 8 000-workgroup grids at 512^3):
            // from the first round the rule can fire in (no-decrease twice: m >= minIter), every fourth round the host looks
            if (m >= minIter && ((m - minIter) & 3) == 0 && m + 1 < maxIter) {
                int done = 0;
                THX_CHECK(hipMemcpyAsync(&done, &stop->done, sizeof(int), hipMemcpyDeviceToHost, st));
                THX_CHECK(hipStreamSynchronize(st));
                if (done) break;
            }
            continue;
        }
        unsigned bits = 0;
        THX_CHECK(hipMemcpyAsync(&bits, r->diff, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        THX_CHECK(hipStreamSynchronize(st));
        diffCPrev = diffC;
        memcpy(&diffC, &bits, sizeof(float));
        iters = m + 1;
        if (knobs().recoTrace) fprintf(stderr, "hand round %d diffC %g\n", m, diffC);
        if ((double)diffC > (double)diffCPrev * 0.95) nNoDec += 1; else nNoDec = 0;   // as balance_W below
        if (((double)diffC < 1e-2) || ((m >= minIter) && (nNoDec == 2))) break;
    }
    if (TILED) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_real<TXZ>), dim3(P / 16, P), dim3(256), ldsT, st, r->rl, r->W, P, nc, nTx, 0);
        THX_CHECK(hipMemcpyAsync(r->W, r->rl, nNat * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    THX_LAUNCH_CHECK();
    if (devStop) {
        if (resultDev) {   // the caller collects (iters, diffC) later: nothing is read back here
            THX_CHECK(hipMemcpyAsync(resultDev, r->stop, sizeof(RecoStop), hipMemcpyDefault, st));
            iters = -1;
        } else {
            RecoStop hs;
            THX_CHECK(hipMemcpyAsync(&hs, r->stop, sizeof(RecoStop), hipMemcpyDeviceToHost, st));
            THX_CHECK(hipStreamSynchronize(st));
            iters = hs.iters; diffC = hs.diffC;
        }
    } else if (resultDev) {
        RecoStop hs = {1, iters, nNoDec, diffC, diffCPrev};
        THX_CHECK(hipMemcpyAsync(resultDev, &hs, sizeof(RecoStop), hipMemcpyDefault, st));
        THX_CHECK(hipStreamSynchronize(st));   // (hs is a stack temporary)
    }
    *itersOut = iters;
    *diffCOut = diffC;
    return 0;
}

constexpr int kFftzDefault1024 = 16;   // z pass at P = 1024: 16 = sixteen points per thread in 512-thread workgroups, two per CU (13.25 ms per round; bit-identical); 4 = eight points per thread in 1 024-thread workgroups, one per CU (14.06 ms); THX_FFTZ_WAVES overrides

template <int NS, int R>
static int balance_W_hand(thx_reco* r, const float* T, int maxRadius, int maxIter, int minIter, int* itersOut, float* diffCOut,
                          void* resultDev, hipStream_t st)
{
    // THX_FFTZ_WAVES = 4 / 8 (A/B); default: 4 for the 1024-point instance (no spills), 8 otherwise
    // THX_FFTZ_WAVES = 16: the sixteen-points-per-thread form (P = 1024 only)
    const int wps = knobs().fftzWaves > 0 ? knobs().fftzWaves : ((NS == 3 && R == 2) ? kFftzDefault1024 : 8);
    const bool nat = knobs().recoNatural;
#define THX_BW(tiled, w) return balance_W_hand_t<NS, R, tiled, w>(r, T, maxRadius, maxIter, minIter, itersOut, diffCOut, resultDev, st)
    if constexpr (NS == 3 && R == 2) {
        if (wps == 16) { if (nat) THX_BW(false, 16); THX_BW(true, 16); }
    }
    if (wps <= 4) { if (nat) THX_BW(false, 4); THX_BW(true, 4); }
    if (nat) THX_BW(false, 8);
    THX_BW(true, 8);
#undef THX_BW
}

extern "C" {

// The gridding-weight iteration of Reconstructor::reconstruct (src/Reconstructor.cpp:1379-1551): W (r->W) must hold the
// initial weights, T the floored T; the device-resident loop of C2R -> kernel multiply -> R2C -> W update + checkC.
static int balance_W(thx_reco* r, const float* T, int maxRadius, int maxIter, int minIter, int* itersOut, float* diffCOut,
                     void* resultDev, hipStream_t st)
{
    const int PF = r->PF, pf = r->pf, ncpF = padded_nc(PF);
    int iters = 0, nNoDec = 0;
    float diffC = 3.402823466e+38f, diffCPrev = 3.402823466e+38f;
    const long np = (long)r->N * pf;
    const bool pow2 = ((np & (np - 1)) == 0) && ((PF & (PF - 1)) == 0);
    {
        // THX_FFT=rocfft (read once at load): library transforms for every size (A/B and fallback)
        if (r->handNS && pow2 && !knobs().fftRocfft) {
#define THX_HAND(ns, rr) return balance_W_hand<ns, rr>(r, T, maxRadius, maxIter, minIter, itersOut, diffCOut, resultDev, st)
            switch (PF) {
                case 64: THX_HAND(2, 1);
                case 128: THX_HAND(2, 2);
                case 256: THX_HAND(2, 4);
                case 512: THX_HAND(3, 1);
                case 1024: THX_HAND(3, 2);
                default: break;
            }
#undef THX_HAND
        }
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_updateW_calcC<false>), dim3(PF, PF), dim3(256), 0, st, r->W, r->C, T, PF, ncpF, pf,
                       maxRadius, r->diff);
    for (int m = 0; m < maxIter; m++) {
        THX_FFT_CHECK(hipfftExecC2R(r->c2rF, reinterpret_cast<hipfftComplex*>(r->C), r->rl));
        if (pow2)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_convolute_rl<true>), dim3(PF / 2 + 1, PF / 2 + 1), dim3(256), 0, st, r->rl,
                               PF, r->N * pf, r->tab, r->nf, r->rnf, r->rs, 1);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_convolute_rl<false>), dim3(PF / 2 + 1, PF / 2 + 1), dim3(256), 0, st, r->rl,
                               PF, r->N * pf, r->tab, r->nf, r->rnf, r->rs, 1);
        THX_FFT_CHECK(hipfftExecR2C(r->r2cF, r->rl, reinterpret_cast<hipfftComplex*>(r->C)));
        THX_CHECK(hipMemsetAsync(r->diff, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_updateW_calcC<true>), dim3(PF, PF), dim3(256), 0, st, r->W, r->C, T, PF, ncpF, pf,
                           maxRadius, r->diff);
        unsigned bits = 0;
        THX_CHECK(hipMemcpyAsync(&bits, r->diff, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        THX_CHECK(hipStreamSynchronize(st));
        diffCPrev = diffC;
        memcpy(&diffC, &bits, sizeof(float));
        iters = m + 1;
        if (knobs().recoTrace) fprintf(stderr, "rocfft round %d diffC %g\n", m, diffC);
        // src/Reconstructor.cpp:1542-1550 (DIFF_C_DECREASE_THRES 0.95, DIFF_C_THRES 1e-2, N_DIFF_C_NO_DECREASE 2); the
        // comparisons run in double against RFLOAT operands as in the reference
        if ((double)diffC > (double)diffCPrev * 0.95) nNoDec += 1; else nNoDec = 0;
        if (((double)diffC < 1e-2) || ((m >= minIter) && (nNoDec == 2))) break;
    }
    THX_LAUNCH_CHECK();
    if (resultDev) {   // (library-transform path: the host drives the loop and knows the result)
        RecoStop hs = {1, iters, nNoDec, diffC, diffCPrev};
        THX_CHECK(hipMemcpyAsync(resultDev, &hs, sizeof(RecoStop), hipMemcpyDefault, st));
        THX_CHECK(hipStreamSynchronize(st));
    }
    *itersOut = iters;
    *diffCOut = diffC;
    return 0;
}

static int reconstruct_impl(thx_reco* r, const float* F, float* T, int maxRadius, const float* FSC_host, int nFSC, int joinHalf, int MAP,
                            int gridCorr, float* dstRL, int* nIterOut, float* diffCOut, void* resultDev, void* stream);

int thx_reco_reconstruct_dev(thx_reco* r, const float* F, float* T, int maxRadius, const float* FSC_host, int nFSC,
                             int joinHalf, int MAP, int gridCorr, float* dstRL, int* nIterOut, float* diffCOut,
                             void* stream)
{
    return reconstruct_impl(r, F, T, maxRadius, FSC_host, nFSC, joinHalf, MAP, gridCorr, dstRL, nIterOut, diffCOut, nullptr, stream);
}

// the same without a host synchronisation: result [5 words] (DEVICE or page-locked HOST memory) receives {done, rounds, -, diffC
// (float bits), -} of the gridding loop when the stream gets there -- the iteration driver queues its 2 K reconstructions per half
// back to back and reads the round counts once at the end
int thx_reco_reconstruct_async_dev(thx_reco* r, const float* F, float* T, int maxRadius, const float* FSC_host, int nFSC,
                                   int joinHalf, int MAP, int gridCorr, float* dstRL, void* result, void* stream)
{
    THX_REQUIRE(result, "result is NULL");
    if (!gridCorr) THX_CHECK(hipMemsetAsync(result, 0, sizeof(RecoStop), as_stream(stream)));
    return reconstruct_impl(r, F, T, maxRadius, FSC_host, nFSC, joinHalf, MAP, gridCorr, dstRL, nullptr, nullptr, result, stream);
}

static int reconstruct_impl(thx_reco* r, const float* F, float* T, int maxRadius, const float* FSC_host, int nFSC, int joinHalf, int MAP,
                            int gridCorr, float* dstRL, int* nIterOut, float* diffCOut, void* resultDev, void* stream)
{
    THX_REQUIRE(r && F && T && dstRL, "NULL pointer");
    THX_REQUIRE(!MAP || (FSC_host && nFSC > 0 && nFSC <= 4096), "MAP needs an FSC vector (<= 4096 shells)");
    THX_REQUIRE(maxRadius * r->pf < r->PF / 2 - 1, "maxRadius too large for the reconstruction grid");
    hipStream_t st = as_stream(stream);
    const int PF = r->PF, PN = r->PN, pf = r->pf;
    const size_t nHalfF = (size_t)PF * PF * (PF / 2 + 1);
    const size_t nHalfN = (size_t)PN * PN * (PN / 2 + 1);
    THX_FFT_CHECK(hipfftSetStream(r->r2cF, st));
    THX_FFT_CHECK(hipfftSetStream(r->c2rF, st));
    if (r->haveN) THX_FFT_CHECK(hipfftSetStream(r->c2rN, st));
    const int ncpN = padded_nc(PN);
    if (MAP) {
        THX_CHECK(hipMemcpyAsync(r->fscDev, FSC_host, nFSC * sizeof(float), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_wiener_T, dim3(nblk(nHalfF)), dim3(256), 0, st, T, PF, pf, maxRadius, r->fscDev, nFSC,
                           joinHalf, 5);  // WIENER_FACTOR_MIN_R
    }
    hipLaunchKernelGGL(k_initW_floorT, dim3(nblk(nHalfF)), dim3(256), 0, st, r->W, T, PF, pf, maxRadius);
    int iters = 0;
    float diffC = 3.402823466e+38f;
    if (gridCorr) {
        THX_RC(balance_W(r, T, maxRadius, r->maxIter, r->minIter, &iters, &diffC, resultDev, st));  // MAX_N_ITER_BALANCE, MIN_N_ITER_BALANCE
    } else {
        hipLaunchKernelGGL(k_W_nogridcorr, dim3(nblk(nHalfF)), dim3(256), 0, st, r->W, T, PF, pf, maxRadius);
    }
    hipLaunchKernelGGL(k_FW, dim3(nblk(nHalfN)), dim3(256), 0, st, r->C, PN, ncpN, reinterpret_cast<const float2*>(F), r->W, PF,
                       pf, maxRadius);
    THX_FFT_CHECK(hipfftExecC2R(r->c2rN, reinterpret_cast<hipfftComplex*>(r->C), r->rl));
    hipLaunchKernelGGL(k_extract_tik, dim3(nblk((size_t)r->N * r->N * r->N)), dim3(256), 0, st, dstRL, r->rl, PN, r->N, pf, 1);
    THX_LAUNCH_CHECK();
    if (nIterOut) *nIterOut = iters;
    if (diffCOut) *diffCOut = diffC;
    return 0;
}

int thx_reco_set_projectee_dev(thx_reco* r, const float* refRL, float* volume, void* stream)
{
    THX_REQUIRE(r && refRL && volume, "NULL pointer");
    hipStream_t st = as_stream(stream);
    const int PN = r->PN;
    THX_FFT_CHECK(hipfftSetStream(r->r2cStd, st));
    hipLaunchKernelGGL(k_pad_gridcorr, dim3(nblk((size_t)PN * PN * PN)), dim3(256), 0, st, r->rl, refRL, r->N, r->pf);
    THX_LAUNCH_CHECK();
    THX_FFT_CHECK(hipfftExecR2C(r->r2cStd, r->rl, reinterpret_cast<hipfftComplex*>(volume)));
    return 0;
}

// FFT plans of the small N^3 transforms are cached per (device, stream, size, direction): plan creation costs far more
// than the transform (the reference's FFT::fw re-plans every call with FFTW_ESTIMATE, src/FFT.cpp:176-199).  One plan per
// stream: a hipFFT plan carries its stream and work area, so two host threads on two streams must not share one.
static std::mutex g_plan3dMtx;
static std::map<std::tuple<int, hipStream_t, int, int>, hipfftHandle> g_plan3d;
void thx_release_reco_plans_(int dev, hipStream_t st)   /* internal (called by thx_release_stream) */
{
    std::lock_guard<std::mutex> g(g_plan3dMtx);
    for (auto it = g_plan3d.begin(); it != g_plan3d.end();)
        if (std::get<0>(it->first) == dev && std::get<1>(it->first) == st) { (void)hipfftDestroy(it->second); it = g_plan3d.erase(it); } else ++it;
}
static int cached_plan(hipfftHandle* out, int n, hipfftType type, hipStream_t st)
{
    std::mutex& mtx = g_plan3dMtx;
    auto& cache = g_plan3d;
    int dev = 0;
    THX_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mtx);
    auto key = std::make_tuple(dev, st, n, (int)type);
    auto it = cache.find(key);
    if (it == cache.end()) {
        hipfftHandle p;
        THX_FFT_CHECK(hipfftPlan3d(&p, n, n, n, type));
        THX_FFT_CHECK(hipfftSetStream(p, st));
        it = cache.emplace(key, p).first;
    }
    *out = it->second;
    return 0;
}

int thx_fft3d_fw_dev(const float* rl, float* ft, int n, void* stream)
{
    THX_REQUIRE(rl && ft, "NULL pointer");
    hipfftHandle p;
    int rc = cached_plan(&p, n, HIPFFT_R2C, as_stream(stream));
    if (rc) return rc;
    THX_FFT_CHECK(hipfftExecR2C(p, const_cast<float*>(rl), reinterpret_cast<hipfftComplex*>(ft)));
    return 0;
}

int thx_fft3d_bw_dev(float* ft, float* rl, int n, void* stream)
{
    THX_REQUIRE(rl && ft, "NULL pointer");
    hipfftHandle p;
    int rc = cached_plan(&p, n, HIPFFT_C2R, as_stream(stream));
    if (rc) return rc;
    THX_FFT_CHECK(hipfftExecC2R(p, reinterpret_cast<hipfftComplex*>(ft), rl));
    hipLaunchKernelGGL(k_scale_rl, dim3(nblk((size_t)n * n * n)), dim3(256), 0, as_stream(stream), rl, (size_t)n * n * n);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_fsc_dev(float* fsc, int nShell, const float* A, const float* B, int dim, void* stream)
{
    THX_REQUIRE(fsc && A && B && nShell > 0 && nShell <= 4096, "bad arguments");
    hipStream_t st = as_stream(stream);
    double* acc = nullptr;
    const int nW = kFscBlocks * 4;
    acc = reinterpret_cast<double*>(scratch(st, 2, (size_t)(nW + 1) * 3 * nShell * sizeof(double)));
    THX_REQUIRE(acc, "device scratch allocation failed");
    double* part = acc + 3 * nShell;
    THX_REQUIRE(4 * 3 * (size_t)nShell * sizeof(double) <= 64 * 1024, "too many shells for the per-wave tables");
    hipLaunchKernelGGL(k_fsc_rows, dim3(kFscBlocks), dim3(256), 4 * 3 * (size_t)nShell * sizeof(double), st, part, nShell,
                       reinterpret_cast<const float2*>(A), reinterpret_cast<const float2*>(B), dim);
    hipLaunchKernelGGL(k_fsc_reduce, dim3((3 * nShell + 255) / 256), dim3(256), 0, st, acc, part, 3 * nShell, nW);
    hipLaunchKernelGGL(k_fsc_final, dim3((nShell + 255) / 256), dim3(256), 0, st, fsc, acc, nShell);
    THX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Staged form of Reconstructor::reconstructG (src/Reconstructor.cpp:1835-2330): the reference strings these calls
// together with host FFTs in between; each entry keeps that contract on caller-owned host arrays.  The arithmetic is that
// of the CPU statement (Reconstructor::reconstruct), i.e. of thx_reco_reconstruct_dev, stage by stage.
// ---------------------------------------------------------------------------------------------
static inline size_t half_grid(int dim) { return (size_t)dim * dim * (dim / 2 + 1); }

int thx_ExposePT_host(int gpuIdx, float* T3D, int maxRadius, int pf, int dim, const float* FSC, int nFSC, int joinHalf,
                      int wienerF)
{
    THX_REQUIRE(T3D && FSC && nFSC > 0 && dim > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    DevBuf dT, dF;
    THX_RC(dT.upload(T3D, half_grid(dim) * sizeof(float)));
    THX_RC(dF.upload(FSC, nFSC * sizeof(float)));
    hipLaunchKernelGGL(k_wiener_T, dim3(nblk(half_grid(dim))), dim3(256), 0, nullptr, dT.as<float>(), dim, pf, maxRadius,
                       dF.as<float>(), nFSC, joinHalf, wienerF);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpy(T3D, dT.p, half_grid(dim) * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int thx_ExposeWT_host(int gpuIdx, const float* T3D, float* W3D, const float* tab, int tabSize, float nf, int maxRadius,
                      int pf, int dim, int maxIter, int minIter, int size)
{
    THX_REQUIRE(T3D && W3D && tab && tabSize > 0 && tabSize <= kTabN + 1 && dim > 0 && pf > 0 && dim % pf == 0,
                "bad arguments (the kernel table has at most 1e5 + 1 entries)");
    THX_CHECK(hipSetDevice(gpuIdx));
    thx_reco* r = nullptr;
    THX_RC(thx_reco_create(&r, dim / pf, size, pf, 1.9f, 15.f));
    // the caller's tabulated kernel and normalisation replace the ones thx_reco_create derived from (a, alpha)
    int rc = 0;
    DevBuf dT;
    hipError_t e = hipMemset(r->tab, 0, (kTabN + 1) * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(r->tab, tab, tabSize * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("table upload failed: %s", hipGetErrorString(e)); rc = (int)e; }
    r->nf = nf;
    r->rnf = (float)(1.0 / (double)nf);
    if (!rc) rc = dT.upload(T3D, half_grid(dim) * sizeof(float));
    if (!rc) {
        hipLaunchKernelGGL(k_initW_floorT, dim3(nblk(half_grid(dim))), dim3(256), 0, nullptr, r->W, dT.as<float>(), dim, pf,
                           maxRadius);
        int iters = 0;
        float diffC = 0.f;
        (void)hipfftSetStream(r->r2cF, nullptr);
        (void)hipfftSetStream(r->c2rF, nullptr);
        rc = balance_W(r, dT.as<float>(), maxRadius, maxIter, minIter, &iters, &diffC, nullptr, nullptr);
    }
    if (!rc) {
        e = hipMemcpy(W3D, r->W, half_grid(dim) * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("hipMemcpy D2H failed: %s", hipGetErrorString(e)); rc = (int)e; }
    }
    thx_reco_destroy(r);
    return rc;
}

int thx_ExposeWT_plain_host(int gpuIdx, const float* T3D, float* W3D, int maxRadius, int pf, int dim)
{
    THX_REQUIRE(T3D && W3D && dim > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    DevBuf dT, dW;
    THX_RC(dT.upload(T3D, half_grid(dim) * sizeof(float)));
    THX_RC(dW.alloc(half_grid(dim) * sizeof(float)));
    hipLaunchKernelGGL(k_initW_floorT, dim3(nblk(half_grid(dim))), dim3(256), 0, nullptr, dW.as<float>(), dT.as<float>(), dim,
                       pf, maxRadius);
    hipLaunchKernelGGL(k_W_nogridcorr, dim3(nblk(half_grid(dim))), dim3(256), 0, nullptr, dW.as<float>(), dT.as<float>(), dim,
                       pf, maxRadius);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpy(W3D, dW.p, half_grid(dim) * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int thx_AllocDevicePoint_host(int gpuIdx, float** dev_C, float** dev_W, float** dev_T, float** dev_tab, float** devDiff,
                              float** devMax, int** devCount, void** stream, int streamNum, int tabSize, int dim)
{
    THX_REQUIRE(dev_C && dev_W && dev_T && dev_tab && devMax && stream && streamNum >= 1 && dim > 0 && tabSize > 0,
                "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    const size_t n = half_grid(dim);
    const size_t nTab = (size_t)(tabSize > kTabN + 1 ? tabSize : kTabN + 1);
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(dev_C), n * 2 * sizeof(float)));  // also holds the dim^3 real grid
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(dev_W), n * sizeof(float)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(dev_T), n * sizeof(float)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(dev_tab), nTab * sizeof(float)));
    THX_CHECK(hipMemset(*dev_tab, 0, nTab * sizeof(float)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(devMax), (size_t)dim * sizeof(float)));  // RECONSTRUCTOR_CHECK_C_MAX
    if (devDiff) *devDiff = nullptr;   // RECONSTRUCTOR_CHECK_C_AVERAGE is off (include/Config.h:101)
    if (devCount) *devCount = nullptr;
    for (int i = 0; i < streamNum; i++) {
        hipStream_t st;
        THX_CHECK(hipStreamCreate(&st));
        stream[i] = reinterpret_cast<void*>(st);
    }
    return 0;
}

int thx_HostDeviceInit_host(int gpuIdx, const float* T3D, const float* tab, float* dev_W, float* dev_T, float* dev_tab,
                            void** stream, int streamNum, int tabSize, int maxRadius, int pf, int dim)
{
    THX_REQUIRE(T3D && tab && dev_W && dev_T && dev_tab && stream && streamNum >= 1, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    hipStream_t st = as_stream(stream[0]);
    THX_CHECK(hipMemcpyAsync(dev_tab, tab, (size_t)tabSize * sizeof(float), hipMemcpyHostToDevice, st));
    THX_CHECK(hipMemcpyAsync(dev_T, T3D, half_grid(dim) * sizeof(float), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_initW_floorT, dim3(nblk(half_grid(dim))), dim3(256), 0, st, dev_W, dev_T, dim, pf, maxRadius);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipStreamSynchronize(st));
    return 0;
}

int thx_ExposeC_host(int gpuIdx, float* C3D, float* dev_C, const float* dev_T, float* dev_W, void** stream, int streamNum,
                     int dim)
{
    THX_REQUIRE(C3D && dev_C && dev_T && dev_W && stream && streamNum >= 1, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    hipStream_t st = as_stream(stream[0]);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_updateW_calcC<false>), dim3(dim, dim), dim3(256), 0, st, dev_W,
                       reinterpret_cast<float2*>(dev_C), dev_T, dim, dim / 2 + 1, 1, 0, nullptr);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpyAsync(C3D, dev_C, half_grid(dim) * 2 * sizeof(float), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipStreamSynchronize(st));
    return 0;
}

int thx_ExposeForConvC_host(int gpuIdx, float* C3D_rl, float* dev_C, const float* dev_tab, void** stream, float step,
                            float nf, int streamNum, int tabSize, int pf, int size, int dim)
{
    THX_REQUIRE(C3D_rl && dev_C && dev_tab && stream && streamNum >= 1, "bad arguments");
    THX_REQUIRE(dim % 4 == 0 && dim <= 2 * kMaxHalfP, "dim must be a multiple of 4, at most 2048");
    THX_REQUIRE(step == 1.0f / kTabN && tabSize <= kTabN + 1, "the tabulated kernel must be the 1e5-step table on [0, 1]");
    THX_CHECK(hipSetDevice(gpuIdx));
    hipStream_t st = as_stream(stream[0]);
    const size_t n = (size_t)dim * dim * dim;
    THX_CHECK(hipMemcpyAsync(dev_C, C3D_rl, n * sizeof(float), hipMemcpyHostToDevice, st));
    const long np = (long)size * pf;
    const bool pow2 = ((np & (np - 1)) == 0) && ((dim & (dim - 1)) == 0);
    const float rnf = (float)(1.0 / (double)nf), rs = (float)(1.0 / (double)(1.0f / kTabN));
    if (pow2)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_convolute_rl<true>), dim3(dim / 2 + 1, dim / 2 + 1), dim3(256), 0, st, dev_C, dim,
                           size * pf, dev_tab, nf, rnf, rs, 0);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_convolute_rl<false>), dim3(dim / 2 + 1, dim / 2 + 1), dim3(256), 0, st, dev_C, dim,
                           size * pf, dev_tab, nf, rnf, rs, 0);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpyAsync(C3D_rl, dev_C, n * sizeof(float), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipStreamSynchronize(st));
    return 0;
}

int thx_ExposeWC_host(int gpuIdx, const float* C3D, float* dev_C, float* cmax, float* dev_W, float* devMax, void** stream,
                      float* diffC, int streamNum, int maxRadius, int pf, int dim)
{
    THX_REQUIRE(C3D && dev_C && dev_W && devMax && stream && diffC && streamNum >= 1, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    hipStream_t st = as_stream(stream[0]);
    THX_CHECK(hipMemcpyAsync(dev_C, C3D, half_grid(dim) * 2 * sizeof(float), hipMemcpyHostToDevice, st));
    THX_CHECK(hipMemsetAsync(devMax, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(k_recalcW_max, dim3(dim, dim), dim3(256), 0, st, dev_W, reinterpret_cast<const float2*>(dev_C), dim,
                       maxRadius * pf, reinterpret_cast<unsigned*>(devMax));
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpyAsync(diffC, devMax, sizeof(float), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipStreamSynchronize(st));
    if (cmax) cmax[0] = *diffC;   // the reference leaves per-slice maxima here; their maximum is diffC
    return 0;
}

int thx_FreeDevHostPoint_host(int gpuIdx, float** dev_C, float** dev_W, float** dev_T, float** dev_tab, float** devDiff,
                              float** devMax, int** devCount, void** stream, float* volumeW, int streamNum, int dim)
{
    THX_REQUIRE(dev_C && dev_W && dev_T && dev_tab && devMax && stream && volumeW, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    THX_CHECK(hipMemcpy(volumeW, *dev_W, half_grid(dim) * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < streamNum; i++) {
        THX_CHECK(hipStreamDestroy(as_stream(stream[i])));
        stream[i] = nullptr;
    }
    THX_CHECK(hipFree(*dev_C)); *dev_C = nullptr;
    THX_CHECK(hipFree(*dev_W)); *dev_W = nullptr;
    THX_CHECK(hipFree(*dev_T)); *dev_T = nullptr;
    THX_CHECK(hipFree(*dev_tab)); *dev_tab = nullptr;
    THX_CHECK(hipFree(*devMax)); *devMax = nullptr;
    if (devDiff && *devDiff) { THX_CHECK(hipFree(*devDiff)); *devDiff = nullptr; }
    if (devCount && *devCount) { THX_CHECK(hipFree(*devCount)); *devCount = nullptr; }
    return 0;
}

static int expose_pf(int gpuIdx, float* padDst, float* padDstR, const float* F3D, const float* W3D, int maxRadius, int pf,
                     int pdim, int fdim)
{
    THX_REQUIRE(F3D && W3D && pdim >= fdim && fdim > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    DevBuf dF, dW, dP, dR;
    THX_RC(dF.upload(F3D, half_grid(fdim) * 2 * sizeof(float)));
    THX_RC(dW.upload(W3D, half_grid(fdim) * sizeof(float)));
    THX_RC(dP.alloc(half_grid(pdim) * 2 * sizeof(float)));
    hipLaunchKernelGGL(k_FW, dim3(nblk(half_grid(pdim))), dim3(256), 0, nullptr, dP.as<float2>(), pdim, pdim / 2 + 1,
                       dF.as<float2>(), dW.as<float>(), fdim, pf, maxRadius);
    THX_LAUNCH_CHECK();
    if (padDst) THX_CHECK(hipMemcpy(padDst, dP.p, half_grid(pdim) * 2 * sizeof(float), hipMemcpyDeviceToHost));
    if (padDstR) {
        THX_RC(dR.alloc((size_t)pdim * pdim * pdim * sizeof(float)));
        THX_RC(thx_fft3d_bw_dev(dP.as<float>(), dR.as<float>(), pdim, nullptr));
        THX_CHECK(hipMemcpy(padDstR, dR.p, (size_t)pdim * pdim * pdim * sizeof(float), hipMemcpyDeviceToHost));
    }
    return 0;
}

int thx_ExposePFW_host(int gpuIdx, float* padDst, const float* F3D, const float* W3D, int maxRadius, int pf, int pdim,
                       int fdim)
{
    THX_REQUIRE(padDst, "padDst is NULL");
    return expose_pf(gpuIdx, padDst, nullptr, F3D, W3D, maxRadius, pf, pdim, fdim);
}

int thx_ExposePF_host(int gpuIdx, float* padDst, float* padDstR, const float* F3D, const float* W3D, int maxRadius, int pf,
                      int pdim, int fdim)
{
    THX_REQUIRE(padDstR, "padDstR is NULL");
    return expose_pf(gpuIdx, padDst, padDstR, F3D, W3D, maxRadius, pf, pdim, fdim);
}

int thx_ExposeCorrF_host(int gpuIdx, float* dst, const float* mkbRL, float nf, int dim)
{
    (void)nf;  // only the RECONSTRUCTOR_MKB_KERNEL build uses it; include/Config.h:95-97 selects the trilinear kernel
    THX_REQUIRE(dst && mkbRL && dim > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    const size_t n = (size_t)dim * dim * dim, h = dim / 2 + 1;
    DevBuf dD, dM;
    THX_RC(dD.upload(dst, n * sizeof(float)));
    THX_RC(dM.upload(mkbRL, h * h * h * sizeof(float)));
    hipLaunchKernelGGL(k_correctF, dim3(nblk(n)), dim3(256), 0, nullptr, dD.as<float>(), dM.as<float>(), dim);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpy(dst, dD.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int thx_ExposeCorrF_fft_host(int gpuIdx, const float* dstN, float* dstFT, const float* mkbRL, float nf, int dim)
{
    (void)nf;
    THX_REQUIRE(dstN && dstFT && mkbRL && dim > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    const size_t n = (size_t)dim * dim * dim, h = dim / 2 + 1;
    DevBuf dD, dM, dF;
    THX_RC(dD.upload(dstN, n * sizeof(float)));
    THX_RC(dM.upload(mkbRL, h * h * h * sizeof(float)));
    THX_RC(dF.alloc(half_grid(dim) * 2 * sizeof(float)));
    hipLaunchKernelGGL(k_correctF, dim3(nblk(n)), dim3(256), 0, nullptr, dD.as<float>(), dM.as<float>(), dim);
    THX_LAUNCH_CHECK();
    THX_RC(thx_fft3d_fw_dev(dD.as<float>(), dF.as<float>(), dim, nullptr));
    THX_CHECK(hipMemcpy(dstFT, dF.p, half_grid(dim) * 2 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
