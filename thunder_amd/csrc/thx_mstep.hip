// thx_mstep.hip -- M-step kernels: direct Fourier insertion (back-projection), T/F normalisation and
// symmetrisation.  Reference behaviour: src/Optimiser.cpp:7038-7241 (HOT LOOP C),
// src/Reconstructor.cpp:407-422,782-863,2455-2476,2676-2690, src/Image/Volume.cpp:340-375,565-712,
// include/Geometry/Transformation.h:105-131,170-194.  gfx950 only.
#include "thx_common.h"
#include "thx_insert.h"

namespace thx {

// ---------------------------------------------------------------------------------------------
// Insertion.  One thread per listed pixel of one image; the image row (dat, ctf) is read ONCE and
// kept in registers while the block walks all mReco draws (rotation, shift[, defocus]) of that image,
// whose parameters are wave-uniform scalar loads.  Each pixel-sample is a trilinear scatter of
// w*ctf*img into F (8 x complex) and w*ctf^2 into T (8 x real) with hardware fp32 atomics
// (global_atomic_add_f32): the MI355X counterpart of the reference's `#pragma omp atomic`
// (src/Image/Volume.cpp:584-587,676-677).  grid (ceil(nPxl/256), nImg).
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_insert(InsertArgs a)
{
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < a.nPxl;
    const size_t volSize = (size_t)a.P * a.P * (a.P / 2 + 1);
    int ic = 0, ir = 0;
    float2 dv = make_float2(0.f, 0.f);
    float cf = 0.f;
    if (valid) {
        ic = a.iCol[p];
        ir = a.iRow[p];
        dv = a.datP[(size_t)img * a.nPxl + p];
        cf = a.ctfP[(size_t)img * a.nPxl + p];
    }
    const int icp = ic * a.opf, irp = ir * a.opf;  // _iColPad / _iRowPad, src/Optimiser.cpp:8031-8033
    const float wgt = a.w[img];
    const double offx = a.offS ? a.offS[2 * img] : 0.0, offy = a.offS ? a.offS[2 * img + 1] : 0.0;
    double ox = 0, oy = 0, oz = 0;

    for (int m = 0; m < a.mReco; m++) {
        const size_t dm = (size_t)img * a.mReco + m;
        const double* R = a.rotMat + dm * 9;
        const double tx = a.trans[2 * dm] - offx, ty = a.trans[2 * dm + 1] - offy;  // (tran - _offset[l])
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            // dir = -rot3D * dvec3(tx, ty, 0), src/Optimiser.cpp:7226-7232
            ox += -(R[0] * tx + R[3] * ty);
            oy += -(R[1] * tx + R[4] * ty);
            oz += -(R[2] * tx + R[5] * ty);
        }
        if (!valid) continue;
        // translate(transImgP, orignImgP, -(tran - offset)(0), -(tran - offset)(1), ...), :7160-7169
        const float rCol = (float)(-tx) / a.idim, rRow = (float)(-ty) / a.idim;
        const float2 tv = cmul(dv, ramp_value(rCol, rRow, ic, ir));
        float c = cf;
        if (a.cSearch) {
            const CtfConst cc = ctf_const(a.attr[img], a.dfac[dm]);
            c = ctf_value(cc, a.pixelSize, a.idim, a.idim, ic, ir);
        }
        // src[i] * ctf[i] * 1 * w, left to right (src/Reconstructor.cpp:830-833)
        float vre = tv.x * c, vim = tv.y * c;
        vre = vre * 1.0f; vim = vim * 1.0f;
        vre = vre * wgt; vim = vim * wgt;
        const float tval = pow2f_(c) * 1.0f * wgt;
        const double cx = R[0] * icp + R[3] * irp;
        const double cy = R[1] * icp + R[4] * irp;
        const double cz = R[2] * icp + R[5] * irp;
        const float x = (float)cx, y = (float)cy, z = (float)cz;
        if (!coord_in_grid(x, y, z, a.P)) continue;
        TriCell cell;
        tri_cell(cell, x, y, z, a.P);
        if (cell.conj) vim = -vim;
        const int k = a.cls ? a.cls[dm] : 0;
        float2* F = a.F + (size_t)k * volSize;
        float* T = a.T + (size_t)k * volSize;
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    const float wv = cell.w[kk * 4 + jj * 2 + ii];
                    const long idx = cell.rowOff[kk][jj] + ii;
                    unsafeAtomicAdd(&F[idx].x, vre * wv);
                    unsafeAtomicAdd(&F[idx].y, vim * wv);
                    unsafeAtomicAdd(&T[idx], tval * wv);
                }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.O) {
        unsafeAtomicAdd(&a.O[0], ox);
        unsafeAtomicAdd(&a.O[1], oy);
        unsafeAtomicAdd(&a.O[2], oz);
        if (a.counter) atomicAdd(a.counter, a.mReco);
    }
}

// ---------------------------------------------------------------------------------------------
// Insertion through an LDS brick (the production path, k_insert_win below; k_insert above is the plain reference form).
//
// Measured on MI355X (tools/atomic_bench.hip, tools/lds_atomic_bench.hip): fp32 global atomics retire ~19 G
// *transactions*/s chip-wide (one per XCD per clock) and a transaction may carry up to 16 consecutive floats, so
// scattered 4-byte atomics (k_insert) reach 2 % of the HBM roofline while runs of 16 consecutive floats are 16x faster;
// ds_add_f32 retires 0.33 lanes/clk/CU whatever the address pattern, ds_add_u32 6-8.  Hence: accumulate in an LDS brick
// in 32-bit FIXED POINT (order-independent integer sums), flush the brick along the volume's contiguous x axis.
//
// Fixed point: T is unsigned with scale 2^(32 - ceil(log2(2 mReco))) / max(ctf^2 w), F signed with one bit less over
// max|dat| |ctf| w (for one rotation the trilinear weights a voxel collects sum to <= 1).  A term enters the brick only if
// its T part is >= kMinQuanta quanta (< 1 % rounding); smaller terms (trilinear weight ~1e-6, CTF zeros: ~1 % of terms)
// go to the volume as float atomics, F and T together -- so T can never round to zero where F does not (which would let
// the gridding weights W ~ 1 / (T*W conv K) explode).
//
// Hermitian fold in brick coordinates: a folded sample (x < 0 -> (X,Y,Z) = -(x,y,z), conjugated) addresses the brick
// at (-1-X, -Y, -Z); non-folded samples at (X, Y, Z).  The two half-spaces stay disjoint (X = 0 of a folded sample is
// brick x = -1, so F(0,j,k) and F(0,-j,-k) remain independent accumulators, SURVEY 8a note H) and adjacent.
// (A pixel-tile form of this kernel -- workgroup = (image, 8x8-pixel tile), every draw's copy of the tile in one brick --
// was 25 % faster for coincident draws and 5x slower for a particle filter's 1-degree clouds; removed, see DESIGN.md.)
// ---------------------------------------------------------------------------------------------
constexpr float kMinQuanta = 64.f;  // smallest T term (in fixed-point quanta) accumulated in the LDS brick
constexpr int kMaxU = 16;           // unique shifts whose ramps are tabulated in LDS (else computed per member)

// ---------------------------------------------------------------------------------------------
// Insert plan.  The mReco draws of one image come from a resampled particle filter (Particle::rand picks among
// resampled support points, src/Particle.cpp:2109-2178), so many draws share the same rotation and the shifts take
// only ~mLT distinct values.  Draws with bit-identical (rotation[, class][, defocus factor]) form a GROUP: their
// trilinear cell and weights are identical, so  sum_m (img * ramp_m) * ctf * w * wv  is inserted once as
// (img * sum_m ramp_m) * ctf * w * wv  (distributivity: results equal to rounding) and T gets n_g * ctf^2 * w * wv.
// Per image: [0] G, [1] U, gStart[mReco+1], ord[mReco] (draws sorted by group), uid[mReco] (unique-shift id per draw),
// gRep[mReco] (representative draw of group g), tRep[mReco] (representative draw of unique shift u).
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(128) void k_insert_plan(int* __restrict__ plan, const double* __restrict__ rotMat,
                                                     const double* __restrict__ trans, const int* __restrict__ cls,
                                                     const double* __restrict__ dfac, int cSearch, int mReco)
{
    extern __shared__ int sp[];   // first[mReco] | firstT[mReco] | gid[mReco] | uid[mReco] | cnt[mReco]
    int* first = sp;
    int* firstT = sp + mReco;
    int* gid = sp + 2 * mReco;
    int* uid = sp + 3 * mReco;
    int* cnt = sp + 4 * mReco;
    const int img = blockIdx.x, tid = threadIdx.x;
    const long long* R = reinterpret_cast<const long long*>(rotMat) + (size_t)img * mReco * 9;
    const long long* Tt = reinterpret_cast<const long long*>(trans) + (size_t)img * mReco * 2;
    const int* c = cls ? cls + (size_t)img * mReco : nullptr;
    const long long* D = (cSearch && dfac) ? reinterpret_cast<const long long*>(dfac) + (size_t)img * mReco : nullptr;
    for (int m = tid; m < mReco; m += blockDim.x) {
        int f = m, ft = m;
        for (int q = 0; q < m; q++) {
            bool same = true;
            for (int e = 0; e < 6; e++) same = same && (R[9 * q + e] == R[9 * m + e]);
            if (c) same = same && (c[q] == c[m]);
            if (D) same = same && (D[q] == D[m]);
            if (same) { f = q; break; }
        }
        for (int q = 0; q < m; q++)
            if (Tt[2 * q] == Tt[2 * m] && Tt[2 * q + 1] == Tt[2 * m + 1]) { ft = q; break; }
        first[m] = f;
        firstT[m] = ft;
    }
    __syncthreads();
    int* out = plan + (size_t)img * plan_stride(mReco);
    int* gStart = out + 2;
    int* ord = gStart + mReco + 1;
    int* ouid = ord + mReco;
    int* gRep = ouid + mReco;
    int* tRep = gRep + mReco;
    if (tid == 0) {
        int G = 0, U = 0;
        for (int m = 0; m < mReco; m++) {
            if (first[m] == m) { gid[m] = G; gRep[G] = m; cnt[G] = 0; G++; } else gid[m] = gid[first[m]];
            if (firstT[m] == m) { uid[m] = U; tRep[U] = m; U++; } else uid[m] = uid[firstT[m]];
        }
        for (int m = 0; m < mReco; m++) cnt[gid[m]]++;
        int run = 0;
        for (int g = 0; g < G; g++) { gStart[g] = run; run += cnt[g]; cnt[g] = gStart[g]; }
        gStart[G] = run;
        for (int m = 0; m < mReco; m++) { const int pos = cnt[gid[m]]++; ord[pos] = m; }
        out[0] = G;
        out[1] = U;
    }
    __syncthreads();
    for (int m = tid; m < mReco; m += blockDim.x) ouid[m] = uid[m];
}

// sum of the group counts of a launch's images (plan[0] of each) into a running device counter: the adds the window kernel
// actually issues are 24 per (listed pixel, GROUP), not per draw
__global__ void k_plan_groups(unsigned long long* __restrict__ total, const int* __restrict__ plan, int nImg, int stride)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long g = l < nImg ? (unsigned long long)plan[(size_t)l * stride] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
    if ((threadIdx.x & 63) == 0 && g) atomicAdd(total, g);
}

template <int W>
__device__ __forceinline__ int comp3(int x, int y, int z) { return W == 0 ? x : (W == 1 ? y : z); }

struct DrawTables {      // LDS-resident per-image tables built from the insert plan
    const double* R;     // [G][6] rotation columns of each group's representative draw
    const int* gStart;   // [G+1]
    const int* mUid;     // [mReco] unique-shift id of the members, grouped
    const int* gInfo;    // [G][2]: class, representative draw
    const float* slope;  // [U][2] ramp slopes of the unique shifts
    int G, U;
};

// ---------------------------------------------------------------------------------------------
// Insertion, volume-window form (robust to the spread of the draws).
//
// A particle filter's draws are ~1 degree apart (4-16 voxels at radius 250), so the brick cannot follow a tile of image
// pixels: it is a fixed WINDOW of the volume -- kWd x kWd voxels across, kWz thick along the sheared dominant axis of the image's
// reference plane -- and a workgroup owns (image, row of windows): for every window and slab it visits every group of
// draws, enumerates the pixels whose trilinear cell can reach the window (inverse 2x2 map of the window corners: the
// candidates), accumulates the terms that fall INSIDE into the LDS brick (fixed point, as above) and skips the rest --
// a neighbouring window or slab takes them -- then flushes along the volume's x axis.  Every term is added exactly once,
// the flush traffic is that of one large brick, and only ~40 % of the candidates are hits (cheap: positions only).
// ---------------------------------------------------------------------------------------------
#ifndef THX_KWD   // window geometry; overridable at compile time for parameter sweeps (tools/insert_probe.py)
#define THX_KWD 16
#define THX_KWZ 16
#define THX_KIPIX 24
#endif
constexpr int kWd = THX_KWD;                // window edge in (p, q), voxels
constexpr int kWz = THX_KWZ;                // slab thickness along the sheared axis, voxels
constexpr int kWinVox = kWd * kWd * kWz;    // 4096 voxels x 12 B = 48 KB
constexpr int kIPix = THX_KIPIX;            // per-window tabulated pixel range per axis (pixel data, separable ramps)
#ifndef THX_KWINTHREADS
#define THX_KWINTHREADS 512
#endif
#ifdef THX_PROFILING   // THX_INSERT_DEBUG bits (skip LDS adds / flush / group loop) exist only in profiling builds
constexpr bool kWinProfiling = true;
#else
constexpr bool kWinProfiling = false;
#endif
constexpr int kWinThreads = THX_KWINTHREADS; // 2 workgroups per CU (LDS): 512 threads = 4 waves per SIMD at <= 128 VGPRs

struct InsertWinArgs {
    InsertArgs a;
    const int* pixIndex;
    const int* plan;
    const float2* bounds;   // [nImg]: max(|re| + |im|) of the image row, max |ctf|
    long long* accF;        // [nK][vol][2] fixed-point accumulators of this launch (see acc_add)
    long long* accT;        // [nK][vol]
    const int* gexp;        // [2]: E_F, E_T (k_insert_scale)
    float minQuanta;
    int nW;                 // windows per side; window w covers [pOrg + w kWd, pOrg + (w+1) kWd)
    int pOrg;
    float rMax2;            // (largest sample radius + 2)^2, voxels
    int debug;              // builds with -DTHX_PROFILING only (THX_INSERT_DEBUG): 1 skip LDS adds, 2 skip flush, 4 skip the group loop, 8 skip the exact stage, 16 exact geometry only
};

__global__ __launch_bounds__(256) void k_insert_bounds(float2* __restrict__ bounds, const float2* __restrict__ datP,
                                                       const float* __restrict__ ctfP, int nPxl)
{
    __shared__ float sa[4], sc[4];
    const int img = blockIdx.x;
    float am = 0.f, cm = 0.f;
    for (int p = threadIdx.x; p < nPxl; p += blockDim.x) {
        const float2 d = datP[(size_t)img * nPxl + p];
        am = fmaxf(am, fabsf(d.x) + fabsf(d.y));
        cm = fmaxf(cm, fabsf(ctfP[(size_t)img * nPxl + p]));
    }
    am = wave_max(am); cm = wave_max(cm);
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = am; sc[threadIdx.x >> 6] = cm; }
    __syncthreads();
    if (threadIdx.x == 0)
        bounds[img] = make_float2(fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3])), fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])));
}

// exponents of an image's brick quanta: the largest powers of two with |F term| * 2^eF <= 2^(31 - lg), T * 2^eT <= 2^(32 - lg)
__device__ __forceinline__ void image_exponents(float boundF, float boundT, int lg, int& eF, int& eT)
{
    int ebF = 0, ebT = 0;
    (void)frexpf(boundF, &ebF);   // bound = m 2^eb, m in [0.5, 1)
    (void)frexpf(boundT, &ebT);
    eF = 31 - lg - ebF;
    eT = 32 - lg - ebT;
}

// one launch-wide pair of exponents: 7 bits below the finest image quantum (room for the sub-quantum "tiny" terms), the
// finest being taken at most 2^8 finer than that of the image with the largest bound (coarser images shift left exactly;
// a rare image more than 2^8 below the largest shifts right, i.e. is rounded to the launch's quantum -- 2^-46 of the
// largest term).  Headroom: 65535 images x 2^31 x 2^15 < 2^63.
__global__ __launch_bounds__(256) void k_insert_scale(double* __restrict__ ext, const float2* __restrict__ bounds,
                                                      const float* __restrict__ w, int nImg, int mReco, int cSearch)
{
    // ext [4] = max E_F, -min E_F, max E_T, -min E_T over the images (as doubles: the hemisphere takes their maximum with
    // one ncclMax all-reduce, so that every rank of a half works in the SAME quanta and the integer sums of the ranks add
    // up to exactly what one rank would have accumulated)
    __shared__ int sMaxF[4], sMinF[4], sMaxT[4], sMinT[4];
    int lg = 32 - __clz(2 * mReco - 1);
    lg = lg > 20 ? 20 : lg;
    int maxF = INT_MIN, minF = INT_MAX, maxT = INT_MIN, minT = INT_MAX;
    for (int l = threadIdx.x; l < nImg; l += blockDim.x) {
        const float2 bnd = bounds[l];
        const float cmax = cSearch ? 1.0f : bnd.y, wg = fabsf(w[l]);
        const float bF = bnd.x * cmax * wg, bT = cmax * cmax * wg;
        int eF, eT;
        image_exponents(bF, bT, lg, eF, eT);
        if (bF > 0.f) { maxF = max(maxF, eF); minF = min(minF, eF); }
        if (bT > 0.f) { maxT = max(maxT, eT); minT = min(minT, eT); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        maxF = max(maxF, __shfl_xor(maxF, o, 64)); minF = min(minF, __shfl_xor(minF, o, 64));
        maxT = max(maxT, __shfl_xor(maxT, o, 64)); minT = min(minT, __shfl_xor(minT, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { const int wv = threadIdx.x >> 6; sMaxF[wv] = maxF; sMinF[wv] = minF; sMaxT[wv] = maxT; sMinT[wv] = minT; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 0; wv < 4; wv++) { maxF = max(maxF, sMaxF[wv]); minF = min(minF, sMinF[wv]); maxT = max(maxT, sMaxT[wv]); minT = min(minT, sMinT[wv]); }
        // "no image" is encoded as -1e9 on all four so that the maximum over ranks ignores it
        ext[0] = maxF == INT_MIN ? -1e9 : (double)maxF; ext[1] = minF == INT_MAX ? -1e9 : -(double)minF;
        ext[2] = maxT == INT_MIN ? -1e9 : (double)maxT; ext[3] = minT == INT_MAX ? -1e9 : -(double)minT;
    }
}

// gexp = 7 bits below the finest image quantum, the finest taken at most 2^8 finer than the coarsest's; `spare` bits are
// given back when more than 2^16 images (all ranks of the half) accumulate into one volume: 2^16 x 2^31 x 2^15 < 2^63
__global__ void k_insert_scale_final(int* __restrict__ gexp, const double* __restrict__ ext, int spare)
{
    const bool haveF = ext[0] > -1e8, haveT = ext[2] > -1e8;
    const int maxF = (int)ext[0], minF = -(int)ext[1], maxT = (int)ext[2], minT = -(int)ext[3];
    gexp[0] = haveF ? min(maxF, minF + 8) + 7 - spare : 0;
    gexp[1] = haveT ? min(maxT, minT + 8) + 7 - spare : 0;
}

// F += accF 2^-E_F, T += accT 2^-E_T for the voxels the launch touched; one thread per voxel (deterministic)
__global__ __launch_bounds__(256) void k_insert_convert(float2* __restrict__ F, float* __restrict__ T, const long long* __restrict__ accF,
                                                        const long long* __restrict__ accT, const int* __restrict__ gexp, size_t n)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const long long re = accF[2 * e], im = accF[2 * e + 1], tt = accT[e];
    if ((re | im | tt) == 0) return;
    const double iF = ldexp(1.0, -gexp[0]), iT = ldexp(1.0, -gexp[1]);
    float2 f = F[e];
    f.x = f.x + (float)((double)re * iF);
    f.y = f.y + (float)((double)im * iF);
    F[e] = f;
    T[e] = T[e] + (float)((double)tt * iT);
}

struct WinGeom {
    int p0, q0, w0;        // window origin in brick coordinates (w0 = slab base relative to the sheared plane)
    int ui0, uj0;          // origin of the tabulated pixel range
    const float4* pix;     // LDS [kIPix][kIPix]: re, im, ctf, listed
    const float2* ecol;    // LDS [kMaxU][kIPix] exp(-i 2 pi i tx_u / N) (valid when U <= kMaxU)
    const float2* erow;    // LDS [kMaxU][kIPix]
    float sp, sq;
    float devLo, devHi;           // range of (slab offset of a cell's voxel) - (height of its sample): see win_enqueue
    float scaleF, scaleT, minQ;   // scaleF / scaleT: the image's brick quanta per unit, powers of two
    int shF, shT;                 // left shifts taking brick quanta to the launch's global quanta (negative: right)
    float gF, gT;                 // global quanta per unit, 2^E_F / 2^E_T
};

// Rarely taken branches of insert_win_group, kept out of line: inlined, their temporaries (sincos / atan2 sequences, 64-bit
// address arithmetic) set the register peak of the hot loop and push its live values into scratch.
__device__ __attribute__((noinline)) float2 insert_ramp_sum_slow(const float* slope, const int* mUid, int m0, int m1, int pi, int pj)
{
    float2 S = make_float2(0.f, 0.f);
    for (int i = m0; i < m1; i++) {
        const int u = mUid[i];
        const float2 r = ramp_value(slope[2 * u], slope[2 * u + 1], pi, pj);
        S.x += r.x;
        S.y += r.y;
    }
    return S;
}
// The volume accumulators of the window kernel are 64-bit FIXED POINT (quanta 2^-E_F / 2^-E_T of one unit, one pair of
// exponents per launch, k_insert_scale): integer atomic adds commute, so F and T come out bit-identical run to run whatever
// the order in which workgroups flush -- the float atomics of the first version made the gridding loop's round count move
// by +-10 % from run to run.  k_insert_convert adds the accumulators to the caller's float volumes afterwards.
// left shifts are exact; a right shift (an image more than 2^8 below the launch's largest) rounds to nearest, ties away from
// zero -- an arithmetic shift alone would floor, i.e. bias every negative F term of such an image towards -inf
__device__ __forceinline__ long long shift_ll(long long v, int sh)
{
    if (sh >= 0) return v << sh;
    const long long half = 1LL << (-sh - 1);
    return v >= 0 ? (v + half) >> (-sh) : -((-v + half) >> (-sh));
}

__device__ __forceinline__ void insert_tiny_term(long long* F, long long* T, int P, int X, int Y, int Z, float re, float im, float tt,
                                                 float gF, float gT)
{
    const long nc = P / 2 + 1;
    const long gi = ((long)(Z >= 0 ? Z : Z + P) * P + (Y >= 0 ? Y : Y + P)) * nc + X;
    // F and T of a term travel together: a T part below the launch's quantum (a CTF zero: T ~ ctf^2, F ~ ctf) drops the
    // whole term -- T = 0 under F != 0 lets the gridding weights W ~ 1 / (T W conv K) explode (observed: maps x 70)
    const long long t = __float2ll_rn(tt * gT);
    if (t == 0) return;
    acc_add(F, T, gi, __float2ll_rn(re * gF), __float2ll_rn(im * gF), t);
}

// A group whose plane is far from the image's reference plane (|normal component along the shear axis| < kFarGroup, i.e.
// more than ~18 degrees away: a draw from another posterior mode): the sheared-window geometry degenerates for it (slopes
// -n/gna unbounded), so k_insert_win skips it and k_insert_far (one workgroup per image, launched behind it) adds it with
// plain float atomics -- the arithmetic of k_insert for the group's summed ramps.  A separate kernel on purpose: inlined
// or called from k_insert_win its registers cost the hot kernel 48 spilled VGPRs (1.5x slower).  Rare by construction;
// correctness never depends on how the draws are spread.
constexpr float kFarGroup = 0.3f;

// accumulate group gi_'s candidates of this window/slab (one wave), AX = dominant axis of the reference plane
// Geometry of one (group, pixel) sample against the current window / slab: cell origin, brick coordinates of its
// 8 voxels and the mask of those that lie inside.  Returns false when nothing of the cell belongs here.
struct WinSample {
    float x, y, z, fx, fy, fz;
    int X0, Y0, Z0, sg;
    int pI[2], qI[2], offA[2][2];
    unsigned inMask;
    bool conj;
};

template <int AX>
__device__ __forceinline__ bool win_sample(const WinGeom& g, const double* R, int opf, int P, int pi, int pj, WinSample& w)
{
    constexpr int pa = AX == 0 ? 1 : 0;
    constexpr int qa = AX == 2 ? 1 : 2;
    const int icp = pi * opf, irp = pj * opf;
    float x = (float)(R[0] * icp + R[3] * irp);
    float y = (float)(R[1] * icp + R[4] * irp);
    float z = (float)(R[2] * icp + R[5] * irp);
    if (!coord_in_grid(x, y, z, P)) return false;
    w.conj = false;
    if (!(x >= 0.0f)) { x *= -1.0f; y *= -1.0f; z *= -1.0f; w.conj = true; }
    w.x = x; w.y = y; w.z = z;
    w.fx = floorf(x); w.fy = floorf(y); w.fz = floorf(z);
    w.X0 = (int)w.fx; w.Y0 = (int)w.fy; w.Z0 = (int)w.fz;
    w.sg = w.conj ? -1 : 1;
    const int b0x = w.conj ? -1 - w.X0 : w.X0, b0y = w.conj ? -w.Y0 : w.Y0, b0z = w.conj ? -w.Z0 : w.Z0;
    const int bp0 = comp3<pa>(b0x, b0y, b0z), bq0 = comp3<qa>(b0x, b0y, b0z), ba0 = comp3<AX>(b0x, b0y, b0z);
    bool pin[2], qin[2];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        w.pI[d] = bp0 + w.sg * d - g.p0;
        w.qI[d] = bq0 + w.sg * d - g.q0;
        pin[d] = (unsigned)w.pI[d] < (unsigned)kWd;
        qin[d] = (unsigned)w.qI[d] < (unsigned)kWd;
    }
    if (!((pin[0] || pin[1]) && (qin[0] || qin[1]))) return false;   // the cell misses this window's columns
#pragma unroll
    for (int dq = 0; dq < 2; dq++)
#pragma unroll
        for (int dp = 0; dp < 2; dp++)
            w.offA[dq][dp] = ba0 - ((int)floorf(g.sp * (float)(bp0 + w.sg * dp) + g.sq * (float)(bq0 + w.sg * dq)) + g.w0);
    unsigned inMask = 0;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        const int ii = v & 1, jj = (v >> 1) & 1, kk = v >> 2;
        const int dp = comp3<pa>(ii, jj, kk), dq = comp3<qa>(ii, jj, kk), da = comp3<AX>(ii, jj, kk);
        const int off = w.offA[dq][dp] + w.sg * da;
        if (pin[dp] && qin[dq] && ((unsigned)off < (unsigned)kWz)) inMask |= 1u << v;
    }
    w.inMask = inMask;
    return inMask != 0;
}

// The group walk of k_insert_win, in two phases per wave.
// win_enqueue: the wave walks the candidate box of one group and tests every pixel in float -- (p, q) inside the padded
//   window, height above the sheared reference plane inside the padded slab, pixel listed -- and appends the probable hits
//   (pixel + group, 32 bits) to its LDS queue; only ~22 of the 64 lanes of a box trip are real hits, so doing the exact
//   work there would leave two thirds of the VALU idle (the kernel is VALU-bound: 65 % busy at 3.3 waves per SIMD).
// win_process: one queue entry per lane -- exact geometry (fp64 position, cell, voxel mask), the pixel's value for the
//   group, the 8 voxel terms -- run whenever 64 entries are waiting, and once more at the end of the slab for the rest.
//   The queue outlives the groups of a slab, so practically every trip has full lanes.
constexpr int kWinQueue = 128;   // entries per wave: at most 63 waiting + 64 appended per box trip

__device__ __forceinline__ int win_pack(int pi, int pj, int gi) { return pi | ((pj + 1024) << 11) | (gi << 22); }

template <int AX>
__device__ __forceinline__ void win_process(const InsertWinArgs& wa, const WinGeom& g, int* sRe, int* sIm, int* sT,
                                            const DrawTables& dt, int img, float wgt, long long* F, long long* T, int pk)
{
    const InsertArgs& a = wa.a;
    constexpr int pa = AX == 0 ? 1 : 0;
    constexpr int qa = AX == 2 ? 1 : 2;
    const int P = a.P, half = a.idim / 2;
    const int pi = pk & 0x7FF, pj = ((pk >> 11) & 0x7FF) - 1024, gi_ = (int)((unsigned)pk >> 22);
    const double* R = dt.R + 6 * gi_;
    WinSample w;
    if (kWinProfiling && (wa.debug & 8)) { if (pk == 0x7fffffff) sRe[0] = 1; return; }
    if (!win_sample<AX>(g, R, a.opf, P, pi, pj, w)) return;   // a false positive of the float test
    if (kWinProfiling && (wa.debug & 16)) { if (w.inMask == 0x12345) sRe[0] = 1; return; }
    const int m0 = dt.gStart[gi_], m1 = dt.gStart[gi_ + 1];
    const float nmem = (float)(m1 - m0);
    const int ti = pi - g.ui0, tj = pj - g.uj0;
    const bool tab = (unsigned)ti < (unsigned)kIPix && (unsigned)tj < (unsigned)kIPix;
    float2 dv;
    float cf;
    if (tab) {
        const float4 px = g.pix[tj * kIPix + ti];
        dv = make_float2(px.x, px.y); cf = px.z;
    } else {
        const int k = wa.pixIndex[(pj + half) * (half + 1) + pi];
        dv = a.datP[(size_t)img * a.nPxl + k]; cf = a.ctfP[(size_t)img * a.nPxl + k];
    }
    // the value of this pixel for the group: (img * sum of the members' ramps) * ctf * w, T: n * ctf^2 * w
    float2 S = make_float2(0.f, 0.f);
    if (tab && dt.U <= kMaxU) {
        for (int i = m0; i < m1; i++) {   // separable ramp exp(-i a_u i) exp(-i b_u j) from the window's tables
            const int u = dt.mUid[i];
            const float2 ec = g.ecol[u * kIPix + ti], er = g.erow[u * kIPix + tj];
            S.x += ec.x * er.x - ec.y * er.y;
            S.y += ec.x * er.y + ec.y * er.x;
        }
    } else {
        S = insert_ramp_sum_slow(dt.slope, dt.mUid, m0, m1, pi, pj);
    }
    const float2 tv = cmul(dv, S);
    if (a.cSearch) cf = insert_ctf_search(a.attr, a.dfac, img, a.mReco, dt.gInfo[2 * gi_ + 1], a.pixelSize, a.idim, pi, pj);
    float vre = tv.x * cf, vim = tv.y * cf;
    vre = vre * 1.0f; vim = vim * 1.0f;
    vre = vre * wgt; vim = vim * wgt;
    if (w.conj) vim = -vim;
    const float tval = (pow2f_(cf) * 1.0f * wgt) * nmem;
    const float xd = w.x - w.fx, yd = w.y - w.fy, zd = w.z - w.fz;
    const float vx[2] = {1.0f - xd, xd}, vy[2] = {1.0f - yd, yd}, vz[2] = {1.0f - zd, zd};
    // fixed-point scales folded into the pixel's value once (the brick's quantum is 2^-22 of the largest term)
    const float vreS = vre * g.scaleF, vimS = vim * g.scaleF, tvalS = tval * g.scaleT;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        if (!((w.inMask >> v) & 1)) continue;
        const int ii = v & 1, jj = (v >> 1) & 1, kk = v >> 2;
        const int dp = comp3<pa>(ii, jj, kk), dq = comp3<qa>(ii, jj, kk), da = comp3<AX>(ii, jj, kk);
        const float wv = vx[ii] * vy[jj] * vz[kk];
        const int off = w.offA[dq][dp] + w.sg * da;
        const float tq = tvalS * wv;
        if (tq >= g.minQ) {
            const int idx = AX == 0 ? ((w.qI[dq] * kWd + w.pI[dp]) * kWz + off) : ((w.qI[dq] * kWz + off) * kWd + w.pI[dp]);
            if (kWinProfiling && (wa.debug & 1)) { if (idx < 0) sRe[0] = 1; continue; }
            atomicAdd(&sRe[idx], __float2int_rn(vreS * wv));
            atomicAdd(&sIm[idx], __float2int_rn(vimS * wv));
            atomicAdd(reinterpret_cast<unsigned*>(&sT[idx]), __float2uint_rn(tq));
        } else {
            // tiny term: F and T travel together as floats
            insert_tiny_term(F, T, P, w.X0 + ii, w.Y0 + jj, w.Z0 + kk, vre * wv, vim * wv, tval * wv, g.gF, g.gT);
        }
    }
}

// the entries still waiting at the end of a slab
template <int AX>
__device__ __forceinline__ void win_drain(const InsertWinArgs& wa, const WinGeom& g, int* sRe, int* sIm, int* sT,
                                          const DrawTables& dt, int img, float wgt, long long* F, long long* T, volatile int* queue,
                                          int& qn)
{
    if (qn > 0) {
        const int lane = threadIdx.x & 63;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int e0 = lane < qn ? queue[lane] : 0;
        if (lane < qn) win_process<AX>(wa, g, sRe, sIm, sT, dt, img, wgt, F, T, e0);
        qn = 0;
    }
}

template <int AX>
__device__ __forceinline__ void win_enqueue(const InsertWinArgs& wa, const WinGeom& g, int* sRe, int* sIm, int* sT,
                                            const DrawTables& dt, int img, int gi_, int i0, int nI, int j0, int nJ, float wgt,
                                            long long* F, long long* T, volatile int* queue, int& qn)
{
    const InsertArgs& a = wa.a;
    constexpr int pa = AX == 0 ? 1 : 0;
    constexpr int qa = AX == 2 ? 1 : 2;
    const int lane = threadIdx.x & 63;
    const int half = a.idim / 2;
    const double* R = dt.R + 6 * gi_;
    const int nCand = nI * nJ;
    const float rnI = 1.0f / (float)nI;
    // rows pa, qa, AX of the group's rotation: (p, q) of a pixel and its height above the sheared reference plane.  Margins:
    // a voxel of a sample's cell sits at sample + d, d in (-1, 1] per axis, and d_x in [-2, 0) for a Hermitian-folded sample
    // (its brick x is -1 - X).  Its slab offset is  off = (wf - w0) + d_a - sp d_p - sq d_q + frac,  frac in [0, 1) the
    // floor of the shear: off - (wf - w0) lies in [devLo, devHi) (per image, from the signs and sizes of its slopes; at most
    // [-4, 5)), so a voxel inside the slab (0 <= off < kWz) needs wf in (w0 - devHi, w0 + kWz - devLo].  (The first version
    // used fixed -4.5 / +3.5: samples of planes with both slopes near 1 were dropped at slab boundaries -- 1e-5 of the mass,
    // 5e-3 of max T at single voxels; tests/test_fullsize_gpu.py.)
    const float A00 = (float)R[pa] * (float)a.opf, A01 = (float)R[3 + pa] * (float)a.opf, A10 = (float)R[qa] * (float)a.opf,
                A11 = (float)R[3 + qa] * (float)a.opf, A20 = (float)R[AX] * (float)a.opf, A21 = (float)R[3 + AX] * (float)a.opf;
    const float plo = (float)g.p0 - 2.5f, phi = (float)(g.p0 + kWd) + 1.5f, qlo = (float)g.q0 - 2.5f, qhi = (float)(g.q0 + kWd) + 1.5f;
    const float wlo = (float)g.w0 - g.devHi - 0.25f, whi = (float)(g.w0 + kWz) - g.devLo + 0.25f;
    for (int c0 = 0; c0 < nCand; c0 += 64) {
        const int c = c0 + lane;
        bool hit = false;
        int pk = 0;
        if (c < nCand) {
            const int jr = (int)(((float)c + 0.5f) * rnI);
            const int pi = i0 + (c - jr * nI), pj = j0 + jr;
            const float pf_ = A00 * (float)pi + A01 * (float)pj, qf_ = A10 * (float)pi + A11 * (float)pj;
            const float wf_ = (A20 * (float)pi + A21 * (float)pj) - (g.sp * pf_ + g.sq * qf_);
            if (pf_ >= plo && pf_ < phi && qf_ >= qlo && qf_ < qhi && wf_ >= wlo && wf_ < whi) {
                const int ti = pi - g.ui0, tj = pj - g.uj0;
                const bool tab = (unsigned)ti < (unsigned)kIPix && (unsigned)tj < (unsigned)kIPix;
                hit = tab ? (g.pix[tj * kIPix + ti].w != 0.f) : (wa.pixIndex[(pj + half) * (half + 1) + pi] >= 0);
                pk = win_pack(pi, pj, gi_);
            }
        }
        const unsigned long long bal = __ballot(hit);
        if (hit) queue[qn + __popcll(bal & ((1ull << lane) - 1ull))] = pk;
        qn += __popcll(bal);
        if (qn >= 64) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int e0 = queue[lane];
            const int e1 = (64 + lane < qn) ? queue[64 + lane] : 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (64 + lane < qn) queue[lane] = e1;   // the remainder moves to the front (each lane moves its own entry)
            qn -= 64;
            win_process<AX>(wa, g, sRe, sIm, sT, dt, img, wgt, F, T, e0);
        }
    }
}

template <int AX>
__device__ __forceinline__ void insert_win_flush(const InsertArgs& a, const WinGeom& g, int* sRe, int* sIm, int* sT, long long* F,
                                                 long long* T)
{
    constexpr int pa = AX == 0 ? 1 : 0;
    constexpr int qa = AX == 2 ? 1 : 2;
    const int P = a.P;
    const long nc = P / 2 + 1;
    for (int e = threadIdx.x; e < kWinVox; e += kWinThreads) {
        const int ire = sRe[e], iim = sIm[e], itt = sT[e];
        if ((ire | iim | itt) == 0) continue;
        sRe[e] = 0; sIm[e] = 0; sT[e] = 0;
        int p_i, q_i, off;
        if (AX == 0) { off = e % kWz; const int r = e / kWz; q_i = r / kWd; p_i = r - q_i * kWd; }
        else { const int r = e / kWd; p_i = e - r * kWd; off = r % kWz; q_i = r / kWz; }
        const int bp = p_i + g.p0, bq = q_i + g.q0;
        const int ba = off + ((int)floorf(g.sp * (float)bp + g.sq * (float)bq) + g.w0);
        int X = pa == 0 ? bp : ba;
        int Y = pa == 1 ? bp : (qa == 1 ? bq : ba);
        int Z = qa == 2 ? bq : ba;
        if (X < 0) { X = -1 - X; Y = -Y; Z = -Z; }
        const long gi = ((long)(Z >= 0 ? Z : Z + P) * P + (Y >= 0 ? Y : Y + P)) * nc + X;
        const long long t = shift_ll((long long)(unsigned)itt, g.shT);
        if (t == 0) continue;   // (only with a right shift: an image far below the launch's scale) F and T travel together
        acc_add(F, T, gi, shift_ll((long long)ire, g.shF), shift_ll((long long)iim, g.shF), t);
    }
}

// grid (nW, nImg): one workgroup owns a row of windows (fixed q range) of one image
__global__ __launch_bounds__(kWinThreads, kWinThreads / 128) void k_insert_win(InsertWinArgs wa)
{
    const InsertArgs& a = wa.a;
    extern __shared__ __attribute__((aligned(16))) int brick[];
    int* sRe = brick;
    int* sIm = brick + kWinVox;
    int* sT = brick + 2 * kWinVox;
    double* sR = reinterpret_cast<double*>(brick + 3 * kWinVox);            // [mReco][6]
    int* sGStart = reinterpret_cast<int*>(sR + 6 * a.mReco);                // [mReco+1]
    int* sMUid = sGStart + a.mReco + 1;                                      // [mReco]
    int* sGInfo = sMUid + a.mReco;                                           // [mReco][2]
    float* sSlope = reinterpret_cast<float*>(sGInfo + 2 * a.mReco);          // [mReco][2]
    short* sBox = reinterpret_cast<short*>(sSlope + 2 * a.mReco);            // [mReco][4]: i0, nI, j0, nJ (nI = 0: no candidates)
    float* sWr = reinterpret_cast<float*>(sBox + 4 * a.mReco);               // [mReco][2]: w range of the group in this window
    float4* sPix = reinterpret_cast<float4*>((reinterpret_cast<uintptr_t>(sWr + 2 * a.mReco) + 15) & ~(uintptr_t)15);  // [kIPix][kIPix]
    float2* sEc = reinterpret_cast<float2*>(sPix + kIPix * kIPix);           // [kMaxU][kIPix]
    float2* sEr = sEc + kMaxU * kIPix;                                       // [kMaxU][kIPix]
    int* sQueue = reinterpret_cast<int*>(sEr + kMaxU * kIPix);               // [waves][kWinQueue] probable-hit queues (win_enqueue)
    __shared__ int sWlo, sWhi, sCls, sUi0, sUi1, sUj0, sUj1, sNext;

    const int img = blockIdx.y, wqI = blockIdx.x;
    const int tid = threadIdx.x, grp = tid >> 6;
    const int P = a.P, half = a.idim / 2;
    const size_t volSize = (size_t)P * P * (P / 2 + 1);
    const double offx = a.offS ? a.offS[2 * img] : 0.0, offy = a.offS ? a.offS[2 * img + 1] : 0.0;

    const int* plan = wa.plan + (size_t)img * plan_stride(a.mReco);
    const int G = plan[0], U = plan[1];
    const int* pGStart = plan + 2;
    const int* pOrd = pGStart + a.mReco + 1;
    const int* pUid = pOrd + a.mReco;
    const int* pGRep = pUid + a.mReco;
    const int* pTRep = pGRep + a.mReco;
    for (int gi_ = tid; gi_ < G; gi_ += kWinThreads) {
        const int rep = pGRep[gi_];
        const double* R = a.rotMat + ((size_t)img * a.mReco + rep) * 9;
        double* d = sR + 6 * gi_;
        d[0] = R[0]; d[1] = R[1]; d[2] = R[2]; d[3] = R[3]; d[4] = R[4]; d[5] = R[5];
        sGInfo[2 * gi_] = a.cls ? a.cls[(size_t)img * a.mReco + rep] : 0;
        sGInfo[2 * gi_ + 1] = rep;
    }
    for (int i = tid; i <= G; i += kWinThreads) sGStart[i] = pGStart[i];
    for (int i = tid; i < a.mReco; i += kWinThreads) sMUid[i] = pUid[pOrd[i]];
    for (int u = tid; u < U; u += kWinThreads) {
        const size_t dm = (size_t)img * a.mReco + pTRep[u];
        const double tx = a.trans[2 * dm] - offx, ty = a.trans[2 * dm + 1] - offy;
        sSlope[2 * u] = (float)(-tx) / a.idim;
        sSlope[2 * u + 1] = (float)(-ty) / a.idim;
    }
    if (wqI == 0 && tid == 0 && a.O) {   // insertDir (src/Reconstructor.cpp:407-422) once per image
        double ox = 0, oy = 0, oz = 0;
        for (int m = 0; m < a.mReco; m++) {
            const size_t dm = (size_t)img * a.mReco + m;
            const double* R = a.rotMat + dm * 9;
            const double tx = a.trans[2 * dm] - offx, ty = a.trans[2 * dm + 1] - offy;
            ox += -(R[0] * tx + R[3] * ty);
            oy += -(R[1] * tx + R[4] * ty);
            oz += -(R[2] * tx + R[5] * ty);
        }
        unsafeAtomicAdd(&a.O[0], ox);
        unsafeAtomicAdd(&a.O[1], oy);
        unsafeAtomicAdd(&a.O[2], oz);
        if (a.counter) atomicAdd(a.counter, a.mReco);
    }
    for (int e = tid; e < 3 * kWinVox; e += kWinThreads) brick[e] = 0;
    __syncthreads();

    // reference plane = the first group's: dominant axis of its normal, column slopes of the shear
    const double* R0 = sR;
    const float n0 = (float)(R0[1] * R0[5] - R0[2] * R0[4]);
    const float n1 = (float)(R0[2] * R0[3] - R0[0] * R0[5]);
    const float n2 = (float)(R0[0] * R0[4] - R0[1] * R0[3]);
    const float an0 = fabsf(n0), an1 = fabsf(n1), an2 = fabsf(n2);
    const int ax = (an0 >= an1 && an0 >= an2) ? 0 : (an1 >= an2 ? 1 : 2);
    const int pa = ax == 0 ? 1 : 0, qa = ax == 2 ? 1 : 2;
    const float na = ax == 0 ? n0 : (ax == 1 ? n1 : n2);
    WinGeom g;
    g.sp = -(pa == 0 ? n0 : n1) / na;
    g.sq = -(qa == 1 ? n1 : n2) / na;
    {   // d_x in [-2, 1] (Hermitian fold), d_y, d_z in (-1, 1]; q is never the x axis
        const float Lp = pa == 0 ? 2.f : 1.f, La = ax == 0 ? 2.f : 1.f;
        g.devHi = 1.f + fabsf(g.sp) * (g.sp > 0.f ? Lp : 1.f) + fabsf(g.sq) + 1.f;
        g.devLo = -La - fabsf(g.sp) * (g.sp > 0.f ? 1.f : Lp) - fabsf(g.sq);
    }
    const float wgt = a.w[img];
    DrawTables dt;
    dt.R = sR; dt.gStart = sGStart; dt.mUid = sMUid; dt.gInfo = sGInfo; dt.slope = sSlope; dt.G = G; dt.U = U;
    int lg = 32 - __clz(2 * a.mReco - 1);
    lg = lg > 20 ? 20 : lg;
    const float2 bnd = wa.bounds[img];
    const float cmax = a.cSearch ? 1.0f : bnd.y;
    const float boundF = bnd.x * cmax * fabsf(wgt), boundT = cmax * cmax * fabsf(wgt);
    if (!(boundF > 0.f) && !(boundT > 0.f)) return;
    // brick quanta: the largest power of two with  bound * scale <= 2^(31 - lg) (F, signed) / 2^(32 - lg) (T, unsigned)
    int eF, eT;
    image_exponents(boundF, boundT, lg, eF, eT);
    g.scaleF = boundF > 0.f ? ldexpf(1.0f, eF) : 0.f;
    g.scaleT = boundT > 0.f ? ldexpf(1.0f, eT) : 0.f;
    const int EF = wa.gexp[0], ET = wa.gexp[1];
    g.shF = EF - eF; g.shT = ET - eT;
    g.gF = ldexpf(1.0f, EF); g.gT = ldexpf(1.0f, ET);
    g.minQ = wa.minQuanta;
    g.q0 = wa.pOrg + wqI * kWd;
    g.pix = sPix; g.ecol = sEc; g.erow = sEr;
    const float wBound = sqrtf(wa.rMax2) * (1.0f + fabsf(g.sp) + fabsf(g.sq)) + 4.0f;

    const int nPass = a.cls ? a.nK : 1;
    for (int pass = 0; pass < nPass; pass++) {
        if (a.cls) {
            __syncthreads();
            if (tid == 0) sCls = 0;
            __syncthreads();
            for (int gi_ = tid; gi_ < G; gi_ += kWinThreads)
                if (sGInfo[2 * gi_] == pass) sCls = 1;
            __syncthreads();
            if (!sCls) continue;
        }
        long long* F = wa.accF + (size_t)pass * volSize * 2;
        long long* T = wa.accT + (size_t)pass * volSize;
        for (int wpI = 0; wpI < wa.nW; wpI++) {
            g.p0 = wa.pOrg + wpI * kWd;
            // nearest point of the (padded) window to the origin, in the (p, q) projection: beyond every sample?
            {
                const float lo_p = (float)(g.p0 - 2), hi_p = (float)(g.p0 + kWd + 1), lo_q = (float)(g.q0 - 2), hi_q = (float)(g.q0 + kWd + 1);
                const float dp = lo_p > 0.f ? lo_p : (hi_p < 0.f ? -hi_p : 0.f), dq = lo_q > 0.f ? lo_q : (hi_q < 0.f ? -hi_q : 0.f);
                if (dp * dp + dq * dq > wa.rMax2) continue;
            }
            __syncthreads();   // previous window's readers of sBox / sWr / sWlo are done
            if (tid == 0) { sWlo = INT_MAX; sWhi = INT_MIN; sUi0 = INT_MAX; sUi1 = INT_MIN; sUj0 = INT_MAX; sUj1 = INT_MIN; sNext = 0; }
            __syncthreads();
            // ---- per group: candidate pixel box (inverse 2x2 map of the padded window corners) and sheared-w range ----
            // (the six extrema are reduced inside the wave and reach LDS once per wave: 8 same-address LDS atomics from
            // each of 62 lanes, with the other seven waves waiting at the barrier, were 14 % of the kernel)
            int lUi0 = INT_MAX, lUi1 = INT_MIN, lUj0 = INT_MAX, lUj1 = INT_MIN, lWlo = INT_MAX, lWhi = INT_MIN;
            for (int gi_ = tid; gi_ < G; gi_ += kWinThreads) {
                const double* R = sR + 6 * gi_;
                short* box = sBox + 4 * gi_;
                box[1] = 0;
                if (a.cls && sGInfo[2 * gi_] != pass) continue;
                // (p, q) = opf * A (i, j),  A = rows pa, qa of the first two columns of R; det = +-(the normal's component
                // along the shear axis): groups with |det| < kFarGroup are handled by k_insert_far
                const float A00 = (float)R[pa], A01 = (float)R[3 + pa], A10 = (float)R[qa], A11 = (float)R[3 + qa];
                const float det = A00 * A11 - A01 * A10;
                if (fabsf(det) < 0.5f * kFarGroup) continue;
                {
                    const float s = 1.0f / (det * (float)a.opf);
                    float imin = 1e30f, imax = -1e30f, jmin = 1e30f, jmax = -1e30f;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float pc = (float)((c & 1) ? g.p0 + kWd + 1 : g.p0 - 2), qc = (float)((c & 2) ? g.q0 + kWd + 1 : g.q0 - 2);
                        const float fi = (A11 * pc - A01 * qc) * s, fj = (-A10 * pc + A00 * qc) * s;
                        imin = fminf(imin, fi); imax = fmaxf(imax, fi); jmin = fminf(jmin, fj); jmax = fmaxf(jmax, fj);
                    }
                    int i0 = (int)floorf(imin), i1 = (int)ceilf(imax), j0 = (int)floorf(jmin), j1 = (int)ceilf(jmax);
                    i0 = i0 < 0 ? 0 : i0; i1 = i1 > half ? half : i1;
                    j0 = j0 < -half ? -half : j0; j1 = j1 > half - 1 ? half - 1 : j1;
                    if (i1 < i0 || j1 < j0) continue;
                    box[0] = (short)i0; box[1] = (short)(i1 - i0 + 1); box[2] = (short)j0; box[3] = (short)(j1 - j0 + 1);
                    lUi0 = min(lUi0, i0); lUi1 = max(lUi1, i1); lUj0 = min(lUj0, j0); lUj1 = max(lUj1, j1);
                }
                // height of the group's plane above the reference shear at the window corners
                const float gn0 = (float)(R[1] * R[5] - R[2] * R[4]), gn1 = (float)(R[2] * R[3] - R[0] * R[5]),
                            gn2 = (float)(R[0] * R[4] - R[1] * R[3]);
                const float gna = ax == 0 ? gn0 : (ax == 1 ? gn1 : gn2);
                // A draw from a far-away posterior mode can be (nearly) parallel to the shear axis: its slopes -n/gna blow up
                // (gna -> 0; NaN at 0) and its candidate box is the whole image.  Such groups never enter the window walk:
                // k_insert_far adds them with plain atomics, once per image.
                if (fabsf(gna) < kFarGroup) { box[1] = 0; continue; }
                const float gsp = -(pa == 0 ? gn0 : gn1) / gna, gsq = -(qa == 1 ? gn1 : gn2) / gna;
                float wmin = 1e30f, wmax = -1e30f;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float pc = (float)((c & 1) ? g.p0 + kWd + 1 : g.p0 - 2), qc = (float)((c & 2) ? g.q0 + kWd + 1 : g.q0 - 2);
                    const float wv = (gsp - g.sp) * pc + (gsq - g.sq) * qc;
                    wmin = fminf(wmin, wv); wmax = fmaxf(wmax, wv);
                }
                // every sample satisfies |a - sp p - sq q| <= rMax (1 + |sp| + |sq|): slabs beyond that hold nothing
                wmin = fmaxf(wmin, -wBound); wmax = fminf(wmax, wBound);
                if (wmin > wmax) { box[1] = 0; continue; }
                // slabs this group can reach: a voxel offset in [0, kWz) needs w0 in (wf + devLo - kWz, wf + devHi), wf in [wmin, wmax]
                sWr[2 * gi_] = wmin + g.devLo - 0.25f;
                sWr[2 * gi_ + 1] = wmax + g.devHi + 0.25f;
                lWlo = min(lWlo, (int)floorf(wmin + g.devLo - 0.25f));
                lWhi = max(lWhi, (int)ceilf(wmax + g.devHi + 0.25f));
            }
            if (tid < ((G + 63) & ~63)) {   // the waves that held groups (wave-uniform condition)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    lUi0 = min(lUi0, __shfl_xor(lUi0, o, 64)); lUi1 = max(lUi1, __shfl_xor(lUi1, o, 64));
                    lUj0 = min(lUj0, __shfl_xor(lUj0, o, 64)); lUj1 = max(lUj1, __shfl_xor(lUj1, o, 64));
                    lWlo = min(lWlo, __shfl_xor(lWlo, o, 64)); lWhi = max(lWhi, __shfl_xor(lWhi, o, 64));
                }
                if ((tid & 63) == 0) {
                    atomicMin(&sUi0, lUi0); atomicMax(&sUi1, lUi1); atomicMin(&sUj0, lUj0); atomicMax(&sUj1, lUj1);
                    atomicMin(&sWlo, lWlo); atomicMax(&sWhi, lWhi);
                }
            }
            __syncthreads();
            if (sWlo > sWhi) continue;
            // ---- pixel data and separable ramps of the unique shifts for the pixel range in play ----
            g.ui0 = sUi0 <= sUi1 ? sUi0 - ((kIPix - (sUi1 - sUi0 + 1)) > 0 ? (kIPix - (sUi1 - sUi0 + 1)) / 2 : 0) : 0;
            g.uj0 = sUj0 <= sUj1 ? sUj0 - ((kIPix - (sUj1 - sUj0 + 1)) > 0 ? (kIPix - (sUj1 - sUj0 + 1)) / 2 : 0) : -half;
            for (int e = tid; e < kIPix * kIPix && !(kWinProfiling && (wa.debug & 32)); e += kWinThreads) {
                const int tj = e / kIPix, ti = e - tj * kIPix;
                const int pi = g.ui0 + ti, pj = g.uj0 + tj;
                float4 px = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pi >= 0 && pi <= half && pj >= -half && pj < half) {
                    const int k = wa.pixIndex[(pj + half) * (half + 1) + pi];
                    if (k >= 0) {
                        const float2 dv = a.datP[(size_t)img * a.nPxl + k];
                        px = make_float4(dv.x, dv.y, a.ctfP[(size_t)img * a.nPxl + k], 1.f);
                    }
                }
                sPix[e] = px;
            }
            if (U <= kMaxU && !(kWinProfiling && (wa.debug & 64)))
                for (int e = tid; e < 2 * U * kIPix; e += kWinThreads) {
                    const int which = e / (U * kIPix), rem = e - which * U * kIPix, u = rem / kIPix, o = rem - u * kIPix;
                    // exp(-2 pi i n slope): the whole turns are removed in double (n slope is exact to 1e-13), the
                    // remaining fraction of a turn goes through sincospif -- within 1e-7 of the double-precision value
                    double cyc = (double)((which ? g.uj0 : g.ui0) + o) * (double)sSlope[2 * u + which];
                    cyc -= rint(cyc);
                    float sn, cs;
                    sincospif(-2.0f * (float)cyc, &sn, &cs);
                    (which ? sEr : sEc)[u * kIPix + o] = make_float2(cs, sn);
                }
            __syncthreads();
            const int sLo = (sWlo + kWz / 2) >= 0 ? (sWlo + kWz / 2) / kWz : -((-(sWlo + kWz / 2) + kWz - 1) / kWz);
            const int sHi = (sWhi + kWz / 2) >= 0 ? (sWhi + kWz / 2) / kWz : -((-(sWhi + kWz / 2) + kWz - 1) / kWz);
            for (int sl = sLo; sl <= sHi && !(kWinProfiling && (wa.debug & 128)); sl++) {
                g.w0 = sl * kWz - kWz / 2;
                // the waves draw groups from a shared counter: the work per group varies (candidate box, slab overlap) and
                // every slab ends in a barrier, so a static split leaves waves idle at it
                int qn = 0;   // entries waiting in this wave's queue (wave-uniform)
                for (;;) {
                    int gi_ = 0;
                    if ((tid & 63) == 0) gi_ = atomicAdd(&sNext, 1);
                    gi_ = __builtin_amdgcn_readfirstlane(gi_);
                    if (gi_ >= G) break;
                    const short* box = sBox + 4 * gi_;
                    if (box[1] == 0 || (kWinProfiling && (wa.debug & 4))) continue;
                    if (sWr[2 * gi_ + 1] < (float)g.w0 || sWr[2 * gi_] > (float)(g.w0 + kWz)) continue;
                    if (ax == 0) win_enqueue<0>(wa, g, sRe, sIm, sT, dt, img, gi_, box[0], box[1], box[2], box[3], wgt, F, T, sQueue + kWinQueue * grp, qn);
                    else if (ax == 1) win_enqueue<1>(wa, g, sRe, sIm, sT, dt, img, gi_, box[0], box[1], box[2], box[3], wgt, F, T, sQueue + kWinQueue * grp, qn);
                    else win_enqueue<2>(wa, g, sRe, sIm, sT, dt, img, gi_, box[0], box[1], box[2], box[3], wgt, F, T, sQueue + kWinQueue * grp, qn);
                }
                if (ax == 0) win_drain<0>(wa, g, sRe, sIm, sT, dt, img, wgt, F, T, sQueue + kWinQueue * grp, qn);
                else if (ax == 1) win_drain<1>(wa, g, sRe, sIm, sT, dt, img, wgt, F, T, sQueue + kWinQueue * grp, qn);
                else win_drain<2>(wa, g, sRe, sIm, sT, dt, img, wgt, F, T, sQueue + kWinQueue * grp, qn);
                lds_barrier();
                if (tid == 0) sNext = 0;   // nobody draws between this barrier and the one after the flush
                if (kWinProfiling && (wa.debug & 2)) { lds_barrier(); continue; }
                if (ax == 0) insert_win_flush<0>(a, g, sRe, sIm, sT, F, T);
                else if (ax == 1) insert_win_flush<1>(a, g, sRe, sIm, sT, F, T);
                else insert_win_flush<2>(a, g, sRe, sIm, sT, F, T);
                lds_barrier();   // the flush's global atomics stay in flight
            }
        }
    }
}

// grid (nImg), block 256: the groups of an image that k_insert_win leaves out (see kFarGroup).  The reference plane and its
// shear axis are derived exactly as k_insert_win derives them (first group's normal, same float expressions).
__global__ __launch_bounds__(256) void k_insert_far(InsertWinArgs wa)
{
    const InsertArgs& a = wa.a;
    const int img = blockIdx.x;
    const int P = a.P;
    const size_t volSize = (size_t)P * P * (P / 2 + 1);
    const int* plan = wa.plan + (size_t)img * plan_stride(a.mReco);
    const int G = plan[0];
    const int* pGStart = plan + 2;
    const int* pOrd = pGStart + a.mReco + 1;
    const int* pGRep = pOrd + 2 * a.mReco;
    const double* Rimg = a.rotMat + (size_t)img * a.mReco * 9;
    const double* R0 = Rimg + (size_t)pGRep[0] * 9;
    const float n0 = (float)(R0[1] * R0[5] - R0[2] * R0[4]);
    const float n1 = (float)(R0[2] * R0[3] - R0[0] * R0[5]);
    const float n2 = (float)(R0[0] * R0[4] - R0[1] * R0[3]);
    const float an0 = fabsf(n0), an1 = fabsf(n1), an2 = fabsf(n2);
    const int ax = (an0 >= an1 && an0 >= an2) ? 0 : (an1 >= an2 ? 1 : 2);
    const float wgt = a.w[img];
    const float gF = ldexpf(1.0f, wa.gexp[0]), gT = ldexpf(1.0f, wa.gexp[1]);
    const double offx = a.offS ? a.offS[2 * img] : 0.0, offy = a.offS ? a.offS[2 * img + 1] : 0.0;
    for (int gi_ = 0; gi_ < G; gi_++) {
        const int rep = pGRep[gi_];
        const double* R = Rimg + (size_t)rep * 9;
        const float gna = (float)(ax == 0 ? R[1] * R[5] - R[2] * R[4] : (ax == 1 ? R[2] * R[3] - R[0] * R[5] : R[0] * R[4] - R[1] * R[3]));
        if (!(fabsf(gna) < kFarGroup)) continue;
        const int m0 = pGStart[gi_], m1 = pGStart[gi_ + 1];
        const float nmem = (float)(m1 - m0);
        const int k = a.cls ? a.cls[(size_t)img * a.mReco + rep] : 0;
        long long* F = wa.accF + (size_t)k * volSize * 2;
        long long* T = wa.accT + (size_t)k * volSize;
        for (int p = threadIdx.x; p < a.nPxl; p += blockDim.x) {
            const int pi = a.iCol[p], pj = a.iRow[p];
            const float2 dv = a.datP[(size_t)img * a.nPxl + p];
            float cf = a.ctfP[(size_t)img * a.nPxl + p];
            float2 S = make_float2(0.f, 0.f);
            for (int i = m0; i < m1; i++) {   // sum of the members' phase ramps
                const size_t dm = (size_t)img * a.mReco + pOrd[i];
                const double tx = a.trans[2 * dm] - offx, ty = a.trans[2 * dm + 1] - offy;
                const float2 r = ramp_value((float)(-tx) / a.idim, (float)(-ty) / a.idim, pi, pj);
                S.x += r.x;
                S.y += r.y;
            }
            const float2 tv = cmul(dv, S);
            if (a.cSearch) cf = insert_ctf_search(a.attr, a.dfac, img, a.mReco, rep, a.pixelSize, a.idim, pi, pj);
            float vre = tv.x * cf, vim = tv.y * cf;
            vre = vre * 1.0f; vim = vim * 1.0f;
            vre = vre * wgt; vim = vim * wgt;
            const float tval = (pow2f_(cf) * 1.0f * wgt) * nmem;
            const int icp = pi * a.opf, irp = pj * a.opf;
            const float x = (float)(R[0] * icp + R[3] * irp), y = (float)(R[1] * icp + R[4] * irp), z = (float)(R[2] * icp + R[5] * irp);
            if (!coord_in_grid(x, y, z, P)) continue;
            TriCell cell;
            tri_cell(cell, x, y, z, P);
            if (cell.conj) vim = -vim;
#pragma unroll
            for (int kk = 0; kk < 2; kk++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int ii = 0; ii < 2; ii++) {
                        const float wv = cell.w[kk * 4 + jj * 2 + ii];
                        const long idx = cell.rowOff[kk][jj] + ii;
                        const long long t = __float2ll_rn((tval * wv) * gT);
                        if (t != 0) acc_add(F, T, idx, __float2ll_rn((vre * wv) * gF), __float2ll_rn((vim * wv) * gF), t);
                    }
        }
    }
}

// pixel-list position of every (iRow, iCol): table [idim][idim/2+1], -1 where the pixel is not listed
__global__ void k_pix_index(int* __restrict__ pixIndex, const int* __restrict__ iCol, const int* __restrict__ iRow,
                            int nPxl, int idim)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPxl) return;
    const int half = idim / 2;
    const int i = iCol[p], j = iRow[p];
    if (i >= 0 && i <= half && j >= -half && j < half) pixIndex[(j + half) * (half + 1) + i] = p;
}

// ---------------------------------------------------------------------------------------------
// RECONSTRUCTOR_NORMALISE_T_F: sf = 1/T[0]; T *= sf; F *= sf  (src/Reconstructor.cpp:2455-2476)
// ---------------------------------------------------------------------------------------------
__global__ void k_read_sf(const float* T, float* sf) { *sf = (float)(1.0 / (double)T[0]); }

__global__ __launch_bounds__(256) void k_scale_tf(float4* __restrict__ F4, float4* __restrict__ T4, size_t nF4, size_t nT4,
                                                  float* __restrict__ Ftail, float* __restrict__ Ttail, int nFtail,
                                                  int nTtail, const float* __restrict__ sfp)
{
    const float sf = *sfp;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nF4; i += stride) {
        float4 v = F4[i];
        v.x = v.x * sf; v.y = v.y * sf; v.z = v.z * sf; v.w = v.w * sf;
        F4[i] = v;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nT4; i += stride) {
        float4 v = T4[i];
        v.x = v.x * sf; v.y = v.y * sf; v.z = v.z * sf; v.w = v.w * sf;
        T4[i] = v;
    }
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < nFtail) Ftail[threadIdx.x] = Ftail[threadIdx.x] * sf;
        if ((int)threadIdx.x < nTtail) Ttail[threadIdx.x] = Ttail[threadIdx.x] * sf;
    }
}

// ---------------------------------------------------------------------------------------------
// SYMMETRIZE_FT: dst = src + sum_s VOL_TRANSFORM_MAT_FT(src, R_s, r) (gather, dst != src).
// One thread per stored voxel (i fastest -> coalesced dst writes).  Up to 64 symmetry matrices per launch
// in constant kernel arguments.
// ---------------------------------------------------------------------------------------------
struct SymMats {
    double m[24 * 9];
};

template <bool COMPLEX>
__global__ __launch_bounds__(256) void k_symmetrize(float* __restrict__ dst, const float* __restrict__ src,
                                                    const float* __restrict__ base, int P, SymMats sm, int nSym, double r2)
{
    const int nc = P / 2 + 1;
    const size_t n = (size_t)P * P * nc;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int i = (int)(e % nc);
    const int jw = (int)((e / nc) % P), kw = (int)(e / ((size_t)nc * P));
    const int j = jw >= P / 2 ? jw - P : jw, k = kw >= P / 2 ? kw - P : kw;
    const double nx = (double)i, ny = (double)j, nz = (double)k;
    float re, im = 0.f;
    if (COMPLEX) {
        re = base[2 * e];
        im = base[2 * e + 1];
    } else {
        re = base[e];
    }
    for (int s = 0; s < nSym; s++) {
        const double* m = sm.m + 9 * s;
        const double ox = m[0] * nx + m[3] * ny + m[6] * nz;
        const double oy = m[1] * nx + m[4] * ny + m[7] * nz;
        const double oz = m[2] * nx + m[5] * ny + m[8] * nz;
        if (ox * ox + oy * oy + oz * oz < r2) {
            const float x = (float)ox, y = (float)oy, z = (float)oz;
            if (coord_in_grid(x, y, z, P)) {
                if (COMPLEX) {
                    const float2 v = interp_ft(reinterpret_cast<const float2*>(src), P, x, y, z);
                    re = re + v.x;
                    im = im + v.y;
                } else {
                    re = re + interp_ft_real(src, P, x, y, z);
                }
            }
        }
    }
    if (COMPLEX) {
        dst[2 * e] = re;
        dst[2 * e + 1] = im;
    } else {
        dst[e] = re;
    }
}

}  // namespace thx

using namespace thx;

// running count of (image, group) pairs inserted on this device since the last thx_insert_groups_total(reset): one 8-byte
// device word per device, allocated on first use
static unsigned long long* group_counter()
{
    static unsigned long long* ctr[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!ctr[dev]) {
        if (hipMalloc(reinterpret_cast<void**>(&ctr[dev]), sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemset(ctr[dev], 0, sizeof(unsigned long long));
    }
    return ctr[dev];
}

struct thx_comm;
extern "C" int thx_comm_allreduce_max_f64(thx_comm* c, double* buf, size_t count, void* stream);

extern "C" {

size_t thx_insert_acc_bytes(int dim, int nK)
{
    return (size_t)dim * dim * (dim / 2 + 1) * (size_t)(nK > 0 ? nK : 1) * 3 * sizeof(long long);
}

int thx_insert_bounds_dev(float* bounds, const float* datP, const float* ctfP, int nPxl, int nImg, void* stream)
{
    if (nImg <= 0) return 0;
    THX_REQUIRE(bounds && datP && ctfP && nPxl > 0, "bad arguments");
    hipLaunchKernelGGL(k_insert_bounds, dim3(nImg), dim3(256), 0, as_stream(stream), reinterpret_cast<float2*>(bounds),
                       reinterpret_cast<const float2*>(datP), ctfP, nPxl);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_insert_scale_dev(int* gexp, const float* bounds, const float* w, int nImg, int mReco, int cSearch, long nImgHemi,
                         thx_comm* hemi, void* stream)
{
    THX_REQUIRE(gexp && (nImg <= 0 || (bounds && w)) && mReco > 0, "bad arguments");
    hipStream_t st = as_stream(stream);
    double* ext = reinterpret_cast<double*>(scratch(st, 11, 4 * sizeof(double)));
    THX_REQUIRE(ext, "device scratch allocation failed");
    hipLaunchKernelGGL(k_insert_scale, dim3(1), dim3(256), 0, st, ext, reinterpret_cast<const float2*>(bounds), w, nImg > 0 ? nImg : 0,
                       mReco, cSearch);
    THX_RC(thx_comm_allreduce_max_f64(hemi, ext, 4, st));
    int spare = 0;
    for (long n = (nImgHemi > nImg ? nImgHemi : nImg); n > 65536; n = (n + 1) / 2) spare++;
    hipLaunchKernelGGL(k_insert_scale_final, dim3(1), dim3(1), 0, st, gexp, ext, spare);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_insert_accumulate_dev(void* acc, const int* gexp, const float* bounds, double* O, int* counter, int dim, int nK,
                              const float* datP, const float* ctfP, const float* w, const double* rotMat, const double* trans,
                              const double* offS, const int* cls, const thx_ctf_attr* attr, const double* dfac, int cSearch,
                              float pixelSize, const int* iCol, const int* iRow, int opf, int nPxl, int mReco, int idim, int nImg,
                              void* stream)
{
    if (nImg <= 0 || mReco <= 0 || nPxl <= 0) return 0;
    THX_REQUIRE(acc && gexp && bounds && datP && ctfP && w && rotMat && trans && iCol && iRow, "NULL pointer");
    THX_REQUIRE(!cSearch || (attr && dfac), "cSearch needs attr and dfac");
    InsertArgs a;
    a.F = nullptr; a.T = nullptr; a.O = O; a.counter = counter; a.P = dim; a.nK = nK;
    a.datP = reinterpret_cast<const float2*>(datP); a.ctfP = ctfP; a.w = w; a.rotMat = rotMat; a.trans = trans;
    a.offS = offS; a.cls = cls; a.attr = attr; a.dfac = dfac; a.cSearch = cSearch; a.pixelSize = pixelSize;
    a.iCol = iCol; a.iRow = iRow; a.opf = opf; a.nPxl = nPxl; a.mReco = mReco; a.idim = idim;
    hipStream_t st = as_stream(stream);
    const int half = idim / 2;
    int* plan = reinterpret_cast<int*>(scratch(st, 3, (size_t)nImg * plan_stride(mReco) * sizeof(int)));
    THX_REQUIRE(plan, "device scratch allocation failed");
    hipLaunchKernelGGL(k_insert_plan, dim3(nImg), dim3(128), 5 * (size_t)mReco * sizeof(int), st, plan, rotMat, trans, cls, dfac,
                       cSearch, mReco);
    if (unsigned long long* gc = group_counter())
        hipLaunchKernelGGL(k_plan_groups, dim3((nImg + 255) / 256), dim3(256), 0, st, gc, plan, nImg, plan_stride(mReco));
    const size_t volSize = (size_t)dim * dim * (dim / 2 + 1) * (size_t)(nK > 0 ? nK : 1);
    long long* accF = reinterpret_cast<long long*>(acc);
    long long* accT = accF + 2 * volSize;
    // the production form: samples binned by brick of the volume, bricks accumulated over all the images of a chunk
    // (thx_insert_sort.hip); THX_INSERT=win keeps the per-image window kernel below for A/B runs
    if (!knobs().insertWin) return insert_sorted(st, a, plan, gexp, accF, accT, nImg);
    const size_t tb = (size_t)idim * (half + 1) * sizeof(int);
    int* pixIndex = reinterpret_cast<int*>(scratch(st, 0, tb));
    THX_REQUIRE(pixIndex, "device scratch allocation failed");
    THX_CHECK(hipMemsetAsync(pixIndex, 0xFF, tb, st));
    hipLaunchKernelGGL(k_pix_index, dim3((nPxl + 255) / 256), dim3(256), 0, st, pixIndex, iCol, iRow, nPxl, idim);
    const size_t ldsWin = 3 * (size_t)kWinVox * sizeof(int) + (size_t)mReco * 6 * sizeof(double) +
                          ((size_t)(mReco + 1) + mReco + 2 * mReco) * sizeof(int) + 2 * (size_t)mReco * sizeof(float) +
                          4 * (size_t)mReco * sizeof(short) + 2 * (size_t)mReco * sizeof(float) + 32 +
                          (size_t)kIPix * kIPix * sizeof(float4) + 2 * (size_t)kMaxU * kIPix * sizeof(float2) +
                          (size_t)(kWinThreads / 64) * kWinQueue * sizeof(int);
    THX_REQUIRE(ldsWin <= 160 * 1024, "mReco too large for the LDS draw table");
    THX_REQUIRE(mReco <= 1024 && idim <= 2048, "window insertion packs (pixel, group) into 11 + 11 + 10 bits");
    THX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_insert_win), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ldsWin));
    for (int l0 = 0; l0 < nImg; l0 += 65535) {
        const int nl = nImg - l0 < 65535 ? nImg - l0 : 65535;
        InsertArgs b = a;
        b.datP += (size_t)l0 * nPxl; b.ctfP += (size_t)l0 * nPxl; b.w += l0;
        b.rotMat += (size_t)l0 * mReco * 9; b.trans += (size_t)l0 * mReco * 2;
        if (b.offS) b.offS += (size_t)l0 * 2;
        if (b.cls) b.cls += (size_t)l0 * mReco;
        if (b.attr) b.attr += l0;
        if (b.dfac) b.dfac += (size_t)l0 * mReco;
        InsertWinArgs wa;
        wa.a = b; wa.pixIndex = pixIndex; wa.plan = plan + (size_t)l0 * plan_stride(mReco);
        wa.bounds = reinterpret_cast<const float2*>(bounds) + l0;
        wa.accF = accF; wa.accT = accT; wa.gexp = gexp;
        wa.minQuanta = knobs().minQuanta >= 0.f ? knobs().minQuanta : kMinQuanta;
        const int rc = half * opf + 3;
        const int hw = (rc + kWd - 1) / kWd;
        wa.nW = 2 * hw;
        wa.pOrg = -hw * kWd;
        wa.rMax2 = (float)(half * opf + 2) * (float)(half * opf + 2);
        wa.debug = kWinProfiling ? knobs().insertDebug : 0;
        hipLaunchKernelGGL(k_insert_win, dim3(wa.nW, nl), dim3(kWinThreads), ldsWin, st, wa);
        hipLaunchKernelGGL(k_insert_far, dim3(nl), dim3(256), 0, st, wa);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_insert_groups_total(unsigned long long* out, int reset, void* stream)
{
    THX_REQUIRE(out, "NULL pointer");
    unsigned long long* gc = group_counter();
    THX_REQUIRE(gc, "no device counter");
    hipStream_t st = as_stream(stream);
    THX_CHECK(hipMemcpyAsync(out, gc, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if (reset) THX_CHECK(hipMemsetAsync(gc, 0, sizeof(unsigned long long), st));
    THX_CHECK(hipStreamSynchronize(st));
    return 0;
}

int thx_insert_finish_dev(float* F, float* T, const void* acc, const int* gexp, int dim, int nK, void* stream)
{
    THX_REQUIRE(F && T && acc && gexp && dim > 0, "bad arguments");
    const size_t nVox = (size_t)dim * dim * (dim / 2 + 1) * (size_t)(nK > 0 ? nK : 1);
    const long long* accF = reinterpret_cast<const long long*>(acc);
    hipLaunchKernelGGL(k_insert_convert, dim3((unsigned)((nVox + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float2*>(F), T, accF, accF + 2 * nVox, gexp, nVox);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_insert_dev(float* F, float* T, double* O, int* counter, int dim, int nK, const float* datP,
                   const float* ctfP, const float* w, const double* rotMat, const double* trans, const double* offS,
                   const int* cls, const thx_ctf_attr* attr, const double* dfac, int cSearch, float pixelSize,
                   const int* iCol, const int* iRow, int opf, int nPxl, int mReco, int idim, int nImg, void* stream)
{
    if (nImg <= 0 || mReco <= 0 || nPxl <= 0) return 0;
    THX_REQUIRE(F && T && datP && ctfP && w && rotMat && trans && iCol && iRow, "NULL pointer");
    THX_REQUIRE(!cSearch || (attr && dfac), "cSearch needs attr and dfac");
    hipStream_t st = as_stream(stream);
    if (knobs().insertPlain) {   // THX_INSERT_PLAIN=1 (read once at load, thx::knobs): the plain float-atomic form k_insert for A/B runs
        InsertArgs a;
        a.F = reinterpret_cast<float2*>(F); a.T = T; a.O = O; a.counter = counter; a.P = dim; a.nK = nK;
        a.datP = reinterpret_cast<const float2*>(datP); a.ctfP = ctfP; a.w = w; a.rotMat = rotMat; a.trans = trans;
        a.offS = offS; a.cls = cls; a.attr = attr; a.dfac = dfac; a.cSearch = cSearch; a.pixelSize = pixelSize;
        a.iCol = iCol; a.iRow = iRow; a.opf = opf; a.nPxl = nPxl; a.mReco = mReco; a.idim = idim;
        for (int l0 = 0; l0 < nImg; l0 += 65535) {
            const int nl = nImg - l0 < 65535 ? nImg - l0 : 65535;
            InsertArgs b = a;
            b.datP += (size_t)l0 * nPxl; b.ctfP += (size_t)l0 * nPxl; b.w += l0;
            b.rotMat += (size_t)l0 * mReco * 9; b.trans += (size_t)l0 * mReco * 2;
            if (b.offS) b.offS += (size_t)l0 * 2;
            if (b.cls) b.cls += (size_t)l0 * mReco;
            if (b.attr) b.attr += l0;
            if (b.dfac) b.dfac += (size_t)l0 * mReco;
            hipLaunchKernelGGL(k_insert, dim3((nPxl + 255) / 256, nl), dim3(256), 0, st, b);
        }
        THX_LAUNCH_CHECK();
        return 0;
    }
    // one-call form of the session below: bounds -> quanta -> zeroed 64-bit accumulators -> accumulate -> F / T += them
    float* bounds = reinterpret_cast<float*>(scratch(st, 6, (size_t)nImg * 2 * sizeof(float) + 16));
    THX_REQUIRE(bounds, "device scratch allocation failed");
    int* gexp = reinterpret_cast<int*>(bounds + 2 * (size_t)nImg);
    void* acc = scratch(st, 10, thx_insert_acc_bytes(dim, nK));
    THX_REQUIRE(acc, "device scratch allocation failed (fixed-point accumulators)");
    THX_RC(thx_insert_bounds_dev(bounds, datP, ctfP, nPxl, nImg, stream));
    THX_RC(thx_insert_scale_dev(gexp, bounds, w, nImg, mReco, cSearch, nImg, nullptr, stream));
    THX_CHECK(hipMemsetAsync(acc, 0, thx_insert_acc_bytes(dim, nK), st));
    THX_RC(thx_insert_accumulate_dev(acc, gexp, bounds, O, counter, dim, nK, datP, ctfP, w, rotMat, trans, offS, cls, attr, dfac, cSearch,
                                     pixelSize, iCol, iRow, opf, nPxl, mReco, idim, nImg, stream));
    THX_RC(thx_insert_finish_dev(F, T, acc, gexp, dim, nK, stream));
    return 0;
}

int thx_normalise_tf_dev(float* F, float* T, int dim, void* stream)
{
    THX_REQUIRE(F && T, "NULL pointer");
    hipStream_t st = as_stream(stream);
    const size_t n = (size_t)dim * dim * (dim / 2 + 1);
    float* sf = nullptr;
    sf = reinterpret_cast<float*>(scratch(st, 1, sizeof(float)));
    THX_REQUIRE(sf, "device scratch allocation failed");
    hipLaunchKernelGGL(k_read_sf, dim3(1), dim3(1), 0, st, T, sf);
    const size_t nF = 2 * n, nT = n;
    const size_t nF4 = nF / 4, nT4 = nT / 4;
    hipLaunchKernelGGL(k_scale_tf, dim3(2048), dim3(256), 0, st, reinterpret_cast<float4*>(F), reinterpret_cast<float4*>(T),
                       nF4, nT4, F + nF4 * 4, T + nT4 * 4, (int)(nF - nF4 * 4), (int)(nT - nT4 * 4), sf);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_symmetrize_dev(float* dst, const float* src, int dim, int isComplex, const double* symMat_host, int nSym,
                       double r, void* stream)
{
    THX_REQUIRE(dst && src && dst != src, "dst and src must be distinct non-NULL volumes");
    hipStream_t st = as_stream(stream);
    const size_t n = (size_t)dim * dim * (dim / 2 + 1);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (nSym == 0) {
        THX_CHECK(hipMemcpyAsync(dst, src, n * (isComplex ? 2 : 1) * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    // result = src + sum over all elements; processed in batches of 24 matrices, accumulating into dst
    for (int s0 = 0; s0 < nSym; s0 += 24) {
        const int ns = nSym - s0 < 24 ? nSym - s0 : 24;
        SymMats sm;
        memcpy(sm.m, symMat_host + 9 * (size_t)s0, sizeof(double) * 9 * ns);
        const float* base = s0 == 0 ? src : dst;
        if (isComplex)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_symmetrize<true>), dim3(blocks), dim3(256), 0, st, dst, src, base, dim, sm, ns, r * r);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_symmetrize<false>), dim3(blocks), dim3(256), 0, st, dst, src, base, dim, sm, ns, r * r);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
