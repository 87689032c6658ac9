// thx_mstep.hip -- M-step kernels: direct Fourier insertion (back-projection), T/F normalisation and
// symmetrisation.  Reference behaviour: src/Optimiser.cpp:7038-7241 (HOT LOOP C),
// src/Reconstructor.cpp:407-422,782-863,2455-2476,2676-2690, src/Image/Volume.cpp:340-375,565-712,
// include/Geometry/Transformation.h:105-131,170-194.  gfx950 only.
#include <mutex>

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
// Production insertion: the brick-sorted form of thx_insert_sort.hip (k_bin / segment sort / k_acc); k_insert above is the
// plain reference form with float atomics.  What this file keeps of it: the insert plan, the session's fixed-point scale
// (k_insert_bounds / k_insert_scale) and the conversion of the 64-bit accumulators.
//
// Measured on MI355X (tools/probes/atomic_bench.hip, tools/probes/lds_atomic_bench.hip): fp32 global atomics retire ~19 G
// *transactions*/s chip-wide (one per XCD per clock), so scattered 4-byte atomics (k_insert) reach 2 % of the HBM roofline;
// ds_add_f32 retires 0.33 lanes/clk/CU whatever the address pattern, ds_add_u32 6.2, ds_add_u64 4.3.  Hence: accumulate in
// LDS bricks in FIXED POINT (order-independent integer sums), flush a brick along the volume's contiguous x axis.
// (Two earlier forms -- a pixel-tile brick per image and a sheared volume window per image, k_insert_win -- are in the
// history of this file and in DESIGN.md 4.2: 5x / 2.5x slower than the sorted form on a particle filter's clouds.)
// ---------------------------------------------------------------------------------------------

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

// sum of the group counts of a launch's images (plan[0] of each) into a running device counter: the insertion holds one
// record (24 LDS adds) per (listed pixel, GROUP), not per draw
__global__ void k_plan_groups(unsigned long long* __restrict__ total, const int* __restrict__ plan, int nImg, int stride)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long g = l < nImg ? (unsigned long long)plan[(size_t)l * stride] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
    if ((threadIdx.x & 63) == 0 && g) atomicAdd(total, g);
}

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

// exponents of an image's own quanta: the largest powers of two with |F term| * 2^eF <= 2^(31 - lg), T * 2^eT <= 2^(32 - lg)
// (lg = bits of 2 mReco: for one rotation the trilinear weights a voxel collects sum to <= 1, so an image adds less than 2^31
// of its own quanta to any voxel)
__device__ __forceinline__ void image_exponents(float boundF, float boundT, int lg, int& eF, int& eT)
{
    int ebF = 0, ebT = 0;
    (void)frexpf(boundF, &ebF);   // bound = m 2^eb, m in [0.5, 1)
    (void)frexpf(boundT, &ebT);
    eF = 31 - lg - ebF;
    eT = 32 - lg - ebT;
}

// one session-wide pair of exponents: 7 bits below the finest image quantum, the finest being taken at most 2^8 finer than
// that of the image with the largest bound -- every voxel term of every image is rounded ONCE, to this quantum (between 2^-30
// and 2^-38 of its image's largest possible term; thx_insert_sort.hip).  Headroom: 65535 images x 2^31 x 2^15 < 2^63.
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
    static std::mutex mtx;   // (the insertion entry points are called from several host threads, as the library's other caches are)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mtx);
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
    int* plan = reinterpret_cast<int*>(scratch(st, 3, (size_t)nImg * plan_stride(mReco) * sizeof(int)));
    THX_REQUIRE(plan, "device scratch allocation failed");
    // (5 ints of dynamic LDS per draw: above 64 KB -- mReco > 3 276 -- the launch needs the opt-in limit raised)
    THX_REQUIRE(mReco <= 8000, "at most 8 000 draws per image (the insertion plan sorts them in 160 KB of LDS)");
    const size_t planLds = 5 * (size_t)mReco * sizeof(int);
    if (planLds > 48 * 1024)
        THX_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_insert_plan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)planLds));
    hipLaunchKernelGGL(k_insert_plan, dim3(nImg), dim3(128), planLds, st, plan, rotMat, trans, cls, dfac, cSearch, mReco);
    THX_LAUNCH_CHECK();   // (a plan that did not launch must not reach k_plan_counts / k_bin)
    if (unsigned long long* gc = group_counter())
        hipLaunchKernelGGL(k_plan_groups, dim3((nImg + 255) / 256), dim3(256), 0, st, gc, plan, nImg, plan_stride(mReco));
    const size_t volSize = (size_t)dim * dim * (dim / 2 + 1) * (size_t)(nK > 0 ? nK : 1);
    long long* accF = reinterpret_cast<long long*>(acc);
    long long* accT = accF + 2 * volSize;
    // samples binned by brick of the volume, bricks accumulated over all the images of a chunk (thx_insert_sort.hip)
    return insert_sorted(st, a, plan, gexp, accF, accT, nImg);
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
