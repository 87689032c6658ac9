// thx_mstep.hip -- M-step kernels: direct Fourier insertion (back-projection), T/F normalisation and
// symmetrisation.  Reference behaviour: src/Optimiser.cpp:7038-7241 (HOT LOOP C),
// src/Reconstructor.cpp:407-422,782-863,2455-2476,2676-2690, src/Image/Volume.cpp:340-375,565-712,
// include/Geometry/Transformation.h:105-131,170-194.  gfx950 only.
#include "thx_common.h"

namespace thx {

// ---------------------------------------------------------------------------------------------
// Insertion.  One thread per listed pixel of one image; the image row (dat, ctf) is read ONCE and
// kept in registers while the block walks all mReco draws (rotation, shift[, defocus]) of that image,
// whose parameters are wave-uniform scalar loads.  Each pixel-sample is a trilinear scatter of
// w*ctf*img into F (8 x complex) and w*ctf^2 into T (8 x real) with hardware fp32 atomics
// (global_atomic_add_f32): the MI355X counterpart of the reference's `#pragma omp atomic`
// (src/Image/Volume.cpp:584-587,676-677).  grid (ceil(nPxl/256), nImg).
// ---------------------------------------------------------------------------------------------
struct InsertArgs {
    float2* F;
    float* T;
    double* O;
    int* counter;
    int P, nK;
    const float2* datP;
    const float* ctfP;
    const float* w;
    const double* rotMat;
    const double* trans;
    const double* offS;
    const int* cls;
    const thx_ctf_attr* attr;
    const double* dfac;
    int cSearch;
    float pixelSize;
    const int* iCol;
    const int* iRow;
    int opf, nPxl, mReco, idim;
};

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

extern "C" {

int thx_insert_dev(float* F, float* T, double* O, int* counter, int dim, int nK, const float* datP,
                   const float* ctfP, const float* w, const double* rotMat, const double* trans, const double* offS,
                   const int* cls, const thx_ctf_attr* attr, const double* dfac, int cSearch, float pixelSize,
                   const int* iCol, const int* iRow, int opf, int nPxl, int mReco, int idim, int nImg, void* stream)
{
    if (nImg <= 0 || mReco <= 0 || nPxl <= 0) return 0;
    THX_REQUIRE(F && T && datP && ctfP && w && rotMat && trans && iCol && iRow, "NULL pointer");
    THX_REQUIRE(!cSearch || (attr && dfac), "cSearch needs attr and dfac");
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
        hipLaunchKernelGGL(k_insert, dim3((nPxl + 255) / 256, nl), dim3(256), 0, as_stream(stream), b);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_normalise_tf_dev(float* F, float* T, int dim, void* stream)
{
    THX_REQUIRE(F && T, "NULL pointer");
    hipStream_t st = as_stream(stream);
    const size_t n = (size_t)dim * dim * (dim / 2 + 1);
    float* sf = nullptr;
    THX_CHECK(hipMallocAsync(reinterpret_cast<void**>(&sf), sizeof(float), st));
    hipLaunchKernelGGL(k_read_sf, dim3(1), dim3(1), 0, st, T, sf);
    const size_t nF = 2 * n, nT = n;
    const size_t nF4 = nF / 4, nT4 = nT / 4;
    hipLaunchKernelGGL(k_scale_tf, dim3(2048), dim3(256), 0, st, reinterpret_cast<float4*>(F), reinterpret_cast<float4*>(T),
                       nF4, nT4, F + nF4 * 4, T + nT4 * 4, (int)(nF - nF4 * 4), (int)(nT - nT4 * 4), sf);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipFreeAsync(sf, st));
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
