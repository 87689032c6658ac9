// thx_insert.h -- declarations shared by the two insertion forms: the per-image window kernel (thx_mstep.hip) and the
// brick-sorted form (thx_insert_sort.hip).  Reference behaviour: src/Optimiser.cpp:7038-7241, src/Reconstructor.cpp:782-863,
// src/Image/Volume.cpp:565-712.  gfx950 only.
#pragma once
#include "thx_common.h"

namespace thx {

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

// Insert plan of one image (k_insert_plan): [0] G, [1] U, gStart[mReco+1], ord[mReco], uid[mReco], gRep[mReco], tRep[mReco]
__host__ __device__ inline int plan_stride(int mReco) { return 5 * mReco + 3; }

__device__ __attribute__((noinline)) inline float insert_ctf_search(const thx_ctf_attr* attr, const double* dfac, int img, int mReco, int rep,
                                                             float pixelSize, int idim, int pi, int pj)
{
    const CtfConst cc = ctf_const(attr[img], dfac[(size_t)img * mReco + rep]);
    return ctf_value(cc, pixelSize, idim, idim, pi, pj);
}

// The volume accumulators of the insertion are 64-bit FIXED POINT (quanta 2^-E_F / 2^-E_T of one unit, one pair of exponents
// per session, k_insert_scale): integer atomic adds commute, so F and T come out bit-identical run to run whatever the order
// in which workgroups flush.  k_insert_convert adds the accumulators to the caller's float volumes afterwards.
__device__ __forceinline__ void acc_add(long long* F, long long* T, long gi, long long re, long long im, long long tt)
{
    atomicAdd(reinterpret_cast<unsigned long long*>(F + 2 * gi), (unsigned long long)re);
    atomicAdd(reinterpret_cast<unsigned long long*>(F + 2 * gi + 1), (unsigned long long)im);
    atomicAdd(reinterpret_cast<unsigned long long*>(T + gi), (unsigned long long)tt);
}

// Brick-sorted insertion of nImg images (thx_insert_sort.hip): every (listed pixel, group of draws) sample is computed once,
// binned by the 16 x 8 x 8 brick of the volume its trilinear cell starts in, and the bricks are accumulated in LDS over ALL the
// images of a chunk before they are flushed.  `a` carries the chunk-independent arguments (image-indexed pointers at image 0).
int insert_sorted(hipStream_t st, const InsertArgs& a, const int* plan, const int* gexp, long long* accF, long long* accT, int nImg);

}  // namespace thx
