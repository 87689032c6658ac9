// thx_insert_sort.hip -- brick-sorted Fourier insertion (back-projection), the production form of HOT LOOP C.
// Reference behaviour: src/Optimiser.cpp:7038-7241 (the loop over images and draws), src/Reconstructor.cpp:782-863
// (insertP: value * ctf * w into F, ctf^2 * w into T), src/Image/Volume.cpp:565-712 (addFTHalf: trilinear scatter on the
// half-Hermitian grid), include/Functions/Interpolation.h:152-200 (the eight weights).  gfx950 only.
//
// Why this shape.  The per-image window kernel of rounds 1 - 2 (k_insert_win, in this repository's history) kept the IMAGE side
// coalesced and paid on the VOLUME side: one (window, slab) step of one image held ~900 samples between two barriers and one
// flush of a 48 KB brick, and finding the samples of a window cost 2.5 float tests per hit -- 0.12 of the LDS-add rate,
// instruction-bound.  Here the two sides are decoupled by a sort through HBM, 28 bytes per sample each way:
//   k_bin  (image-major): one thread per listed pixel, all groups of the image's draws; every sample is computed ONCE at full
//          lane occupancy -- exact position, cell, fractional offsets, value -- and written as a record into the segment of its
//          brick.  A segment = the records one (256-pixel region, 8 groups) pass sends to one 16 x 8 x 8 brick of cell origins;
//          the pass counts per brick in an LDS hash table, lays its segments out from its STATIC place in the record buffer (no
//          global atomic per pass: one returning atomic on one address per pass serialised the chip), recomputes and scatters;
//          the descriptors collect in LDS and take table space with one atomic per workgroup.
//   sort   the segment descriptors (not the records) by brick: rocPRIM radix sort of ~1/130 of the record count.
//   k_acc  (brick-major): a workgroup takes ~16 k records of consecutive bricks, accumulates each brick's 17 x 9 x 9 voxels in
//          LDS as 64-bit integers over ALL the images of the chunk, and flushes a brick once.
// No window geometry, no shear, no candidate tests, no far-group special case: a sample's brick is a shift of its cell origin.
// Bounds: k_acc the LDS atomic unit (24 ds_add_u64 per record: 0.73 of its rate); k_bin writes its records at about
// the write rate of HBM but is within 12 % of its time without the stores (instruction issue and LDS latency).
//
// Arithmetic.  Every voxel term is rounded ONCE, to the session's 64-bit quanta (k_insert_scale: 2^-E_F, 2^-E_T):
// re = rint((vre 2^E_F) wv), t = rint((tval 2^E_T) wv); a term whose T part rounds to zero is dropped whole (F and T travel
// together: T = 0 under F != 0 lets the gridding weights explode).  That is the rule the window kernel applied to its
// sub-quantum terms, here applied to all of them, so a term carries 7 more bits than a brick term of the window kernel did.
// Integer sums commute: F and T are bit-identical from run to run, for any chunking, and across ranks.
#include "thx_insert.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <mutex>
#include <vector>

namespace thx {

#ifndef THX_BIN_KEEP
#define THX_BIN_KEEP 2   // k_bin between its count and scatter phases keeps, per sample, in registers -- 2: the hash slot of its brick (no
                         // hash_find; 125 VGPRs with 12 ramp slots: 4 waves per SIMD); 1: slot and geometry (no second sample_geom either,
                         // 160 - 168 VGPRs: 3 waves per SIMD); 0: nothing (recompute, look up).  Insertion stage of a 20 000-particle
                         // iteration, same box: 312 / 307 / 303 ms for 0 / 1 / 2 at 12 ramp slots
#endif
#ifndef THX_BRICK_LX
#define THX_BRICK_LX 4
#endif
#ifndef THX_BRICK_LY
#define THX_BRICK_LY 3    // 16 x 8 x 8 cell origins: 33 KB of LDS per brick, FOUR workgroups of k_acc per CU -- measured against 16 x 16 x 8 /
                          // two per CU (-3 .. -5 % on the insertion), 16 x 16 x 4, 8 x 16 x 8 (equal), 16 x 4 x 8, 8 x 8 x 8, 16 x 8 x 4 (all slower)
#endif
#ifndef THX_BRICK_LZ
#define THX_BRICK_LZ 3
#endif
#ifndef THX_ACC_WGS
#define THX_ACC_WGS 4    // workgroups of k_acc per CU the launch bounds ask for (their LDS bricks must fit 160 KB together)
#endif
static_assert(THX_BRICK_LX <= 4 && THX_BRICK_LY <= 4 && THX_BRICK_LZ <= 3, "a record's cell field holds 4 + 4 + 3 bits");
constexpr int kBLx = THX_BRICK_LX, kBLy = THX_BRICK_LY, kBLz = THX_BRICK_LZ;   // log2 of the brick edges in cell origins
constexpr int kBx = 1 << kBLx, kBy = 1 << kBLy, kBz = 1 << kBLz;  // 16 x 8 x 8
constexpr int kVx = kBx + 1, kVy = kBy + 1, kVz = kBz + 1;        // voxels a brick's cells reach
constexpr int kBrickVox = kVx * kVy * kVz;                        // 1377 x 24 B = 33 KB of LDS: four workgroups per CU
constexpr int kBinThreads = 256;                                  // pixels per region
constexpr int kPassGroups = 8;                                    // groups per pass of k_bin
constexpr int kHash = kBinThreads * kPassGroups;                  // >= the distinct bricks of a pass, whatever the input
#ifndef THX_RAMP_U
#define THX_RAMP_U 12   // (mLT = 9 in the reference's configuration; images with more unique shifts take the member-by-member path)
#endif
constexpr int kRampU = THX_RAMP_U;                                        // unique shifts whose ramps a thread keeps in registers
constexpr int kAccThreads = 512;
#ifndef THX_ACC_SPAN
#define THX_ACC_SPAN 16384
#endif
constexpr unsigned kAccSpan = THX_ACC_SPAN;                              // records per workgroup of k_acc
constexpr int kAccStage = kAccThreads;                             // segment descriptors staged in LDS at a time (one per thread)

struct BinArgs {
    InsertArgs a;               // image-indexed pointers at the chunk's first image
    const int* plan;            // k_insert_plan, at the chunk's first image
    const int* gexp;            // [2]: E_F, E_T
    const unsigned* groupsBefore; // [images of the chunk]: groups of the chunk's earlier images (the records' static layout)
    long long* accF;            // [nK][vol][2]
    long long* accT;            // [nK][vol]
    uint4* recA;                // [capR]: xd, yd, zd (float bits), cell
    float* recB;                // [capR][3]: vre 2^E_F, vim 2^E_F (conjugated where folded), tval 2^E_T
    unsigned* segKey;           // [capS] brick id of the segment
    unsigned long long* segVal; // [capS] first record | count << 32
    unsigned* counter;          // segment descriptors reserved so far
    unsigned capS;
    int nBx, nBy, nBz;
};

// float -> int64, round to nearest even, |q| < 2^51: the sum with 1.5 x 2^52 holds the integer in its low mantissa bits
__device__ __forceinline__ long long f2ll_magic(float q)
{
    const double d = (double)q + 6755399441055744.0;
    return __double_as_longlong(d) - 0x4338000000000000LL;
}

struct SampleGeom {
    float xd, yd, zd;
    int key;
    unsigned cell;
    bool conj;
};

// position of pixel (icp, irp) (padded units) under the rotation whose first two columns are R[0..5]: the reference's
// arithmetic (double products and sum, narrowed once; src/Reconstructor.cpp:805-811), the Hermitian fold and the cell
__device__ __forceinline__ bool sample_geom(const double* R, int icp, int irp, int P, int cls, int nBx, int nBy, int nBz, SampleGeom& s)
{
    float x = (float)(R[0] * icp + R[3] * irp);
    float y = (float)(R[1] * icp + R[4] * irp);
    float z = (float)(R[2] * icp + R[5] * irp);
    if (!coord_in_grid(x, y, z, P)) return false;
    s.conj = false;
    if (!(x >= 0.0f)) { x *= -1.0f; y *= -1.0f; z *= -1.0f; s.conj = true; }
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    const int X0 = (int)fx, yb = (int)fy + P / 2, zb = (int)fz + P / 2;
    s.xd = x - fx; s.yd = y - fy; s.zd = z - fz;
    s.key = ((cls * nBz + (zb >> kBLz)) * nBy + (yb >> kBLy)) * nBx + (X0 >> kBLx);
    s.cell = (unsigned)(X0 & (kBx - 1)) | ((unsigned)(yb & (kBy - 1)) << 4) | ((unsigned)(zb & (kBz - 1)) << 8);
    return true;
}

__device__ __forceinline__ unsigned hash_of(int key) { return ((unsigned)key * 2654435761u) >> (32 - 11); }
static_assert(kHash == 2048, "hash_of keeps 11 bits");

__device__ __forceinline__ int hash_insert(int* hKey, int key)
{
    unsigned h = hash_of(key);
    for (;;) {
        const int prev = atomicCAS(&hKey[h], -1, key);
        if (prev == -1 || prev == key) return (int)h;
        h = (h + 1) & (kHash - 1);
    }
}
__device__ __forceinline__ int hash_find(const int* hKey, int key)
{
    unsigned h = hash_of(key);
    while (hKey[h] != key) h = (h + 1) & (kHash - 1);
    return (int)h;
}

// more unique shifts than a thread keeps ramps for: the members' ramps one by one (translate(), src/Image/ImageFunctions.cpp:243-251)
__device__ __attribute__((noinline)) float2 ramp_sum_members(const double* trans, const int* pOrd, size_t dm0, int m0, int m1, double offx,
                                                             double offy, int idim, int pi, int pj)
{
    float2 S = make_float2(0.f, 0.f);
    for (int i = m0; i < m1; i++) {
        const size_t dm = dm0 + pOrd[i];
        const double tx = trans[2 * dm] - offx, ty = trans[2 * dm + 1] - offy;
        const float2 r = ramp_value((float)(-tx) / idim, (float)(-ty) / idim, pi, pj);
        S.x += r.x;
        S.y += r.y;
    }
    return S;
}

// the records of one segment straight into the 64-bit volume accumulators: what k_acc does through LDS, term for term.  Taken
// only by a workgroup of k_bin whose segment descriptors no longer fit the chunk's table (capS), for the segments it holds.
__device__ __forceinline__ void segment_direct(long long* accF, long long* accT, int P, int nBx, int nBy, int nBz, unsigned key,
                                                         const uint4* recA, const float* recB, unsigned off, unsigned cnt)
{
    const long nc = P / 2 + 1;
    const size_t volSize = (size_t)P * P * (P / 2 + 1);
    const int bx = (int)(key % (unsigned)nBx), by = (int)((key / (unsigned)nBx) % (unsigned)nBy);
    const int bzk = (int)(key / ((unsigned)nBx * (unsigned)nBy));
    const int bz = bzk % nBz, cls = bzk / nBz;
    long long* F = accF + (size_t)cls * volSize * 2;
    long long* T = accT + (size_t)cls * volSize;
    for (unsigned r = threadIdx.x; r < cnt; r += blockDim.x) {
        const uint4 ra = recA[off + r];
        const float* rb = recB + 3 * (size_t)(off + r);
        const float vreS = rb[0], vimS = rb[1], tvalS = rb[2];
        const float xd = __uint_as_float(ra.x), yd = __uint_as_float(ra.y), zd = __uint_as_float(ra.z);
        const float vx[2] = {1.0f - xd, xd}, vy[2] = {1.0f - yd, yd}, vz[2] = {1.0f - zd, zd};
        const int X0 = bx * kBx + (int)(ra.w & (kBx - 1)), Y0 = by * kBy + (int)((ra.w >> 4) & (kBy - 1)) - P / 2,
                  Z0 = bz * kBz + (int)((ra.w >> 8) & (kBz - 1)) - P / 2;
        for (int v = 0; v < 8; v++) {
            const int ii = v & 1, jj = (v >> 1) & 1, kk = v >> 2;
            const float wv = vx[ii] * vy[jj] * vz[kk];
            const long long t = f2ll_magic(tvalS * wv);
            if (t == 0) continue;
            const int X = X0 + ii, Y = Y0 + jj, Z = Z0 + kk;
            const long gi = ((long)(Z >= 0 ? Z : Z + P) * P + (Y >= 0 ? Y : Y + P)) * nc + X;
            acc_add(F, T, gi, f2ll_magic(vreS * wv), f2ll_magic(vimS * wv), t);
        }
    }
}

// descriptors a workgroup of k_bin collects in LDS before it reserves table space for them with one global atomic
constexpr int kSegList = kHash;   // a pass may touch up to 8 x 256 bricks

// the workgroup's collected descriptors -> the chunk's table (or, when the table is full, their records straight into the volume)
__device__ __forceinline__ void bin_flush_list(const BinArgs& b, const unsigned* lKey, const unsigned* lVal, unsigned regionBase, int n,
                                               unsigned* sBase)
{
    __syncthreads();   // the list is complete, and the workgroup's record stores are visible to all of it
    if (n == 0) return;
    if (threadIdx.x == 0) *sBase = atomicAdd(b.counter, (unsigned)n);
    __syncthreads();
    const unsigned base = *sBase;
    if ((unsigned long long)base + (unsigned long long)n <= (unsigned long long)b.capS) {
        for (int i = threadIdx.x; i < n; i += kBinThreads) {
            b.segKey[base + i] = lKey[i];
            b.segVal[base + i] = (unsigned long long)(regionBase + (lVal[i] & 0xFFFFFu)) | ((unsigned long long)(lVal[i] >> 20) << 32);
        }
    } else {
        // (what it did reserve inside the table becomes holes: they sort behind every brick and hold nothing)
        for (unsigned i = base + threadIdx.x; i < b.capS; i += kBinThreads) { b.segKey[i] = 0xFFFFFFFFu; b.segVal[i] = 0ull; }
        for (int i = 0; i < n; i++)
            segment_direct(b.accF, b.accT, b.a.P, b.nBx, b.nBy, b.nBz, lKey[i], b.recA, b.recB, regionBase + (lVal[i] & 0xFFFFFu), lVal[i] >> 20);
    }
    __syncthreads();
}

// grid (ceil(nPxl / 256), images of the chunk).  Records have a STATIC place: image l of the chunk owns
// [groupsBefore[l], groupsBefore[l] + G_l) x nRegion x 256 records, region r the G_l x 256 of them from r G_l 256 on, the pass
// that starts with group g0 the 8 x 256 from g0 256 on, and the pass's segments are packed at the front of that span.
// CS: CTF search (a CTF value per sample instead of the image's row); SLOWU: some image of the launch has more unique shifts
// than a thread keeps ramps for -- both compile-time, so that the common instance carries neither path's registers.
template <bool CS, bool SLOWU>
__global__ __launch_bounds__(kBinThreads) void k_bin(BinArgs b)
{
    const InsertArgs& a = b.a;
    __shared__ int hKey[kHash];
    __shared__ int hCnt[kHash];          // samples per slot (count), then the slot's next record (scatter)
    __shared__ unsigned lKey[kSegList];
    __shared__ unsigned lVal[kSegList];   // record offset from the region's base (20 bits) | count << 20
    __shared__ double sR[kPassGroups][6];
    __shared__ int sCntI[kPassGroups][kRampU];
    __shared__ int sCls[kPassGroups], sRep[kPassGroups], sM0[kPassGroups + 1];
    __shared__ int sWS[kBinThreads / 64], sWN[kBinThreads / 64];
    __shared__ unsigned sFlushBase;

    const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = blockIdx.x * kBinThreads + tid;
    const bool listed = p < a.nPxl;
    const int P = a.P;
    const size_t dm0 = (size_t)img * a.mReco;
    const double offx = a.offS ? a.offS[2 * img] : 0.0, offy = a.offS ? a.offS[2 * img + 1] : 0.0;

    const int* plan = b.plan + (size_t)img * plan_stride(a.mReco);
    const int G = plan[0], U = plan[1];
    const int* pGStart = plan + 2;
    const int* pOrd = pGStart + a.mReco + 1;
    const int* pUid = pOrd + a.mReco;
    const int* pGRep = pUid + a.mReco;
    const int* pTRep = pGRep + a.mReco;

    if (blockIdx.x == 0 && tid == 0 && a.O) {   // insertDir (src/Reconstructor.cpp:407-422) once per image
        double ox = 0, oy = 0, oz = 0;
        for (int m = 0; m < a.mReco; m++) {
            const double* R = a.rotMat + (dm0 + m) * 9;
            const double tx = a.trans[2 * (dm0 + m)] - offx, ty = a.trans[2 * (dm0 + m) + 1] - offy;
            ox += -(R[0] * tx + R[3] * ty);
            oy += -(R[1] * tx + R[4] * ty);
            oz += -(R[2] * tx + R[5] * ty);
        }
        unsafeAtomicAdd(&a.O[0], ox);
        unsafeAtomicAdd(&a.O[1], oy);
        unsafeAtomicAdd(&a.O[2], oz);
        if (a.counter) atomicAdd(a.counter, a.mReco);
    }

    // this thread's pixel: coordinates, value, CTF, and the phase ramps of the image's unique shifts
    int pi = 0, pj = 0;
    float2 dv = make_float2(0.f, 0.f);
    float cf = 0.f;
    if (listed) {
        pi = a.iCol[p]; pj = a.iRow[p];
        dv = a.datP[(size_t)img * a.nPxl + p];
        cf = a.ctfP[(size_t)img * a.nPxl + p];
    }
    const int icp = pi * a.opf, irp = pj * a.opf;   // _iColPad / _iRowPad, src/Optimiser.cpp:8031-8033
    float2 ramp[kRampU];
#pragma unroll
    for (int u = 0; u < kRampU; u++) {
        ramp[u] = make_float2(0.f, 0.f);
        if (u < U && (!SLOWU || U <= kRampU)) {
            // translate(transImgP, orignImgP, -(tran - offset)(0), -(tran - offset)(1), ...), src/Optimiser.cpp:7160-7169
            const size_t dm = dm0 + pTRep[u];
            const double tx = a.trans[2 * dm] - offx, ty = a.trans[2 * dm + 1] - offy;
            ramp[u] = ramp_value((float)(-tx) / a.idim, (float)(-ty) / a.idim, pi, pj);
        }
    }
    const float wgt = a.w[img];
    const float gF = ldexpf(1.0f, b.gexp[0]), gT = ldexpf(1.0f, b.gexp[1]);
    const unsigned regionBase = (b.groupsBefore[img] * gridDim.x + blockIdx.x * (unsigned)G) * (unsigned)kBinThreads;
    int lN = 0;   // descriptors waiting in the LDS list (uniform)

    for (int g0 = 0; g0 < G; g0 += kPassGroups) {
        const int ng = G - g0 < kPassGroups ? G - g0 : kPassGroups;
        __syncthreads();   // the previous pass is done with the tables
        for (int i = tid; i < kHash; i += kBinThreads) { hKey[i] = -1; hCnt[i] = 0; }
        if (tid < ng) {
            const int rep = pGRep[g0 + tid];
            const double* R = a.rotMat + (dm0 + rep) * 9;
#pragma unroll
            for (int e = 0; e < 6; e++) sR[tid][e] = R[e];
            sCls[tid] = a.cls ? a.cls[dm0 + rep] : 0;
            sRep[tid] = rep;
        }
        if (tid <= ng) sM0[tid] = pGStart[g0 + tid];
        if (tid < kPassGroups * kRampU) (&sCntI[0][0])[tid] = 0;
        __syncthreads();
        if (!SLOWU || U <= kRampU)   // how many members of each group carry each unique shift
            for (int m = sM0[0] + tid; m < sM0[ng]; m += kBinThreads) {
                int gl = 0;
                while (gl + 1 < ng && m >= sM0[gl + 1]) gl++;
                atomicAdd(&sCntI[gl][pUid[pOrd[m]]], 1);
            }

        // ---- count: samples of this pass per brick ----
#if THX_BIN_KEEP
        // a thread's samples stay in registers for the scatter below (geometry and hash slot once per sample; the loops are
        // unrolled over the pass's 8 groups so that the arrays are registers)
#if THX_BIN_KEEP == 1
        float gXd[kPassGroups], gYd[kPassGroups], gZd[kPassGroups];
#endif
        unsigned gInfo[kPassGroups];   // cell in brick (11 bits) | conj << 11 | hash slot << 12 | valid << 31; 0: no sample
#pragma unroll
        for (int gl = 0; gl < kPassGroups; gl++) {
            gInfo[gl] = 0u;
#if THX_BIN_KEEP == 1
            gXd[gl] = gYd[gl] = gZd[gl] = 0.f;
#endif
            if (gl < ng) {
                SampleGeom s;
                if (listed && sample_geom(sR[gl], icp, irp, P, sCls[gl], b.nBx, b.nBy, b.nBz, s)) {
                    const int h = hash_insert(hKey, s.key);
                    atomicAdd(&hCnt[h], 1);
#if THX_BIN_KEEP == 1
                    gXd[gl] = s.xd; gYd[gl] = s.yd; gZd[gl] = s.zd;
#endif
                    gInfo[gl] = s.cell | (s.conj ? 0x800u : 0u) | ((unsigned)h << 12) | 0x80000000u;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        for (int gl = 0; gl < ng; gl++) {
            SampleGeom s;
            // (LDS atomics of one wave on a few addresses: the hardware's own serialisation is cheaper than sorting the lanes by key)
            if (listed && sample_geom(sR[gl], icp, irp, P, sCls[gl], b.nBx, b.nBy, b.nBz, s)) atomicAdd(&hCnt[hash_insert(hKey, s.key)], 1);
        }
#endif
        __syncthreads();

        // ---- the pass's segments: one per brick it touched, packed from the pass's static record base on ----
        constexpr int kPer = kHash / kBinThreads;
        int c[kPer], sum = 0, nz = 0;
#pragma unroll
        for (int i = 0; i < kPer; i++) { c[i] = hCnt[tid * kPer + i]; sum += c[i]; nz += c[i] > 0; }
        int isum = sum, inz = nz;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int ts = __shfl_up(isum, o, 64), tn = __shfl_up(inz, o, 64);
            if (lane >= o) { isum += ts; inz += tn; }
        }
        if (lane == 63) { sWS[wave] = isum; sWN[wave] = inz; }
        __syncthreads();
        int totalSeg = 0, preS = 0, preN = 0;
#pragma unroll
        for (int wv = 0; wv < kBinThreads / 64; wv++) {
            if (wv < wave) { preS += sWS[wv]; preN += sWN[wv]; }
            totalSeg += sWN[wv];
        }
        if (totalSeg == 0) continue;   // (uniform)
        if (lN + totalSeg > kSegList) {   // (uniform; totalSeg <= kSegList by the static_assert below)
            bin_flush_list(b, lKey, lVal, regionBase, lN, &sFlushBase);
            lN = 0;
        }
        {
            unsigned rOff = (unsigned)g0 * (unsigned)kBinThreads + (unsigned)(preS + isum - sum);   // from the region's base
            int li = lN + preN + inz - nz;
#pragma unroll
            for (int i = 0; i < kPer; i++) {
                if (c[i] <= 0) continue;
                lKey[li] = (unsigned)hKey[tid * kPer + i];
                lVal[li] = rOff | ((unsigned)c[i] << 20);
                hCnt[tid * kPer + i] = (int)(regionBase + rOff);
                rOff += (unsigned)c[i];
                li++;
            }
        }
        lN += totalSeg;
        __syncthreads();

        // ---- scatter: the samples again, now with their values, each into its brick's segment ----
#if THX_BIN_KEEP
#pragma unroll
        for (int gl = 0; gl < kPassGroups; gl++) {
            if (!(gInfo[gl] & 0x80000000u)) continue;
            const unsigned idx = (unsigned)atomicAdd(&hCnt[(gInfo[gl] >> 12) & (kHash - 1)], 1);
            SampleGeom s;
#if THX_BIN_KEEP == 1
            s.xd = gXd[gl]; s.yd = gYd[gl]; s.zd = gZd[gl]; s.cell = gInfo[gl] & 0x7FFu; s.conj = (gInfo[gl] & 0x800u) != 0;
#else   // 2: only the hash slot is kept, the geometry is computed again
            (void)sample_geom(sR[gl], icp, irp, P, sCls[gl], b.nBx, b.nBy, b.nBz, s);
#endif
#else
        for (int gl = 0; gl < ng; gl++) {
            SampleGeom s;
            if (!(listed && sample_geom(sR[gl], icp, irp, P, sCls[gl], b.nBx, b.nBy, b.nBz, s))) continue;
            const unsigned idx = (unsigned)atomicAdd(&hCnt[hash_find(hKey, s.key)], 1);
#endif
            // the value of this pixel for the group: (img * sum of the members' ramps) * ctf * w, T: n * ctf^2 * w
            const int m0 = sM0[gl], m1 = sM0[gl + 1];
            float2 S = make_float2(0.f, 0.f);
            if (!SLOWU || U <= kRampU) {
#pragma unroll
                for (int u = 0; u < kRampU; u++)
                    if (u < U) {
                        const float n = (float)sCntI[gl][u];
                        S.x = fmaf(n, ramp[u].x, S.x);
                        S.y = fmaf(n, ramp[u].y, S.y);
                    }
            } else if (SLOWU) {
                S = ramp_sum_members(a.trans, pOrd, dm0, m0, m1, offx, offy, a.idim, pi, pj);
            }
            const float2 tv = cmul(dv, S);
            float cfv = cf;
            if (CS) cfv = insert_ctf_search(a.attr, a.dfac, img, a.mReco, sRep[gl], a.pixelSize, a.idim, pi, pj);
            // src[i] * ctf[i] * 1 * w, left to right (src/Reconstructor.cpp:830-833)
            float vre = tv.x * cfv, vim = tv.y * cfv;
            vre = vre * 1.0f; vim = vim * 1.0f;
            vre = vre * wgt; vim = vim * wgt;
            if (s.conj) vim = -vim;
            const float tval = (pow2f_(cfv) * 1.0f * wgt) * (float)(m1 - m0);
            b.recA[idx] = make_uint4(__float_as_uint(s.xd), __float_as_uint(s.yd), __float_as_uint(s.zd), s.cell);
            float* rb = b.recB + 3 * (size_t)idx;
            rb[0] = vre * gF; rb[1] = vim * gF; rb[2] = tval * gT;
#if THX_BIN_KEEP
            __builtin_amdgcn_sched_barrier(0);   // one group's record at a time: the unrolled bodies must not pile their loads up in registers
#endif
        }
    }
    bin_flush_list(b, lKey, lVal, regionBase, lN, &sFlushBase);
}

// out[l] = groups of image l; out[nImg] = the largest number of unique shifts of any image (zeroed by the caller)
__global__ void k_plan_counts(int* __restrict__ out, const int* __restrict__ plan, int nImg, int stride)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nImg) return;
    out[l] = plan[(size_t)l * stride];
    atomicMax(&out[nImg], plan[(size_t)l * stride + 1]);
}

__global__ void k_seg_unpack(unsigned* __restrict__ segOff, unsigned* __restrict__ segCnt, unsigned* __restrict__ cum0,
                             const unsigned long long* __restrict__ val, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) cum0[0] = 0u;
    if (i >= n) return;
    const unsigned long long v = val[i];
    segOff[i] = (unsigned)v;
    segCnt[i] = (unsigned)(v >> 32);
}

struct AccArgs {
    const uint4* recA;
    const float* recB;
    const unsigned* segKey;   // sorted
    const unsigned* segOff;
    const unsigned* segCnt;
    const unsigned* cum;      // [nSeg + 1] records before segment i
    int nSeg;
    long long* accF;
    long long* accT;
    int P, nBx, nBy, nBz;
};

// first index in [0, n] whose cum is >= v
__device__ __forceinline__ int lower_bound_u32(const unsigned* __restrict__ cum, int n, unsigned long long v)
{
    int lo = 0, hi = n + 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((unsigned long long)cum[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// workgroup w owns the segments that START in records [w kAccSpan, (w + 1) kAccSpan) of the sorted order
// (HIP's second launch-bound argument is the minimum number of WAVES PER SIMD, not workgroups per CU: THX_ACC_WGS workgroups of
// kAccThreads threads on a CU's four SIMDs)
__global__ __launch_bounds__(kAccThreads, THX_ACC_WGS * kAccThreads / 64 / 4) void k_acc(AccArgs q)
{
    __shared__ long long sRe[kBrickVox], sIm[kBrickVox], sT[kBrickVox];
    __shared__ unsigned sOff[kAccStage], sExc[kAccStage + 1], sWaveTot[kAccThreads / 64];
    __shared__ int sRunEnd, sNext;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long lo = (unsigned long long)blockIdx.x * kAccSpan, hi = lo + kAccSpan;
    if (lo >= (unsigned long long)q.cum[q.nSeg]) return;
    int s0 = lower_bound_u32(q.cum, q.nSeg, lo), s1 = lower_bound_u32(q.cum, q.nSeg, hi);
    s1 = s1 > q.nSeg ? q.nSeg : s1;
    for (int e = tid; e < kBrickVox; e += kAccThreads) { sRe[e] = 0; sIm[e] = 0; sT[e] = 0; }
    const int P = q.P;
    const long nc = P / 2 + 1;
    const size_t volSize = (size_t)P * P * (P / 2 + 1);
    for (int s = s0; s < s1;) {
        const unsigned key = q.segKey[s];
        lds_barrier();   // (LDS only: the previous brick's global atomics stay in flight)
        if (tid == 0) sRunEnd = s1;
        lds_barrier();
        for (int i = s + 1 + tid; i < s1; i += kAccThreads)
            if (q.segKey[i] != key) { atomicMin(&sRunEnd, i); break; }
        lds_barrier();
        const int e = sRunEnd;
        // ---- accumulate the run's segments, a wave at a time; the records of the NEXT batch of 64 are in flight while the
        // current one is added (a wave's loads would otherwise be exposed once per batch: 16 waves per CU do not cover them) ----
        for (int blk = s; blk < e; blk += kAccStage) {   // the run's descriptors, kAccStage at a time, through LDS
        const int nb = e - blk < kAccStage ? e - blk : kAccStage;
        if (blk > s) lds_barrier();   // the previous block's descriptors are no longer read
        // the block's records form ONE logical stream (sExc = records before each segment): a wave takes 64 consecutive
        // records of the stream whatever segments they lie in -- taken segment by segment, a third of the lanes of the LDS
        // atomics would be idle (a segment holds ~130 records: two full batches and a remainder)
        {
            const unsigned c = tid < nb ? q.segCnt[blk + tid] : 0u;
            unsigned inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = (unsigned)__shfl_up((int)inc, o, 64);
                if (lane >= o) inc += t;
            }
            if (lane == 63) sWaveTot[tid >> 6] = inc;
            if (tid < nb) sOff[tid] = q.segOff[blk + tid];
            lds_barrier();
            unsigned pre = 0, tot = 0;
#pragma unroll
            for (int wv = 0; wv < kAccThreads / 64; wv++) {
                if (wv < (tid >> 6)) pre += sWaveTot[wv];
                tot += sWaveTot[wv];
            }
            if (tid < nb) sExc[tid] = pre + inc - c;
            if (tid == 0) { sExc[nb] = tot; sNext = 0; }
        }
        lds_barrier();
        const unsigned nRecBlk = sExc[nb];
        auto next_batch = [&](size_t& rec, bool& ok) -> bool {
            int j = 0;
            if (lane == 0) j = atomicAdd(&sNext, 1);
            const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane(j) * 64u;
            if (r0 >= nRecBlk) return false;
            int lo_ = 0, hi_ = nb - 1;   // (uniform) last segment that starts at or before r0
            while (lo_ < hi_) {
                const int mid = (lo_ + hi_ + 1) >> 1;
                if ((unsigned)__builtin_amdgcn_readfirstlane((int)sExc[mid]) <= r0) lo_ = mid; else hi_ = mid - 1;
            }
            const unsigned r = r0 + (unsigned)lane;
            ok = r < nRecBlk;
            int i = lo_;
            while (ok && sExc[i + 1] <= r) i++;   // (a batch spans a segment or two)
            rec = ok ? (size_t)(sOff[i] + (r - sExc[i])) : 0;
            return true;
        };
        // (the loads are unconditional -- lanes without a record read record 0 -- so that the loaded registers have ONE
        // definition and the wait for them lands at the top of the next trip, after the current batch has been added)
        size_t nxt = 0;
        bool okN = false;
        bool more = next_batch(nxt, okN);
        uint4 raN = q.recA[nxt];
        float v0N = q.recB[3 * nxt], v1N = q.recB[3 * nxt + 1], v2N = q.recB[3 * nxt + 2];
        while (more) {
            const uint4 ra = raN;
            const float vreS = v0N, vimS = v1N, tvalS = v2N;
            const bool act = okN;
            nxt = 0; okN = false;
            more = next_batch(nxt, okN);
            raN = q.recA[nxt];
            v0N = q.recB[3 * nxt]; v1N = q.recB[3 * nxt + 1]; v2N = q.recB[3 * nxt + 2];
            if (act) {
                const float xd = __uint_as_float(ra.x), yd = __uint_as_float(ra.y), zd = __uint_as_float(ra.z);
                const float vx[2] = {1.0f - xd, xd}, vy[2] = {1.0f - yd, yd}, vz[2] = {1.0f - zd, zd};
                const int cx = ra.w & (kBx - 1), cy = (ra.w >> 4) & (kBy - 1), cz = (ra.w >> 8) & (kBz - 1);
                const int base = (cz * kVy + cy) * kVx + cx;
#pragma unroll
                for (int v = 0; v < 8; v++) {
                    const int ii = v & 1, jj = (v >> 1) & 1, kk = v >> 2;
                    const float wv = vx[ii] * vy[jj] * vz[kk];
                    const long long t = f2ll_magic(tvalS * wv);
                    if (t == 0) continue;   // F and T travel together
                    const int idx = base + (kk * kVy + jj) * kVx + ii;
                    atomicAdd(reinterpret_cast<unsigned long long*>(&sRe[idx]), (unsigned long long)f2ll_magic(vreS * wv));
                    atomicAdd(reinterpret_cast<unsigned long long*>(&sIm[idx]), (unsigned long long)f2ll_magic(vimS * wv));
                    atomicAdd(reinterpret_cast<unsigned long long*>(&sT[idx]), (unsigned long long)t);
                }
            }
        }
        }
        lds_barrier();
        // ---- flush the brick: x fastest, i.e. along the volume's contiguous axis ----
        {
            const int bx = (int)(key % (unsigned)q.nBx), by = (int)((key / (unsigned)q.nBx) % (unsigned)q.nBy);
            const int bzk = (int)(key / ((unsigned)q.nBx * (unsigned)q.nBy));
            const int bz = bzk % q.nBz, cls = bzk / q.nBz;
            long long* F = q.accF + (size_t)cls * volSize * 2;
            long long* T = q.accT + (size_t)cls * volSize;
            for (int v = tid; v < kBrickVox; v += kAccThreads) {
                const long long re = sRe[v], im = sIm[v], tt = sT[v];
                if ((re | im | tt) == 0) continue;
                sRe[v] = 0; sIm[v] = 0; sT[v] = 0;
                const int vxI = v % kVx, r = v / kVx, vyI = r % kVy, vzI = r / kVy;
                const int X = bx * kBx + vxI, Y = by * kBy + vyI - P / 2, Z = bz * kBz + vzI - P / 2;
                const long gi = ((long)(Z >= 0 ? Z : Z + P) * P + (Y >= 0 ? Y : Y + P)) * nc + X;
                acc_add(F, T, gi, re, im, tt);
            }
        }
        lds_barrier();   // the flush's global atomics stay in flight
        s = e;
    }
}

// scratch the sorted form keeps per device: THX_INSERT_SCRATCH_MB, else chosen on first use as min(32 GiB, 40 % of the free
// memory); never less than one image's worst case.  (Every chunk flushes each brick it touched once -- 8 MB of 64-bit atomics per
// image at 250-image chunks -- so larger chunks are cheaper: insertion stage of 20 000 particles 374 / 335 / 309 / 302 / 297 ms at
// 4 / 8 / 16 / 32 / 64 GiB, two record buffers each)
static size_t sort_budget_bytes(size_t oneImageWorst)
{
    static size_t chosen[64] = {0};
    static std::mutex mtx;
    size_t b = knobs().insertScratchMB > 0 ? (size_t)knobs().insertScratchMB << 20 : 0;
    if (!b) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        std::lock_guard<std::mutex> lock(mtx);
        if (!chosen[dev]) {
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) freeB = (size_t)4 << 30;
            chosen[dev] = std::max(std::min((size_t)32 << 30, (size_t)(0.4 * (double)freeB)), (size_t)64 << 20);
        }
        b = chosen[dev];
    }
    return std::max(b, oneImageWorst);
}

static inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// two events for the pipelined chunk loop, destroyed on every way out
struct EventPair {
    hipEvent_t e[2] = {nullptr, nullptr};
    ~EventPair() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
};

int insert_sorted(hipStream_t st, const InsertArgs& a, const int* plan, const int* gexp, long long* accF, long long* accT, int nImg)
{
    const int nPxl = a.nPxl, mReco = a.mReco, P = a.P, nK = a.nK > 0 ? a.nK : 1;
    const int nBx = (P / 2 + kBx - 1) / kBx + 1, nBy = (P + kBy - 1) / kBy, nBz = (P + kBz - 1) / kBz;
    const unsigned long long nBrick = (unsigned long long)nBx * nBy * nBz * nK;
    THX_REQUIRE(nBrick < 0x7FFFFFFFull, "volume has too many bricks for 31-bit brick ids");
    THX_REQUIRE(mReco < 4096, "brick-sorted insertion packs a region's record offsets into 20 bits (mReco < 4096)");
    // the images' group counts decide how many fit a chunk: the call's one blocking read-back (page-locked)
    int* gDev = reinterpret_cast<int*>(scratch(st, 13, ((size_t)nImg + 1) * sizeof(int)));
    THX_REQUIRE(gDev, "device scratch allocation failed");
    // host words of the call: [nImg + 1] group counts, [nImg] groups-before, [2] descriptor counts of the two chunks in flight
    int* hostWords = reinterpret_cast<int*>(pinned_host(st, 0, ((size_t)2 * nImg + 4) * sizeof(int)));
    THX_REQUIRE(hostWords, "page-locked host allocation failed");
    int* gHost = hostWords;
    unsigned* pre = reinterpret_cast<unsigned*>(hostWords + nImg + 1);
    unsigned* usedHost = reinterpret_cast<unsigned*>(hostWords + 2 * nImg + 1);
    THX_CHECK(hipMemsetAsync(gDev + nImg, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_plan_counts, dim3((nImg + 255) / 256), dim3(256), 0, st, gDev, plan, nImg, plan_stride(mReco));
    THX_CHECK(hipMemcpyAsync(gHost, gDev, ((size_t)nImg + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipStreamSynchronize(st));
    const bool slowU = gHost[nImg] > kRampU;

    constexpr size_t kRecBytes = sizeof(uint4) + 3 * sizeof(float);
    constexpr size_t kSegBytes = 2 * sizeof(unsigned) + 2 * sizeof(unsigned long long) + 3 * sizeof(unsigned);
    constexpr int kSegShare = 8;   // descriptor slots per 8 records; a pass that finds the table full inserts its segments itself
    const int nRegion = (nPxl + kBinThreads - 1) / kBinThreads;
    const size_t recPerGroup = (size_t)nRegion * kBinThreads;   // static record span of one group of one image
    const size_t perRec = kRecBytes + kSegBytes / kSegShare + 1;
    const size_t oneImageWorst = (size_t)mReco * recPerGroup * perRec + ((size_t)1 << 20);
    size_t budget = sort_budget_bytes(oneImageWorst);
    {
        // the budget was chosen at the first call; what the device can give NOW may be less (another job's data has arrived):
        // never more than what this stream already holds or 40 % of what a new allocation could get
        size_t freeB = 0, totalB = 0;
        const size_t held = scratch_size(st, 12);
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const size_t can = std::max(held, (size_t)(0.4 * (double)(freeB + held)));
            budget = std::max(std::min(budget, can), oneImageWorst);
        }
    }

    // chunks: as many images as a record buffer holds; every image knows the groups of the chunk's images before it.  All of
    // them in ONE buffer if they fit; otherwise the scratch is TWO record buffers and chunk c + 1 is binned while the host
    // waits for the descriptor count of chunk c (the GPU has no idle gap in the loop; measured before: ~100 us per chunk)
    std::vector<int> chunkEnd;
    auto make_chunks = [&](unsigned long long cap) {
        chunkEnd.clear();
        for (int l0 = 0; l0 < nImg;) {
            unsigned long long groups = 0;
            int l1 = l0;
            while (l1 < nImg && l1 - l0 < 65535) {
                const unsigned long long g = (unsigned long long)gHost[l1];
                if (l1 > l0 && (groups + g) * recPerGroup > cap) break;
                pre[l1] = (unsigned)groups;
                groups += g;
                l1++;
            }
            chunkEnd.push_back(l1);
            l0 = l1;
        }
    };
    // (the layout is tried with the chosen budget; if the device cannot give that much any more -- another job's data arrived since
    // the budget was chosen -- with half of it, down to one image's worst case)
    int keyBits = 1;
    while ((1ull << keyBits) <= nBrick) keyBits++;   // 2^keyBits - 1 > every brick id: holes (all ones) sort last
    size_t tmpSort = 0, tmpScan = 0;
    void* tmp = nullptr;
    int nSets = 1;
    unsigned capR = 0, capS = 0;
    size_t oRecA[2] = {0, 0}, oRecB[2] = {0, 0}, oKeyIn[2] = {0, 0}, oValIn[2] = {0, 0};
    size_t oKeyOut = 0, oValOut = 0, oOff = 0, oCnt = 0, oCum = 0, oCtr = 0, oPre = 0;
    char* buf = nullptr;
    for (;;) {
        size_t capR64 = std::min((budget - ((size_t)1 << 20)) / perRec, (size_t)0xFFFF0000u);
        make_chunks(capR64);
        // (two buffers only if each still holds any one image: a job that fills the GPU -- 20 000 x 512^2 -- keeps the one-buffer loop)
        nSets = (chunkEnd.size() > 1 && budget >= 2 * oneImageWorst) ? 2 : 1;
        if (nSets == 2) {
            capR64 /= 2;
            make_chunks(capR64);
        }
        capR = (unsigned)capR64;
        capS = knobs().insertSegCap > 0 ? (unsigned)knobs().insertSegCap : capR / kSegShare + 4096;
        size_t o = 0;
        for (int s = 0; s < nSets; s++) {
            oRecA[s] = o; o += align256((size_t)capR * sizeof(uint4));
            oRecB[s] = o; o += align256((size_t)capR * 3 * sizeof(float));
            oKeyIn[s] = o; o += align256((size_t)capS * sizeof(unsigned));
            oValIn[s] = o; o += align256((size_t)capS * sizeof(unsigned long long));
        }
        oKeyOut = o; o += align256((size_t)capS * sizeof(unsigned));
        oValOut = o; o += align256((size_t)capS * sizeof(unsigned long long));
        oOff = o; o += align256((size_t)capS * sizeof(unsigned));
        oCnt = o; o += align256((size_t)capS * sizeof(unsigned));
        oCum = o; o += align256(((size_t)capS + 1) * sizeof(unsigned));
        oCtr = o; o += 256;
        oPre = o; o += align256((size_t)nImg * sizeof(unsigned));
        buf = reinterpret_cast<char*>(scratch(st, 12, o));
        if (buf) {
            // the sort's workspace belongs to the same layout: if it does not fit next to the records, the whole layout shrinks
            tmpSort = 0; tmpScan = 0;
            THX_CHECK(rocprim::radix_sort_pairs(nullptr, tmpSort, reinterpret_cast<unsigned*>(buf + oKeyIn[0]), reinterpret_cast<unsigned*>(buf + oKeyOut),
                                                reinterpret_cast<unsigned long long*>(buf + oValIn[0]), reinterpret_cast<unsigned long long*>(buf + oValOut),
                                                (size_t)capS, 0u, (unsigned)keyBits, st));
            THX_CHECK(rocprim::inclusive_scan(nullptr, tmpScan, reinterpret_cast<unsigned*>(buf + oCnt), reinterpret_cast<unsigned*>(buf + oCum) + 1, (size_t)capS,
                                              rocprim::plus<unsigned>(), st));
            tmp = scratch(st, 14, std::max(tmpSort, tmpScan));
            if (tmp) break;
            buf = nullptr;
            scratch_release(st, 12);   // (grow-only otherwise: the smaller layout must not sit in the buffer that was too large)
        }
        if (budget <= oneImageWorst) break;
        (void)hipGetLastError();
        budget = std::max(budget / 2, oneImageWorst);
    }
    THX_REQUIRE(buf, "device scratch allocation failed (brick-sorted insertion records)");
    unsigned* keyOut = reinterpret_cast<unsigned*>(buf + oKeyOut);
    unsigned long long* valOut = reinterpret_cast<unsigned long long*>(buf + oValOut);
    unsigned* segOff = reinterpret_cast<unsigned*>(buf + oOff);
    unsigned* segCnt = reinterpret_cast<unsigned*>(buf + oCnt);
    unsigned* cum = reinterpret_cast<unsigned*>(buf + oCum);
    unsigned* ctr = reinterpret_cast<unsigned*>(buf + oCtr);   // [2]
    unsigned* preDev = reinterpret_cast<unsigned*>(buf + oPre);

    THX_REQUIRE(tmp, "device scratch allocation failed (sort workspace)");
    const size_t tmpBytes = std::max(tmpSort, tmpScan);
    THX_CHECK(hipMemcpyAsync(preDev, pre, (size_t)nImg * sizeof(unsigned), hipMemcpyHostToDevice, st));   // (page-locked: no wait)

    EventPair ev;
    for (int s = 0; s < nSets; s++) THX_CHECK(hipEventCreateWithFlags(&ev.e[s], hipEventDisableTiming));
    const int nChunk = (int)chunkEnd.size();
    // k_bin of chunk ci into record buffer `set`, its descriptor count on the way to the host
    auto launch_bin = [&](int ci, int set) -> int {
        const int l0 = ci ? chunkEnd[ci - 1] : 0, nl = chunkEnd[ci] - l0;
        BinArgs b;
        b.a = a;
        b.a.datP += (size_t)l0 * nPxl; b.a.ctfP += (size_t)l0 * nPxl; b.a.w += l0;
        b.a.rotMat += (size_t)l0 * mReco * 9; b.a.trans += (size_t)l0 * mReco * 2;
        if (b.a.offS) b.a.offS += (size_t)l0 * 2;
        if (b.a.cls) b.a.cls += (size_t)l0 * mReco;
        if (b.a.attr) b.a.attr += l0;
        if (b.a.dfac) b.a.dfac += (size_t)l0 * mReco;
        b.plan = plan + (size_t)l0 * plan_stride(mReco);
        b.gexp = gexp; b.groupsBefore = preDev + l0; b.accF = accF; b.accT = accT;
        b.recA = reinterpret_cast<uint4*>(buf + oRecA[set]); b.recB = reinterpret_cast<float*>(buf + oRecB[set]);
        b.segKey = reinterpret_cast<unsigned*>(buf + oKeyIn[set]); b.segVal = reinterpret_cast<unsigned long long*>(buf + oValIn[set]);
        b.counter = ctr + set; b.capS = capS;
        b.nBx = nBx; b.nBy = nBy; b.nBz = nBz;
        THX_CHECK(hipMemsetAsync(ctr + set, 0, sizeof(unsigned), st));
        if (a.cSearch) {
            if (slowU) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bin<true, true>), dim3(nRegion, nl), dim3(kBinThreads), 0, st, b);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bin<true, false>), dim3(nRegion, nl), dim3(kBinThreads), 0, st, b);
        } else {
            if (slowU) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bin<false, true>), dim3(nRegion, nl), dim3(kBinThreads), 0, st, b);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bin<false, false>), dim3(nRegion, nl), dim3(kBinThreads), 0, st, b);
        }
        THX_CHECK(hipMemcpyAsync(usedHost + set, ctr + set, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        THX_CHECK(hipEventRecord(ev.e[set], st));
        return 0;
    };
    // descriptor sort and k_acc of chunk ci (the host needs the descriptor count: rocPRIM takes its sizes from the host)
    auto launch_acc = [&](int ci, int set) -> int {
        const int l1 = chunkEnd[ci];
        THX_CHECK(hipEventSynchronize(ev.e[set]));
        const int nSeg = (int)std::min(usedHost[set], capS);
        const unsigned long long nRecMax = ((unsigned long long)pre[l1 - 1] + (unsigned long long)gHost[l1 - 1]) * recPerGroup;
        if (nSeg <= 0) return 0;
        unsigned* keyIn = reinterpret_cast<unsigned*>(buf + oKeyIn[set]);
        unsigned long long* valIn = reinterpret_cast<unsigned long long*>(buf + oValIn[set]);
        size_t tb = tmpBytes;
        THX_CHECK(rocprim::radix_sort_pairs(tmp, tb, keyIn, keyOut, valIn, valOut, (size_t)nSeg, 0u, (unsigned)keyBits, st));
        hipLaunchKernelGGL(k_seg_unpack, dim3((nSeg + 255) / 256), dim3(256), 0, st, segOff, segCnt, cum, valOut, nSeg);
        tb = tmpBytes;
        THX_CHECK(rocprim::inclusive_scan(tmp, tb, segCnt, cum + 1, (size_t)nSeg, rocprim::plus<unsigned>(), st));
        AccArgs q;
        q.recA = reinterpret_cast<const uint4*>(buf + oRecA[set]); q.recB = reinterpret_cast<const float*>(buf + oRecB[set]);
        q.segKey = keyOut; q.segOff = segOff; q.segCnt = segCnt; q.cum = cum; q.nSeg = nSeg;
        q.accF = accF; q.accT = accT; q.P = P; q.nBx = nBx; q.nBy = nBy; q.nBz = nBz;
        const unsigned nWg = (unsigned)((nRecMax + kAccSpan - 1) / kAccSpan);   // (those beyond the records that exist return at once)
        hipLaunchKernelGGL(k_acc, dim3(nWg), dim3(kAccThreads), 0, st, q);
        return 0;
    };
    // stream order: bin(0) bin(1) | sort+acc(0) bin(2) | sort+acc(1) bin(3) ...: buffer c & 1 is binned into again only after
    // k_acc of chunk c - 2 (same stream), and the host's wait for chunk c's count falls under k_bin of chunk c + 1
    if (launch_bin(0, 0)) return -1;
    for (int ci = 0; ci < nChunk; ci++) {
        if (nSets == 2 && ci + 1 < nChunk && launch_bin(ci + 1, (ci + 1) & 1)) return -1;
        if (launch_acc(ci, nSets == 2 ? (ci & 1) : 0)) return -1;
        if (nSets == 1 && ci + 1 < nChunk && launch_bin(ci + 1, 0)) return -1;
    }
    THX_LAUNCH_CHECK();
    return 0;
}

}  // namespace thx
