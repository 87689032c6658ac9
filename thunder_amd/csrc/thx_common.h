// thx_common.h -- shared host/device helpers of libthunder_amd (gfx950 only).
//
// The whole library is compiled with -ffp-contract=off: every product and sum below is a separately
// rounded IEEE operation unless fmaf()/fma() is written out.  That is what makes the trilinear
// gather/scatter arithmetic bit-identical to the reference's x86 build (-O2 -mavx, no FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/thunder_amd.h"

namespace thx {

void set_error(const char* fmt, ...);

#define THX_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            thx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (int)_e;                                                                      \
        }                                                                                        \
    } while (0)

#define THX_REQUIRE(cond, msg)                                      \
    do {                                                            \
        if (!(cond)) {                                              \
            thx::set_error("%s (%s:%d)", msg, __FILE__, __LINE__);  \
            return -1;                                              \
        }                                                           \
    } while (0)

#define THX_LAUNCH_CHECK() THX_CHECK(hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Grow-only device scratch, one buffer per (stream, slot): valid until the next request for the same (stream, slot).
// Work on one stream is ordered, so reuse across calls needs no synchronisation.  (hipMallocAsync/hipFreeAsync were
// dropped: after a hipDeviceSynchronize the ROCm 7 pool hands released blocks to plain hipMalloc while still reusing
// them, which corrupted caller buffers in a torch-free process -- tests/cpp/mirror_roundtrip.cpp exercises that.)
void* scratch(hipStream_t stream, int slot, size_t bytes);
// page-locked HOST memory with the same lifetime rules (for the few words a call reads back from its stream)
void* pinned_host(hipStream_t stream, int slot, size_t bytes);
// bytes the (stream, slot) scratch buffer currently holds (0: none)
size_t scratch_size(hipStream_t stream, int slot);
// gives the (stream, slot) buffer back to the device (after the stream's queued work)
void scratch_release(hipStream_t stream, int slot);

// Environment switches for A/B runs.  Read ONCE, when the library is first used -- never per launch -- and only
// switches that select between tested code paths.
struct Knobs {
    int expectNSplit;     // THX_EXPECT_NSPLIT = 1..16: pixel splits of the local-search kernel (0 = automatic)
    int expectWgPerCU;    // THX_EXPECT_WG_PER_CU: overrides the occupancy argument of thx_expect_local_dev (-1 = unset)
    bool expectNdSweep;   // THX_EXPECT_ND=sweep: one launch per defocus factor instead of the fused kernel
    float expectSplit;    // THX_EXPECT_SPLIT=m: the near-slab / tail form of the local-search kernel (samples within m voxels of the cloud's
                          // mean slab are fetched one pixel ahead); 0 = the one-at-a-time form (default)
    int expectOrder;      // THX_EXPECT_ORDER = 0 / 1 / 2 / 3: lane <-> rotation of the local-search kernel in storage order / following a ranking of the
                          // image's cloud by the in-plane angle (default: -1.5 % per launch, round 5) / by one of the two tilt components, relative to
                          // its first rotation; bit-identical results
    int expectWgLater;    // THX_EXPECT_WG_LATER: occupancy argument of the local-search kernel for phase indices >= 1 (-1 = as phase 0)
    bool scanSimple;      // THX_SCAN=simple: the rotation-per-thread global-scan kernel for every size (A/B)
    int scanTile;         // THX_SCAN=t42 / t24 / t44: wave tiles of the scan contraction (A/B; default 2 x 2)
    bool insertPlain;     // THX_INSERT_PLAIN=1: plain float-atomic insertion (k_insert)
    long insertScratchMB; // THX_INSERT_SCRATCH_MB: record / descriptor scratch of the brick-sorted insertion (0 = min(32 GiB, 40 % of free))
    long insertSegCap;    // THX_INSERT_SEG_CAP: descriptor table size, to exercise the table-full path in tests (0 = records / 8)
    bool fftRocfft;       // THX_FFT=rocfft: library transforms in the gridding loop for every size
    bool recoTrace;       // THX_RECO_TRACE: print diffC per balancing round
    int fftzWaves;        // THX_FFTZ_WAVES = 4 / 8: register budget of the fused z pass of the gridding loop; 16: its sixteen-points-per-thread form (P = 1024 only, the default there); 0 = per size
    bool recoNatural;     // THX_RECO_WT=natural: W / T of the hand-written gridding loop in the volume's own layout (A/B; default: tiled by z column)
    bool recoHostStop;    // THX_RECO_STOP=host: the gridding loop's stop rule on the host, one 4-byte read-back per round (A/B; default: on the device)
    bool recoReplicate;   // THX_RECO_OWNERS=0: every rank of a half reconstructs every class after an all-reduce, as the reference's ranks do
                          // (A/B; default: class k is reduced to, and reconstructed by, rank k mod (ranks of the half))
    bool commForce;       // THX_COMM_FORCE=1: issue the RCCL calls on one-rank communicators too (1-GPU test of the path)
};
const Knobs& knobs();

constexpr double kM2xPi = 6.28318530717959;  // M_2X_PI, include/Macro.h:14

// ---------------------------------------------------------------------------------------------
// Trilinear cell on the half-Hermitian volume: conjHalf (include/Image/Volume.h:135-147),
// WG_TRI_INTERP_LINEAR (include/Functions/Interpolation.h:152-200), index wrap of iFTHalf
// (include/Image/Volume.h:567-575).  The 8 neighbours are visited k-outer, j, i-inner, as
// getFTHalf(w, x0) / addFTHalf(value, w, x0) do (src/Image/Volume.cpp:491-712); the box fast path and
// the wrapped slow path of the reference are the same arithmetic.
// ---------------------------------------------------------------------------------------------
struct TriCell {
    long rowOff[2][2];  // element offset of (k, j) row start + i0
    float w[8];         // w[k*4 + j*2 + i]
    bool conj;
};

__device__ __forceinline__ void tri_cell(TriCell& c, float x, float y, float z, int P)
{
    c.conj = false;
    if (!(x >= 0.0f)) {
        x *= -1.0f;
        y *= -1.0f;
        z *= -1.0f;
        c.conj = true;
    }
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float xd = x - fx, yd = y - fy, zd = z - fz;
    const float vx[2] = {1.0f - xd, xd}, vy[2] = {1.0f - yd, yd}, vz[2] = {1.0f - zd, zd};
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) c.w[k * 4 + j * 2 + i] = vx[i] * vy[j] * vz[k];
    const long nc = P / 2 + 1;
    const int ja[2] = {y0 >= 0 ? y0 : y0 + P, (y0 + 1) >= 0 ? y0 + 1 : y0 + 1 + P};
    const int ka[2] = {z0 >= 0 ? z0 : z0 + P, (z0 + 1) >= 0 ? z0 + 1 : z0 + 1 + P};
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 2; j++) c.rowOff[k][j] = ((long)ka[k] * P + ja[j]) * nc + x0;
}

// coordinates are valid for the stored half grid when every neighbour index is in range
__device__ __forceinline__ bool coord_in_grid(float x, float y, float z, int P)
{
    const float h = (float)(P / 2);
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    return (ax < h) && (ay < h - 1.0f) && (az < h - 1.0f);
}

// Volume::getByInterpolationFT, src/Image/Volume.cpp:314-338 (complex volume)
__device__ __forceinline__ float2 interp_ft(const float2* __restrict__ vol, int P, float x, float y, float z)
{
    TriCell c;
    tri_cell(c, x, y, z, P);
    float re = 0.0f, im = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float2* p = vol + c.rowOff[k][j];
            const float2 a = p[0], b = p[1];
            re = re + a.x * c.w[k * 4 + j * 2];
            im = im + a.y * c.w[k * 4 + j * 2];
            re = re + b.x * c.w[k * 4 + j * 2 + 1];
            im = im + b.y * c.w[k * 4 + j * 2 + 1];
        }
    return make_float2(re, c.conj ? -im : im);
}

// Cell-packed projector volume: for every cell origin (z, y, x) of the half grid the 8 corner values of its trilinear cell
// stored contiguously -- k outer, j, i inner, 8 x complex64 = 64 bytes, 64-byte aligned -- so that one sample's gather is
// ONE contiguous 64-byte read instead of four 16-byte reads from four different 128-byte lines.  8x the memory of the
// volume (4.3 GB at P = 512, of 288 GB).  Same values, same operation order as interp_ft: bit-identical results.
typedef float thx_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 interp_ft_packed(const float4* __restrict__ cells, int P, float x, float y, float z)
{
    bool conj = false;
    if (!(x >= 0.0f)) { x *= -1.0f; y *= -1.0f; z *= -1.0f; conj = true; }
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float xd = x - fx, yd = y - fy, zd = z - fz;
    const float vx[2] = {1.0f - xd, xd}, vy[2] = {1.0f - yd, yd}, vz[2] = {1.0f - zd, zd};
    const long nc = P / 2 + 1;
    const float4* c = cells + (((long)(z0 >= 0 ? z0 : z0 + P) * P + (y0 >= 0 ? y0 : y0 + P)) * nc + x0) * 4;
    // (re, im) as ONE two-lane value: v_pk_mul_f32 / v_pk_add_f32 on the register pairs the 16-byte loads deliver -- the same
    // products and the same sums in the same order as the scalar form (re and im never meet), half the instructions
    thx_v2f acc = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float4 ab = c[k * 2 + j];   // (i = 0: .x .y), (i = 1: .z .w)
            const float w0 = vx[0] * vy[j] * vz[k], w1 = vx[1] * vy[j] * vz[k];
            const thx_v2f a0 = {ab.x, ab.y}, a1 = {ab.z, ab.w}, ww0 = {w0, w0}, ww1 = {w1, w1};
            acc = acc + a0 * ww0;
            acc = acc + a1 * ww1;
        }
    return make_float2(acc.x, conj ? -acc.y : acc.y);
}

// interp_ft_packed in two halves for a software-pipelined caller: the cell of a sample (fold included), and the combination of its
// four 16-byte rows -- the same products and sums in the same order (bit-identical)
__device__ __forceinline__ const float4* packed_cell(const float4* __restrict__ cells, int P, float x, float y, float z)
{
    if (!(x >= 0.0f)) { x *= -1.0f; y *= -1.0f; z *= -1.0f; }
    const int x0 = (int)floorf(x), y0 = (int)floorf(y), z0 = (int)floorf(z);
    const long nc = P / 2 + 1;
    return cells + (((long)(z0 >= 0 ? z0 : z0 + P) * P + (y0 >= 0 ? y0 : y0 + P)) * nc + x0) * 4;
}
__device__ __forceinline__ float2 packed_combine(const float4 c[4], float x, float y, float z)
{
    bool conj = false;
    if (!(x >= 0.0f)) { x *= -1.0f; y *= -1.0f; z *= -1.0f; conj = true; }
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    const float xd = x - fx, yd = y - fy, zd = z - fz;
    const float vx[2] = {1.0f - xd, xd}, vy[2] = {1.0f - yd, yd}, vz[2] = {1.0f - zd, zd};
    thx_v2f acc = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float4 ab = c[k * 2 + j];
            const float w0 = vx[0] * vy[j] * vz[k], w1 = vx[1] * vy[j] * vz[k];
            const thx_v2f a0 = {ab.x, ab.y}, a1 = {ab.z, ab.w}, ww0 = {w0, w0}, ww1 = {w1, w1};
            acc = acc + a0 * ww0;
            acc = acc + a1 * ww1;
        }
    return make_float2(acc.x, conj ? -acc.y : acc.y);
}

// same for a real volume (T): conjugation is a no-op
__device__ __forceinline__ float interp_ft_real(const float* __restrict__ vol, int P, float x, float y, float z)
{
    TriCell c;
    tri_cell(c, x, y, z, P);
    float re = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float* p = vol + c.rowOff[k][j];
            re = re + p[0] * c.w[k * 4 + j * 2];
            re = re + p[1] * c.w[k * 4 + j * 2 + 1];
        }
    return re;
}

// TSGSL_pow_2/3/4 (src/Precision.cpp:263-276): RFLOAT -> double power -> RFLOAT
__device__ __host__ __forceinline__ float pow2f_(float x) { double d = x; return (float)(d * d); }
__device__ __host__ __forceinline__ float pow3f_(float x) { double d = x; return (float)(d * d * d); }
__device__ __host__ __forceinline__ float pow4f_(float x) { double d = x; double d2 = d * d; return (float)(d2 * d2); }

// gsl_hypot (external/packages/gsl-2.4/sys/hypot.c:24-55)
__device__ __host__ __forceinline__ double gsl_hypot_(double x, double y)
{
    double xabs = fabs(x), yabs = fabs(y), mn, mx;
    if (xabs < yabs) { mn = xabs; mx = yabs; } else { mn = yabs; mx = xabs; }
    if (mn == 0) return mx;
    double u = mn / mx;
    return mx * sqrt(1 + u * u);
}

// gsl_hypot3 (external/packages/gsl-2.4/sys/hypot.c:57-76)
__device__ __host__ __forceinline__ double gsl_hypot3_(double x, double y, double z)
{
    double xabs = fabs(x), yabs = fabs(y), zabs = fabs(z);
    double w = xabs > (yabs > zabs ? yabs : zabs) ? xabs : (yabs > zabs ? yabs : zabs);
    if (w == 0.0) return 0.0;
    return w * sqrt((xabs / w) * (xabs / w) + (yabs / w) * (yabs / w) + (zabs / w) * (zabs / w));
}

// TIK_RL(r) = j0(pi r)^2, src/Functions/Functions.cpp:236-239 with gsl_sf_bessel_j0
// (external/packages/gsl-2.4/specfunc/bessel_j.c:35-58)
__device__ __host__ __forceinline__ float tik_rl(float r)
{
    float x = (float)(3.14159265358979323846 * r);
    double xd = x, ax = fabs(xd), j;
    if (ax < 0.5) {
        const double y = xd * xd;
        const double c1 = -1.0 / 6.0, c2 = 1.0 / 120.0, c3 = -1.0 / 5040.0, c4 = 1.0 / 362880.0,
                     c5 = -1.0 / 39916800.0, c6 = 1.0 / 6227020800.0;
        j = 1.0 + y * (c1 + y * (c2 + y * (c3 + y * (c4 + y * (c5 + y * c6)))));
    } else {
        j = sin(xd) / xd;
    }
    return pow2f_((float)j);
}

// Per-image CTF constants of src/CTF.cpp:129-135
struct CtfConst {
    float w1, w2, K1, K2, dU, dV, theta, phaseShift;
};

__device__ __host__ __forceinline__ CtfConst ctf_const(const thx_ctf_attr& a, double dfac)
{
    CtfConst c;
    float lambda = (float)(12.2643247 / sqrt(a.voltage * (1 + a.voltage * 0.978466e-6)));
    c.w1 = sqrtf(1 - pow2f_(a.amplitudeContrast));
    c.w2 = a.amplitudeContrast;
    c.K1 = (float)(3.14159265358979323846 * lambda);
    c.K2 = (float)(1.57079632679489661923 * a.Cs * pow3f_(lambda));
    // defocusU * d : RFLOAT * double evaluated in double, narrowed at the CTF() call (src/Optimiser.cpp:7188-7189)
    c.dU = (float)(a.defocusU * dfac);
    c.dV = (float)(a.defocusV * dfac);
    c.theta = a.defocusTheta;
    c.phaseShift = a.phaseShift;
    return c;
}

// one CTF value, src/CTF.cpp:140-150
__device__ __forceinline__ float ctf_value(const CtfConst& c, float pixelSize, int nCol, int nRow, int iCol, int iRow)
{
    float u = (float)gsl_hypot_((double)(iCol / (pixelSize * nCol)), (double)(iRow / (pixelSize * nRow)));
    float angle = (float)(atan2((double)iRow, (double)iCol) - c.theta);
    float defocus = -(c.dU + c.dV + (c.dU - c.dV) * cosf(2 * angle)) / 2;
    float ki = c.K1 * defocus * pow2f_(u) + c.K2 * pow4f_(u) - c.phaseShift;
    return -c.w1 * sinf(ki) + c.w2 * cosf(ki);
}

// phase ramp of translate(), src/Image/ImageFunctions.cpp:243-251: COMPLEX_POLAR(-phase)
__device__ __forceinline__ float2 ramp_value(float rCol, float rRow, int iCol, int iRow)
{
    float phase = (float)(kM2xPi * (iCol * rCol + iRow * rRow));
    float s, c;
    sincosf(-phase, &s, &c);
    return make_float2(c, s);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// 64-lane wave reductions (DPP through __shfl_xor)
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// 64-lane sum with DPP row operations (6 full-rate VALU adds, no LDS crossbar): every lane of the result holds the total
// (read back from lane 63 through an SGPR).  Summation order differs from wave_sum's butterfly.
__device__ __forceinline__ float wave_sum_dpp(float v)
{
#define THX_DPP_ADD(ctrl, rmask)                                                                                      \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
    THX_DPP_ADD(0xB1, 0xF);   // quad_perm [1,0,3,2]
    THX_DPP_ADD(0x4E, 0xF);   // quad_perm [2,3,0,1]
    THX_DPP_ADD(0x141, 0xF);  // row_half_mirror
    THX_DPP_ADD(0x140, 0xF);  // row_mirror: every lane holds its 16-lane row sum
    THX_DPP_ADD(0x142, 0xA);  // row_bcast15 into rows 1 and 3
    THX_DPP_ADD(0x143, 0xC);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
#undef THX_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, i.e.
// s_waitcnt vmcnt(0): every wave would sit at the barrier until its outstanding GLOBAL operations -- here thousands of
// fire-and-forget atomics of a brick flush, microseconds each -- have completed.  Use where only LDS contents are handed
// over between the phases.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// caller-owned host buffer <-> device staging of the Interface.h-shaped *_host entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 4);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return (int)e;
        }
        return 0;
    }
    int upload(const void* h, size_t bytes)
    {
        int rc = alloc(bytes);
        if (rc) return rc;
        hipError_t e = hipMemcpy(p, h, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("hipMemcpy H2D failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        return 0;
    }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

#define THX_RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// one image per call (thx_estep.hip; what thx_ExpectLocalM_host runs): workspace bytes, and the fused gather + likelihood + weights
// of ONE image split over every 256-pixel chunk.  volOrCells: the padded FT, or (packed) its cell-packed copy.
size_t expect_local_single_workspace(int nPxl, int nR, int nT, int nD);
int expect_local_single(const float* volOrCells, bool packed, int vdim, int pf, int idim, const int* iCol, const int* iRow, int nPxl,
                        const float* datP, const float* ctfP, const float* sigRcpP, const double* rotMat, int nR, const double* trans,
                        int nT, int nD, double pC, const double* pR, const double* pT, const double* pD, float* wC, float* wR, float* wT,
                        float* wD, float* baseLine, void* workspace, hipStream_t st, unsigned* done = nullptr, unsigned doneVal = 0);
// (done / doneVal: a host-visible word the finalise kernel sets, after a system-scope fence, once the image's outputs are written)

// rotate3D(quaternion), src/Geometry/Euler.cpp:181-189: R = I + 2 q0 A + 2 A A, column-major out.  ONE statement of the arithmetic for
// the device kernel (k_rotmat) and for host callers that stage matrices themselves (thx_ExpectLocalRTD_host): IEEE double
// multiplications and additions in a fixed order (-ffp-contract=off on both sides), so the two give the same bits.
__host__ __device__ inline void rotate3d_colmajor(const double* q, double* mat)
{
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const double A[3][3] = {{0, -q3, q2}, {q3, 0, -q1}, {-q2, q1, 0}};
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[r][k] * A[k][c];
            mat[c * 3 + r] = (r == c ? 1.0 : 0.0) + 2 * q0 * A[r][c] + 2 * s;
        }
}

}  // namespace thx
