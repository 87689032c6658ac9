// thx_next.hip -- the callers either side of the E/M loop:
//   * re-mask of the particle images (Optimiser::reMaskImg, src/Optimiser.cpp:6093-6149; ReMask, Interface.h:517):
//     batched 2-D c2r -> 1/size scale x soft mask -> r2c with rocFFT (through hipFFT), IN PLACE on the image stack
//   * re-centring phase ramps on whole images / volumes (Optimiser::reCentreImg :6065-6090, TranslateI/TranslateI2D
//     Interface.h:504-515, translate() src/Image/ImageFunctions.cpp:269-284,322-339,363-384)
//   * the noise-model update (Optimiser::allReduceSigma, src/Optimiser.cpp:6395-6710): per-image shell spectra of the
//     residual against the top-pose projection, group accumulation, closing arithmetic.
// All of it is HBM-bound streaming or gather work; none of it is reshaped for MFMA.
#include <hipfft/hipfft.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "thx_common.h"

#include <rocprim/rocprim.hpp>

namespace thx {

#define THX_FFT_CHECK2(expr)                                                               \
    do {                                                                                   \
        hipfftResult _r = (expr);                                                          \
        if (_r != HIPFFT_SUCCESS) {                                                        \
            thx::set_error("%s failed: hipfftResult %d (%s:%d)", #expr, (int)_r, __FILE__, __LINE__); \
            return 1000 + (int)_r;                                                         \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------------------------------------------
// soft mask, softMask(Image& mask, r, ew), src/Functions/Mask.cpp:334-350.  Built on the host exactly as the
// reference's CPU path builds it (glibc cos in double), uploaded once per (device, idim, r, ew).
// ---------------------------------------------------------------------------------------------
static int cached_mask(const float** out, int idim, float r, float ew)
{
    static std::mutex mtx;
    static std::map<std::tuple<int, int, float, float>, float*> cache;
    int dev = 0;
    THX_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mtx);
    auto key = std::make_tuple(dev, idim, r, ew);
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<float> m((size_t)idim * idim);
        for (long j = -idim / 2; j < idim / 2; j++)
            for (long i = -idim / 2; i < idim / 2; i++) {
                float u = (float)gsl_hypot_((double)i, (double)j);
                size_t idx = (size_t)(j >= 0 ? j : j + idim) * idim + (size_t)(i >= 0 ? i : i + idim);
                if (u > r + ew) m[idx] = 0;
                else if (u >= r) m[idx] = (float)(0.5 + 0.5 * cos((u - r) / ew * 3.14159265358979323846));
                else m[idx] = 1;
            }
        float* d = nullptr;
        THX_CHECK(hipMalloc(&d, m.size() * sizeof(float)));
        THX_CHECK(hipMemcpy(d, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice));
        it = cache.emplace(key, d).first;
    }
    *out = it->second;
    return 0;
}

// batched in-place 2-D real plans on the padded layout: real rows of 2*(idim/2+1) floats overlay the complex rows
// (one plan per (device, stream, shape): a hipFFT plan carries its stream and work area)
static std::mutex g_plan2dMtx;
static std::map<std::tuple<int, hipStream_t, int, int, int>, hipfftHandle> g_plan2d;
void release_next_plans(int dev, hipStream_t st)
{
    std::lock_guard<std::mutex> g(g_plan2dMtx);
    for (auto it = g_plan2d.begin(); it != g_plan2d.end();)
        if (std::get<0>(it->first) == dev && std::get<1>(it->first) == st) { (void)hipfftDestroy(it->second); it = g_plan2d.erase(it); } else ++it;
}
static int cached_plan2d(hipfftHandle* out, int idim, int batch, hipfftType type, hipStream_t st)
{
    std::mutex& mtx = g_plan2dMtx;
    auto& cache = g_plan2d;
    int dev = 0;
    THX_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mtx);
    auto key = std::make_tuple(dev, st, idim, batch, (int)type);
    auto it = cache.find(key);
    if (it == cache.end()) {
        hipfftHandle p;
        int n[2] = {idim, idim};
        int nc = idim / 2 + 1;
        int rEmbed[2] = {idim, 2 * nc}, cEmbed[2] = {idim, nc};
        if (type == HIPFFT_C2R)
            THX_FFT_CHECK2(hipfftPlanMany(&p, 2, n, cEmbed, 1, idim * nc, rEmbed, 1, idim * 2 * nc, type, batch));
        else
            THX_FFT_CHECK2(hipfftPlanMany(&p, 2, n, rEmbed, 1, idim * 2 * nc, cEmbed, 1, idim * nc, type, batch));
        THX_FFT_CHECK2(hipfftSetStream(p, st));
        it = cache.emplace(key, p).first;
    }
    *out = it->second;
    return 0;
}

// FFT::bwExecutePlan's SCALE_RL(img, 1.0 / sizeRL) (float x double, src/FFT.cpp:353-354) then MUL_RL(img, mask)
// (src/Optimiser.cpp:6137-6138).  grid (idim, nImg), one row per block, float2 per thread-iteration.
__global__ __launch_bounds__(128) void k_scale_mask(float* __restrict__ rl, const float* __restrict__ mask, int idim,
                                                    int ld, double inv)
{
    const int row = blockIdx.x;
    float2* p = reinterpret_cast<float2*>(rl + ((size_t)blockIdx.y * idim + row) * ld);
    const float2* m = reinterpret_cast<const float2*>(mask + (size_t)row * idim);
    for (int c = threadIdx.x; c < idim / 2; c += blockDim.x) {
        float2 v = p[c];
        const float2 w = m[c];
        v.x = (float)(v.x * inv);
        v.y = (float)(v.y * inv);
        v.x *= w.x;
        v.y *= w.y;
        p[c] = v;
    }
}

// translate(Image& dst, const Image& src, [r,] tx, ty): grid (idim, nImg), row per block.
__global__ __launch_bounds__(128) void k_translate_image(float2* __restrict__ dst, const float2* __restrict__ src,
                                                         const double* __restrict__ t, int idim, float r2, int useR)
{
    const int row = blockIdx.x, l = blockIdx.y;
    const int nc = idim / 2 + 1;
    const long j = row < idim / 2 ? row : row - idim;
    const float rCol = (float)t[2 * l] / idim, rRow = (float)t[2 * l + 1] / idim;
    const size_t base = ((size_t)l * idim + row) * nc;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) {
        if (useR && !((double)i * (double)i + (double)j * (double)j < (double)r2)) continue;
        const float2 w = ramp_value(rCol, rRow, i, (int)j);
        dst[base + i] = cmul(src[base + i], w);
    }
}

// translate(Volume& dst, const Volume& src, r, tx, ty, tz): grid (dim, dim) = (row j, slice k).
__global__ __launch_bounds__(128) void k_translate_volume(float2* __restrict__ dst, const float2* __restrict__ src,
                                                          int dim, float r2, float rCol, float rRow, float rSlc)
{
    const int nc = dim / 2 + 1;
    const long j = (int)blockIdx.x < dim / 2 ? (long)blockIdx.x : (long)blockIdx.x - dim;
    const long k = (int)blockIdx.y < dim / 2 ? (long)blockIdx.y : (long)blockIdx.y - dim;
    const size_t base = ((size_t)blockIdx.y * dim + blockIdx.x) * nc;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) {
        if (!((double)i * i + (double)j * j + (double)k * k < (double)r2)) continue;
        const float phase = (float)(kM2xPi * (i * rCol + j * rRow + k * rSlc));
        float s, c;
        sincosf(-phase, &s, &c);
        dst[base + i] = cmul(src[base + i], make_float2(c, s));
    }
}

// ---------------------------------------------------------------------------------------------
// Sigma update.  Shell-sorted pixel table of the disc {i in [0,r], j in [-r,r), i^2+j^2 < r^2, round(|ij|) < r}
// (powerSpectrum's set, src/Functions/Spectrum.cpp:171-184), built on the host once per (device, rSig).
// ---------------------------------------------------------------------------------------------
struct ShellTable {
    short2* ij = nullptr;     // device, sorted by shell, scan order (j outer, i inner) inside a shell
    int* start = nullptr;     // device [rSig + 1]
    int n = 0;
};

static int cached_shells(const ShellTable** out, int rSig)
{
    static std::mutex mtx;
    static std::map<std::pair<int, int>, ShellTable> cache;
    int dev = 0;
    THX_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mtx);
    auto key = std::make_pair(dev, rSig);
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<std::vector<short2>> sh(rSig);
        for (long j = -rSig; j < rSig; j++)
            for (long i = 0; i <= rSig; i++) {
                if (!((double)i * i + (double)j * j < (double)pow2f_((float)rSig))) continue;
                int u = (int)rint(gsl_hypot_((double)i, (double)j));
                if (u < rSig) sh[u].push_back(make_short2((short)i, (short)j));
            }
        std::vector<short2> flat;
        std::vector<int> start(rSig + 1, 0);
        for (int u = 0; u < rSig; u++) {
            start[u] = (int)flat.size();
            flat.insert(flat.end(), sh[u].begin(), sh[u].end());
        }
        start[rSig] = (int)flat.size();
        ShellTable t;
        t.n = (int)flat.size();
        THX_CHECK(hipMalloc(&t.ij, flat.size() * sizeof(short2)));
        THX_CHECK(hipMalloc(&t.start, start.size() * sizeof(int)));
        THX_CHECK(hipMemcpy(t.ij, flat.data(), flat.size() * sizeof(short2), hipMemcpyHostToDevice));
        THX_CHECK(hipMemcpy(t.start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice));
        it = cache.emplace(key, t).first;
    }
    *out = &it->second;
    return 0;
}

// One block per image, one wave per shell (round-robin): per-shell sums of |ctf.P.ramp|^2, |img|^2,
// |img - ctf.P.ramp_M|^2, |imgOri - ctf.P.ramp_N|^2, each divided by the shell's pixel count
// (src/Optimiser.cpp:6443-6565 + powerSpectrum).  Lane-strided partial sums + a wave tree: deterministic, but a
// different summation order from the reference's serial scan (tolerance in tests/test_parity_gpu.py).
constexpr int kSigThreads = 256;
// PACKED: `volumes` are cell-packed copies (thx_projector_pack_dev): one 64-byte request per sample instead of four 16-byte ones
// (each of which costs the memory system a whole request); the same arithmetic, bit-identical spectra
template <bool PACKED>
__global__ __launch_bounds__(kSigThreads) void k_sigma_spectra(
    float* __restrict__ spec, const float2* __restrict__ volumes, const int* __restrict__ volIdx, int P, int pf,
    int idim, int projR, int rSig, const short2* __restrict__ ij, const int* __restrict__ shellStart,
    const float2* __restrict__ img, const float2* __restrict__ imgOri, const thx_ctf_attr* __restrict__ attr,
    const double* __restrict__ dfac, float pixelSize, const double* __restrict__ rotMat,
    const double* __restrict__ tran, const double* __restrict__ offset)
{
    const int l = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = idim / 2 + 1;
    const size_t imgSize = (size_t)idim * nc;
    const float2* vol = volumes + (size_t)(volIdx ? volIdx[l] : 0) * P * P * (P / 2 + 1) * (PACKED ? 8 : 1);
    const float2* im = img + (size_t)l * imgSize;
    const float2* io = imgOri + (size_t)l * imgSize;
    const double* m = rotMat + 9 * (size_t)l;
    const CtfConst cc = ctf_const(attr[l], dfac ? dfac[l] : 1.0);
    // RFLOAT nTransCol = t(0) narrowed at the translate() call, then / nColRL (src/Image/ImageFunctions.cpp:329-330)
    const double tM0 = tran[2 * l], tM1 = tran[2 * l + 1];
    const double tN0 = tM0 - (offset ? offset[2 * l] : 0.0), tN1 = tM1 - (offset ? offset[2 * l + 1] : 0.0);
    const float rColM = (float)tM0 / idim, rRowM = (float)tM1 / idim;
    const float rColN = (float)tN0 / idim, rRowN = (float)tN1 / idim;
    const int projR2 = projR * projR;

    for (int u = wave; u < rSig; u += kSigThreads / 64) {
        const int b = shellStart[u], e = shellStart[u + 1];
        float sS = 0.f, sD = 0.f, sM = 0.f, sN = 0.f;
        for (int p = b + lane; p < e; p += 64) {
            const short2 q = ij[p];
            const int i = q.x, j = q.y;
            const size_t idx = (size_t)(j >= 0 ? j : j + idim) * nc + i;
            const float2 a = im[idx], o = io[idx];
            float2 pM = make_float2(0.f, 0.f), pN = pM;
            if (i * i + j * j < projR2) {
                const double nx = (double)(i * pf), ny = (double)(j * pf);
                const double ox = m[0] * nx + m[3] * ny, oy = m[1] * nx + m[4] * ny, oz = m[2] * nx + m[5] * ny;
                const float2 s = PACKED ? interp_ft_packed(reinterpret_cast<const float4*>(vol), P, (float)ox, (float)oy, (float)oz)
                                        : interp_ft(vol, P, (float)ox, (float)oy, (float)oz);
                pM = cmul(s, ramp_value(rColM, rRowM, i, j));
                pN = cmul(s, ramp_value(rColN, rRowN, i, j));
            }
            const float c = ctf_value(cc, pixelSize, idim, idim, i, j);
            pM.x *= c; pM.y *= c; pN.x *= c; pN.y *= c;
            sS += pM.x * pM.x + pM.y * pM.y;
            sD += a.x * a.x + a.y * a.y;
            const float2 dM = make_float2(pM.x * -1 + a.x, pM.y * -1 + a.y);
            const float2 dN = make_float2(pN.x * -1 + o.x, pN.y * -1 + o.y);
            sM += dM.x * dM.x + dM.y * dM.y;
            sN += dN.x * dN.x + dN.y * dN.y;
        }
        sS = wave_sum(sS); sD = wave_sum(sD); sM = wave_sum(sM); sN = wave_sum(sN);
        if (lane == 0) {
            const unsigned cnt = (unsigned)(e - b);
            float* o4 = spec + (size_t)l * 4 * rSig;
            o4[u] = sS / cnt;
            o4[rSig + u] = sD / cnt;
            o4[2 * rSig + u] = sM / cnt;
            o4[3 * rSig + u] = sN / cnt;
        }
    }
}

// Optimiser::normCorrection, src/Optimiser.cpp:6201-6394, as include/Config.h configures it (OPTIMISER_NORM_MASK, _CTF_ON_THE_FLY,
// _RECENTRE_IMAGE_EACH_ITERATION; _ADJUST_2D_IMAGE_NOISE_ZERO_MEAN off): per image the power of what the top pose's slice does
// not explain, norm_l = sum over rL^2 <= i^2 + j^2 < rNorm^2 of |img - ctf . P . ramp(t)|^2 (the masked image _img; P inside
// Projector::_maxRadius).  One block per image over the half-image rows; lane-strided partial sums + a fixed tree (the
// reference adds the pixels serially in RFLOAT: tolerance in tests/test_next_gpu.py).
template <bool PACKED>
__global__ __launch_bounds__(kSigThreads) void k_norm_residual(
    float* __restrict__ norm, const float2* __restrict__ volumes, const int* __restrict__ volIdx, int P, int pf, int idim,
    int projR, float rL2, float rNorm2, const float2* __restrict__ img, const thx_ctf_attr* __restrict__ attr,
    const double* __restrict__ dfac, float pixelSize, const double* __restrict__ rotMat, const double* __restrict__ tran)
{
    __shared__ float red[kSigThreads / 64];
    const int l = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = idim / 2 + 1;
    const size_t imgSize = (size_t)idim * nc;
    const float2* vol = volumes + (size_t)(volIdx ? volIdx[l] : 0) * P * P * (P / 2 + 1) * (PACKED ? 8 : 1);
    const float2* im = img + (size_t)l * imgSize;
    const double* m = rotMat + 9 * (size_t)l;
    const CtfConst cc = ctf_const(attr[l], dfac ? dfac[l] : 1.0);
    const float rCol = (float)tran[2 * l] / idim, rRow = (float)tran[2 * l + 1] / idim;
    const int projR2 = projR * projR;
    float s = 0.f;
    for (size_t e = threadIdx.x; e < imgSize; e += kSigThreads) {
        const int row = (int)(e / nc), i = (int)(e - (size_t)row * nc);
        const int j = row < idim / 2 ? row : row - idim;
        const int q = i * i + j * j;
        if (!((float)q >= rL2 && (float)q < rNorm2)) continue;
        const float2 a = im[e];
        float2 p = make_float2(0.f, 0.f);
        if (q < projR2) {
            const double nx = (double)(i * pf), ny = (double)(j * pf);
            const double ox = m[0] * nx + m[3] * ny, oy = m[1] * nx + m[4] * ny, oz = m[2] * nx + m[5] * ny;
            p = cmul(PACKED ? interp_ft_packed(reinterpret_cast<const float4*>(vol), P, (float)ox, (float)oy, (float)oz)
                            : interp_ft(vol, P, (float)ox, (float)oy, (float)oz), ramp_value(rCol, rRow, i, j));
        }
        const float c = ctf_value(cc, pixelSize, idim, idim, i, j);
        p.x *= c; p.y *= c;
        const float2 d = make_float2(p.x * -1 + a.x, p.y * -1 + a.y);
        s += d.x * d.x + d.y * d.y;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) norm[l] = (red[0] + red[1]) + (red[2] + red[3]);
}

// gsl_stats_float_quantile_from_sorted_data(sorted, 1, n, 0.5) (external/packages/gsl-2.4/statistics/quantiles_source.c): RFLOAT m
__global__ void k_median_sorted(float* __restrict__ out, const float* __restrict__ sorted, int n)
{
    const double index = 0.5 * (n - 1);
    const int lhs = (int)index;
    const double delta = index - lhs;
    double r = 0;
    if (n > 0) r = lhs == n - 1 ? (double)sorted[lhs] : (1 - delta) * (double)sorted[lhs] + delta * (double)sorted[lhs + 1];
    *out = (float)r;
}

// _img[l][i] *= sqrt(m / norm(l)); _imgOri[l][i] *= sqrt(m / norm(l))   (src/Optimiser.cpp:6380-6392)
__global__ __launch_bounds__(256) void k_norm_scale(float2* __restrict__ img, float2* __restrict__ imgOri, const float* __restrict__ norm,
                                                    const float* __restrict__ median, size_t imgSize)
{
    const int l = blockIdx.y;
    const float f = sqrtf(*median / norm[l]);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < imgSize; e += (size_t)gridDim.x * blockDim.x) {
        float2 a = img[(size_t)l * imgSize + e], o = imgOri[(size_t)l * imgSize + e];
        a.x *= f; a.y *= f; o.x *= f; o.y *= f;
        img[(size_t)l * imgSize + e] = a;
        imgOri[(size_t)l * imgSize + e] = o;
    }
}

// Group accumulation, src/Optimiser.cpp:6567-6597 (w = 1): one block per group, 8 image-lanes x 128 shell-lanes,
// fixed combination order.  order = image indices sorted by group, gStart [nGroup+1].
__global__ __launch_bounds__(1024) void k_sigma_accum(float* __restrict__ sigM, float* __restrict__ sigN,
                                                      float* __restrict__ svd, const float* __restrict__ spec,
                                                      const int* __restrict__ order, const int* __restrict__ gStart,
                                                      int rSig)
{
    __shared__ float part[3][8][128];
    const int g = blockIdx.x, tx = threadIdx.x & 127, ty = threadIdx.x >> 7;
    const int b = gStart[g], e = gStart[g + 1];
    const int ncol = rSig + 1;
    for (int u0 = 0; u0 < rSig; u0 += 128) {
        const int u = u0 + tx;
        float aM = 0.f, aN = 0.f, aS = 0.f;
        if (u < rSig)
            for (int q = b + ty; q < e; q += 8) {
                const float* s = spec + (size_t)order[q] * 4 * rSig;
                aM += s[2 * rSig + u] / 2;
                aN += s[3 * rSig + u] / 2;
                aS += sqrtf(s[u] / s[rSig + u]);
            }
        part[0][ty][tx] = aM; part[1][ty][tx] = aN; part[2][ty][tx] = aS;
        __syncthreads();
        if (ty == 0 && u < rSig) {
            float tM = 0.f, tN = 0.f, tS = 0.f;
            for (int y = 0; y < 8; y++) { tM += part[0][y][tx]; tN += part[1][y][tx]; tS += part[2][y][tx]; }
            sigM[(size_t)g * ncol + u] += tM;
            sigN[(size_t)g * ncol + u] += tN;
            svd[(size_t)g * ncol + u] += tS;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float w = (float)(e - b);
        sigM[(size_t)g * ncol + rSig] += w;
        sigN[(size_t)g * ncol + rSig] += w;
        svd[(size_t)g * ncol + rSig] += w;
    }
}

// closing arithmetic, src/Optimiser.cpp:6654-6707
__global__ void k_sigma_final(float* __restrict__ sig, float* __restrict__ sigRcp, const float* __restrict__ sigM,
                              const float* __restrict__ sigN, const float* __restrict__ svd, int nGroup, int rSig,
                              int group, float alpha)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nGroup * rSig) return;
    const int g = idx / rSig, j = idx - g * rSig, src = group ? g : 0, ncol = rSig + 1;
    const float m = sigM[(size_t)src * ncol + j] / sigM[(size_t)src * ncol + rSig];
    const float n = sigN[(size_t)src * ncol + j] / sigN[(size_t)src * ncol + rSig];
    const float s = svd[(size_t)src * ncol + j] / svd[(size_t)src * ncol + rSig];
    const float ratio = (float)(1.0 < (double)s ? 1.0 : (double)s);
    const float v = ratio * m + (1 - ratio) * alpha * n;
    sig[idx] = v;
    sigRcp[idx] = (float)(-0.5 / v);
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_remask_dev(float* imgFT, int nImg, int idim, float maskRadiusPx, float ew, void* stream)
{
    if (nImg == 0) return 0;
    THX_REQUIRE(imgFT && nImg > 0 && idim > 0 && (idim % 2) == 0, "bad arguments");
    hipStream_t st = as_stream(stream);
    const float* mask = nullptr;
    int rc = cached_mask(&mask, idim, maskRadiusPx, ew);
    if (rc) return rc;
    const int nc = idim / 2 + 1;
    const size_t imgSize = (size_t)idim * nc * 2;  // floats
    const int kBatch = 1024;
    for (int b = 0; b < nImg; b += kBatch) {
        const int nb = nImg - b < kBatch ? nImg - b : kBatch;
        float* p = imgFT + (size_t)b * imgSize;
        hipfftHandle c2r, r2c;
        if ((rc = cached_plan2d(&c2r, idim, nb, HIPFFT_C2R, st))) return rc;
        if ((rc = cached_plan2d(&r2c, idim, nb, HIPFFT_R2C, st))) return rc;
        THX_FFT_CHECK2(hipfftExecC2R(c2r, reinterpret_cast<hipfftComplex*>(p), p));
        hipLaunchKernelGGL(k_scale_mask, dim3(idim, nb), dim3(128), 0, st, p, mask, idim, 2 * nc,
                           1.0 / ((double)idim * idim));
        THX_LAUNCH_CHECK();
        THX_FFT_CHECK2(hipfftExecR2C(r2c, p, reinterpret_cast<hipfftComplex*>(p)));
    }
    return 0;
}

int thx_translate_image_dev(float* dst, const float* src, const double* trans, int nImg, int idim, float r,
                            void* stream)
{
    THX_REQUIRE(dst && src && trans && idim > 0, "bad arguments");
    if (nImg <= 0) return 0;
    hipLaunchKernelGGL(k_translate_image, dim3(idim, nImg), dim3(128), 0, as_stream(stream),
                       reinterpret_cast<float2*>(dst), reinterpret_cast<const float2*>(src), trans, idim,
                       r >= 0 ? pow2f_(r) : 0.f, r >= 0 ? 1 : 0);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_translate_volume_dev(float* dst, const float* src, int dim, float r, double ox, double oy, double oz,
                             void* stream)
{
    THX_REQUIRE(dst && src && dim > 0 && r >= 0, "bad arguments");
    const float rCol = (float)ox / dim, rRow = (float)oy / dim, rSlc = (float)oz / dim;
    hipLaunchKernelGGL(k_translate_volume, dim3(dim, dim), dim3(128), 0, as_stream(stream),
                       reinterpret_cast<float2*>(dst), reinterpret_cast<const float2*>(src), dim, pow2f_(r), rCol, rRow,
                       rSlc);
    THX_LAUNCH_CHECK();
    return 0;
}

static int sigma_spectra_impl(float* spec, const float* volumes, const int* volIdx, int vdim, int pf, int idim, int projR,
                              int rSig, const float* img, const float* imgOri, const thx_ctf_attr* attr,
                              const double* dfac, float pixelSize, const double* rotMat, const double* trans,
                              const double* offset, int nImg, void* stream, bool packed)
{
    THX_REQUIRE(spec && volumes && img && imgOri && attr && rotMat && trans, "NULL pointer");
    THX_REQUIRE(rSig > 0 && rSig <= idim / 2 && projR >= 0 && projR * pf < vdim / 2 - 1, "radius out of range");
    if (nImg <= 0) return 0;
    const ShellTable* t = nullptr;
    int rc = cached_shells(&t, rSig);
    if (rc) return rc;
    if (packed)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sigma_spectra<true>), dim3(nImg), dim3(kSigThreads), 0, as_stream(stream), spec,
                           reinterpret_cast<const float2*>(volumes), volIdx, vdim, pf, idim, projR, rSig, t->ij, t->start,
                           reinterpret_cast<const float2*>(img), reinterpret_cast<const float2*>(imgOri), attr, dfac,
                           pixelSize, rotMat, trans, offset);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sigma_spectra<false>), dim3(nImg), dim3(kSigThreads), 0, as_stream(stream), spec,
                           reinterpret_cast<const float2*>(volumes), volIdx, vdim, pf, idim, projR, rSig, t->ij, t->start,
                           reinterpret_cast<const float2*>(img), reinterpret_cast<const float2*>(imgOri), attr, dfac,
                           pixelSize, rotMat, trans, offset);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_sigma_spectra_dev(float* spec, const float* volumes, const int* volIdx, int vdim, int pf, int idim, int projR,
                          int rSig, const float* img, const float* imgOri, const thx_ctf_attr* attr,
                          const double* dfac, float pixelSize, const double* rotMat, const double* trans,
                          const double* offset, int nImg, void* stream)
{
    return sigma_spectra_impl(spec, volumes, volIdx, vdim, pf, idim, projR, rSig, img, imgOri, attr, dfac, pixelSize, rotMat, trans, offset,
                              nImg, stream, false);
}

int thx_sigma_spectra_packed_dev(float* spec, const float* cells, const int* volIdx, int vdim, int pf, int idim, int projR,
                                 int rSig, const float* img, const float* imgOri, const thx_ctf_attr* attr,
                                 const double* dfac, float pixelSize, const double* rotMat, const double* trans,
                                 const double* offset, int nImg, void* stream)
{
    return sigma_spectra_impl(spec, cells, volIdx, vdim, pf, idim, projR, rSig, img, imgOri, attr, dfac, pixelSize, rotMat, trans, offset,
                              nImg, stream, true);
}

static int norm_residual_impl(float* norm, const float* volumes, const int* volIdx, int vdim, int pf, int idim, int projR, float rL,
                              float rNorm, const float* img, const thx_ctf_attr* attr, const double* dfac, float pixelSize,
                              const double* rotMat, const double* trans, int nImg, void* stream, bool packed)
{
    THX_REQUIRE(norm && volumes && img && attr && rotMat && trans, "NULL pointer");
    THX_REQUIRE(projR >= 0 && projR * pf < vdim / 2 - 1 && rL >= 0 && rNorm >= 0, "radius out of range");
    if (nImg <= 0) return 0;
    if (packed)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_norm_residual<true>), dim3(nImg), dim3(kSigThreads), 0, as_stream(stream), norm,
                           reinterpret_cast<const float2*>(volumes), volIdx, vdim, pf, idim, projR, pow2f_(rL), pow2f_(rNorm),
                           reinterpret_cast<const float2*>(img), attr, dfac, pixelSize, rotMat, trans);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_norm_residual<false>), dim3(nImg), dim3(kSigThreads), 0, as_stream(stream), norm,
                           reinterpret_cast<const float2*>(volumes), volIdx, vdim, pf, idim, projR, pow2f_(rL), pow2f_(rNorm),
                           reinterpret_cast<const float2*>(img), attr, dfac, pixelSize, rotMat, trans);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_norm_residual_dev(float* norm, const float* volumes, const int* volIdx, int vdim, int pf, int idim, int projR, float rL,
                          float rNorm, const float* img, const thx_ctf_attr* attr, const double* dfac, float pixelSize,
                          const double* rotMat, const double* trans, int nImg, void* stream)
{
    return norm_residual_impl(norm, volumes, volIdx, vdim, pf, idim, projR, rL, rNorm, img, attr, dfac, pixelSize, rotMat, trans, nImg, stream, false);
}

int thx_norm_residual_packed_dev(float* norm, const float* cells, const int* volIdx, int vdim, int pf, int idim, int projR, float rL,
                                 float rNorm, const float* img, const thx_ctf_attr* attr, const double* dfac, float pixelSize,
                                 const double* rotMat, const double* trans, int nImg, void* stream)
{
    return norm_residual_impl(norm, cells, volIdx, vdim, pf, idim, projR, rL, rNorm, img, attr, dfac, pixelSize, rotMat, trans, nImg, stream, true);
}

int thx_median_f32_dev(float* out, const float* values, int n, void* stream)
{
    THX_REQUIRE(out && values && n > 0, "bad arguments");
    hipStream_t st = as_stream(stream);
    float* sorted = reinterpret_cast<float*>(scratch(st, 15, (size_t)n * sizeof(float)));
    THX_REQUIRE(sorted, "device scratch allocation failed");
    size_t tb = 0;
    THX_CHECK(rocprim::radix_sort_keys(nullptr, tb, values, sorted, (size_t)n, 0u, 32u, st));
    void* tmp = scratch(st, 16, tb);
    THX_REQUIRE(tmp, "device scratch allocation failed");
    THX_CHECK(rocprim::radix_sort_keys(tmp, tb, values, sorted, (size_t)n, 0u, 32u, st));
    hipLaunchKernelGGL(k_median_sorted, dim3(1), dim3(1), 0, st, out, sorted, n);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_norm_scale_dev(float* img, float* imgOri, const float* norm, const float* median, int idim, int nImg, void* stream)
{
    THX_REQUIRE(img && imgOri && norm && median && idim > 0, "bad arguments");
    if (nImg <= 0) return 0;
    const size_t imgSize = (size_t)idim * (idim / 2 + 1);
    for (int l0 = 0; l0 < nImg; l0 += 65535) {
        const int nl = nImg - l0 < 65535 ? nImg - l0 : 65535;
        hipLaunchKernelGGL(k_norm_scale, dim3(16, nl), dim3(256), 0, as_stream(stream), reinterpret_cast<float2*>(img) + (size_t)l0 * imgSize,
                           reinterpret_cast<float2*>(imgOri) + (size_t)l0 * imgSize, norm + l0, median, imgSize);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_sigma_accum_dev(float* sigM, float* sigN, float* svd, const float* spec, const int* groupID_host, int nImg,
                        int nGroup, int rSig, int group, void* stream)
{
    THX_REQUIRE(sigM && sigN && svd && spec && nGroup > 0 && rSig > 0, "bad arguments");
    THX_REQUIRE(!group || groupID_host, "groupID required when group != 0");
    if (nImg <= 0) return 0;
    hipStream_t st = as_stream(stream);
    const int nG = group ? nGroup : 1;
    // counting sort of the images by (1-based) group, on the host: the reference keeps _groupID on the host too
    std::vector<int> tab((size_t)nImg + nG + 1, 0);
    int* start = tab.data();
    int* order = tab.data() + nG + 1;
    for (int l = 0; l < nImg; l++) {
        const int g = group ? groupID_host[l] - 1 : 0;
        THX_REQUIRE(g >= 0 && g < nG, "groupID out of range (1-based, as Optimiser::_groupID)");
        start[g + 1]++;
    }
    for (int g = 0; g < nG; g++) start[g + 1] += start[g];
    {
        std::vector<int> fill(start, start + nG);
        for (int l = 0; l < nImg; l++) order[fill[group ? groupID_host[l] - 1 : 0]++] = l;
    }
    int* d = reinterpret_cast<int*>(scratch(st, 5, tab.size() * sizeof(int)));
    THX_REQUIRE(d, "device scratch allocation failed");
    // synchronous copy: tab is a stack-lifetime host buffer
    THX_CHECK(hipStreamSynchronize(st));
    THX_CHECK(hipMemcpy(d, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sigma_accum, dim3(nG), dim3(1024), 0, st, sigM, sigN, svd, spec, d + nG + 1, d, rSig);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_sigma_final_dev(float* sig, float* sigRcp, const float* sigM, const float* sigN, const float* svd, int nGroup,
                        int rSig, int group, float maskRadius, int size, float pixelSize, void* stream)
{
    THX_REQUIRE(sig && sigRcp && sigM && sigN && svd && nGroup > 0 && rSig > 0, "bad arguments");
    const float q = maskRadius / (size * pixelSize);
    const float alpha = (float)sqrt(3.14159265358979323846 * (double)q * (double)q);
    const int n = nGroup * rSig;
    hipLaunchKernelGGL(k_sigma_final, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), sig, sigRcp, sigM, sigN,
                       svd, nGroup, rSig, group, alpha);
    THX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
