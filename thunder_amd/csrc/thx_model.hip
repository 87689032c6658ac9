// thx_model.hip -- Model::compareTwoHemispheres on the device: the gold-standard FSC of the two half
// maps incl. its mask-corrected form (random-phase substitution), and the low-resolution averaging of the two halves.
// Reference behaviour: src/Model.cpp:307-700 (3-D mode), randomPhase src/Functions/Spectrum.cpp:365-386, resP :339-363,
// softMask(Volume&, r, ew) / softMask(dst, src, alpha, bg) src/Functions/Mask.cpp:470-531, FSC src/Functions/Spectrum.cpp:302-337.
// The half maps never leave HBM: forward / backward transforms by rocFFT (thx_fft3d_*), everything else element-wise kernels.
#include <vector>

#include "thx_common.h"
#include "thx_philox.h"

namespace thx {

// softMask(Volume& mask, r, ew): src/Functions/Mask.cpp:470-486.  Volume in the in-memory (wrapped-index) layout [N][N][N].
__global__ __launch_bounds__(256) void k_core_mask(float* __restrict__ mask, int N, float r, float ew)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)N * N * N) return;
    const int iw = (int)(e % N), jw = (int)((e / N) % N), kw = (int)(e / ((size_t)N * N));
    const int i = iw >= N / 2 ? iw - N : iw, j = jw >= N / 2 ? jw - N : jw, k = kw >= N / 2 ? kw - N : kw;
    const float u = (float)gsl_hypot3_((double)i, (double)j, (double)k);   // NORM_3 narrowed to RFLOAT
    float v;
    if (u > r + ew) v = 0.f;
    else if (u >= r) v = (float)(0.5 + 0.5 * cos((u - r) / ew * 3.14159265358979323846));
    else v = 1.f;
    mask[e] = v;
}

// softMask(Volume& dst, const Volume& src, r, ew, bg), src/Functions/Mask.cpp:499-521, in place: beyond r + ew the
// background, inside r untouched, in between bg * w + src * (1 - w) with w = 0.5 - 0.5 cos((u - r) / ew * pi) narrowed to RFLOAT
__global__ __launch_bounds__(256) void k_soft_mask_volume(float* __restrict__ vol, int N, float r, float ew, float bg)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)N * N * N) return;
    const int iw = (int)(e % N), jw = (int)((e / N) % N), kw = (int)(e / ((size_t)N * N));
    const int i = iw >= N / 2 ? iw - N : iw, j = jw >= N / 2 ? jw - N : jw, k = kw >= N / 2 ? kw - N : kw;
    const float u = (float)gsl_hypot3_((double)i, (double)j, (double)k);   // NORM_3 narrowed to RFLOAT
    if (u > r + ew) vol[e] = bg;
    else if (u >= r) {
        const float w = (float)(0.5 - 0.5 * cos((u - r) / ew * 3.14159265358979323846));
        vol[e] = bg * w + vol[e] * (1 - w);
    }
}

// softMask(dst, src, alpha, bg): dst = bg * w + src * (1 - w), w = 1 - alpha  (src/Functions/Mask.cpp:510-521)
__global__ __launch_bounds__(256) void k_alpha_mask(float* dst, const float* src, const float* __restrict__ alpha, float bg, size_t n)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float w = 1 - alpha[e];
    dst[e] = bg * w + src[e] * (1 - w);
}

// randomPhase(dst, src, r): src/Functions/Spectrum.cpp:365-386.  Every STORED half-space element beyond shell r is turned
// by an independent uniform phase (the reference loops VOLUME_FOR_EACH_PIXEL_FT over the stored half and draws from GSL's
// global generator; here the phase of element e is Philox(seed, e, call, 9)).  phases (optional) receives the angles.
__global__ __launch_bounds__(256) void k_random_phase(float2* __restrict__ dst, const float2* __restrict__ src, int N, int r,
                                                      unsigned long long seed, unsigned call, float* __restrict__ phases)
{
    const int nc = N / 2 + 1;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)N * N * nc) return;
    const int i = (int)(e % nc), jw = (int)((e / nc) % N), kw = (int)(e / ((size_t)nc * N));
    const int j = jw >= N / 2 ? jw - N : jw, k = kw >= N / 2 ? kw - N : kw;
    const int u = (int)rint(gsl_hypot3_((double)i, (double)j, (double)k));   // AROUND(NORM_3(i, j, k))
    float2 v = src[e];
    float ph = 0.f;
    if (u > r) {
        double uu[4];
        draw_u4(uu, seed, (unsigned)(e & 0xFFFFFFFFu), call, 9u, (unsigned)(e >> 32));
        ph = (float)(uu[0] * 2 * 3.14159265358979323846);   // TSGSL_ran_flat(engine, 0, 2 * M_PI) as RFLOAT
        float s, c;
        sincosf(ph, &s, &c);
        v = cmul(v, make_float2(c, s));
    }
    dst[e] = v;
    if (phases) phases[e] = ph;
}

// A = B = (A + B) / 2 inside QUAD_3(i, j, k) < r^2 (src/Model.cpp:663-674); r2 < 0: everywhere (:620-627, :688-696)
__global__ __launch_bounds__(256) void k_average_halves(float2* __restrict__ A, float2* __restrict__ B, int N, float r2)
{
    const int nc = N / 2 + 1;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)N * N * nc) return;
    if (r2 >= 0.f) {
        const int i = (int)(e % nc), jw = (int)((e / nc) % N), kw = (int)(e / ((size_t)nc * N));
        const int j = jw >= N / 2 ? jw - N : jw, k = kw >= N / 2 ? kw - N : kw;
        const float q = (float)((double)i * i + (double)j * j + (double)k * k);   // QUAD_3 narrowed to RFLOAT
        if (!(q < r2)) return;
    }
    const float2 a = A[e], b = B[e];
    const float2 avg = make_float2((a.x + b.x) / 2, (a.y + b.y) / 2);
    A[e] = avg;
    B[e] = avg;
}

static unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace thx

using namespace thx;

extern "C" {

int thx_core_mask_dev(float* mask, int N, float r, float ew, void* stream)
{
    THX_REQUIRE(mask && N > 0, "bad arguments");
    hipLaunchKernelGGL(k_core_mask, dim3(nblk((size_t)N * N * N)), dim3(256), 0, as_stream(stream), mask, N, r, ew);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_soft_mask_volume_dev(float* vol, int N, float r, float ew, float bg, void* stream)
{
    THX_REQUIRE(vol && N > 0 && ew > 0, "bad arguments");
    hipLaunchKernelGGL(k_soft_mask_volume, dim3(nblk((size_t)N * N * N)), dim3(256), 0, as_stream(stream), vol, N, r, ew, bg);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_random_phase_dev(float* dst, const float* src, int N, int r, unsigned long long seed, unsigned call, float* phases,
                         void* stream)
{
    THX_REQUIRE(dst && src && N > 0, "bad arguments");
    hipLaunchKernelGGL(k_random_phase, dim3(nblk((size_t)N * N * (N / 2 + 1))), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float2*>(dst), reinterpret_cast<const float2*>(src), N, r, seed, call, phases);
    THX_LAUNCH_CHECK();
    return 0;
}

// resP(fsc, thres, pf, rL, inverse = false), src/Functions/Spectrum.cpp:339-363
static int res_p(const float* fsc, int n, float thres, int pf, int rL)
{
    int result;
    for (result = rL; result < n; result++)
        if (fsc[result] < thres) break;
    result--;
    return result / pf;
}

int thx_compare_hemispheres_dev(float* A, float* B, int N, int rU, float* fscHost, const float* maskRL, float coreR, float ew,
                                int avgFlag, int avgR, unsigned long long seed, unsigned call, int* randomPhaseThresOut,
                                void* stream)
{
    THX_REQUIRE(A && B && N > 0 && (N % 2) == 0 && rU > 0 && rU <= N / 2 + 1, "bad arguments");
    hipStream_t st = as_stream(stream);
    const size_t nFT = (size_t)N * N * (N / 2 + 1), nRL = (size_t)N * N * N;
    std::vector<float> fsc(rU, 0.f);
    float* fscD = reinterpret_cast<float*>(scratch(st, 8, 3 * (size_t)rU * sizeof(float)));
    THX_REQUIRE(fscD, "device scratch allocation failed");
    if (fscHost) {
        const bool masked = maskRL != nullptr || coreR > 0.f;
        THX_RC(thx_fsc_dev(fscD, rU, A, B, N, st));                                  // fscUnmask (or THE fsc when unmasked)
        THX_CHECK(hipMemcpyAsync(fsc.data(), fscD, rU * sizeof(float), hipMemcpyDeviceToHost, st));
        THX_CHECK(hipStreamSynchronize(st));
        if (masked) {
            // work volumes: two FTs, one real volume, the core mask
            float* ws = reinterpret_cast<float*>(scratch(st, 9, (4 * nFT + 2 * nRL) * sizeof(float)));
            THX_REQUIRE(ws, "device scratch allocation failed");
            float *ftA = ws, *ftB = ws + 2 * nFT, *rl = ws + 4 * nFT, *mask = ws + 4 * nFT + nRL;
            const float* alpha = maskRL;
            if (!alpha) {   // _coreFSC: softMask(mask, _coreR, EDGE_WIDTH_RL)
                THX_RC(thx_core_mask_dev(mask, N, coreR, ew, st));
                alpha = mask;
            }
            const int rpThres = res_p(fsc.data(), rU, 0.8f, 1, 1);                   // src/Model.cpp:431
            if (randomPhaseThresOut) *randomPhaseThresOut = rpThres;
            // FSC of the masked, phase-randomised halves
            THX_RC(thx_random_phase_dev(ftA, A, N, rpThres, seed, call, nullptr, st));
            THX_RC(thx_random_phase_dev(ftB, B, N, rpThres, seed, call + 1, nullptr, st));
            for (float* ft : {ftA, ftB}) {
                THX_RC(thx_fft3d_bw_dev(ft, rl, N, st));
                hipLaunchKernelGGL(k_alpha_mask, dim3(nblk(nRL)), dim3(256), 0, st, rl, rl, alpha, 0.f, nRL);
                THX_RC(thx_fft3d_fw_dev(rl, ft, N, st));
            }
            THX_RC(thx_fsc_dev(fscD + rU, rU, ftA, ftB, N, st));                     // fscRFMask
            // FSC of the masked halves
            const float* src[2] = {A, B};
            float* dstFT[2] = {ftA, ftB};
            for (int h = 0; h < 2; h++) {
                THX_CHECK(hipMemcpyAsync(dstFT[h], src[h], 2 * nFT * sizeof(float), hipMemcpyDeviceToDevice, st));
                THX_RC(thx_fft3d_bw_dev(dstFT[h], rl, N, st));
                hipLaunchKernelGGL(k_alpha_mask, dim3(nblk(nRL)), dim3(256), 0, st, rl, rl, alpha, 0.f, nRL);
                THX_RC(thx_fft3d_fw_dev(rl, dstFT[h], N, st));
            }
            THX_RC(thx_fsc_dev(fscD + 2 * rU, rU, ftA, ftB, N, st));                 // fscMask
            THX_LAUNCH_CHECK();
            std::vector<float> rf(rU), mk(rU);
            THX_CHECK(hipMemcpyAsync(rf.data(), fscD + rU, rU * sizeof(float), hipMemcpyDeviceToHost, st));
            THX_CHECK(hipMemcpyAsync(mk.data(), fscD + 2 * rU, rU * sizeof(float), hipMemcpyDeviceToHost, st));
            THX_CHECK(hipStreamSynchronize(st));
            for (int i = 0; i < rU; i++)   // "true FSC", src/Model.cpp:553-561
                fsc[i] = i < rpThres + 2 ? mk[i] : (mk[i] - rf[i]) / (1 - rf[i]);
        }
        memcpy(fscHost, fsc.data(), rU * sizeof(float));
    }
    if (avgFlag) {
        const float r2 = avgR >= 0 ? pow2f_((float)avgR) : -1.f;
        hipLaunchKernelGGL(k_average_halves, dim3(nblk(nFT)), dim3(256), 0, st, reinterpret_cast<float2*>(A),
                           reinterpret_cast<float2*>(B), N, r2);
        THX_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
