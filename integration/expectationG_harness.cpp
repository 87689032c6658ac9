// expectationG_harness.cpp -- the CALLER side of the local search of the reference's -DGPU_VERSION build, restated as a C++ OpenMP loop
// over the C ABI (include/thunder_amd.h): what an unchanged Optimiser::expectationG does around the plug-in surface
// (src/Optimiser.cpp:2180-3393), so that the drop-in path can be timed -- and tested -- the way the reference drives it:
//   * host threads over the images (`#pragma omp parallel for`, :2790), each with its own page-locked staging arrays
//     (ExpectLocalHostA, :2353-2400) and device slot (ExpectLocalP, :2840);
//   * per image and phase: the filter's support points copied into the staging arrays (:2905-2950), then -- UNDER A PER-GPU LOCK,
//     omp_set_lock(&mtx[gpuIdx]) (:2960) -- ExpectLocalRTD, ExpectLocalPreI3D, ExpectLocalM; the lock is released (:3077) and the
//     weights are read (setUC / setUR / setUT, :3079-3100).
// `lock` = 1 reproduces that lock (one image-phase in flight per GPU: the latency-bound form an unchanged Optimiser.cpp gets);
// `lock` = 0 lets the threads run free on their own ManagedCalPoints / streams (what a caller that drops the lock would get).
// The particle filter's update between phases is not part of this harness (the same support points in every phase: the same work).
// Test / bench infrastructure: built into thunder_amd/lib/libthx_harness.so by thunder_amd/build.py, never linked into the library.
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "thunder_amd.h"

extern "C" {

struct thx_harness_args {
    int gpu, N, pf, nPxl, nImg, mLR, mLT, phases, threads, lock;
    const float* volume;          // host padded FT [P][P][P/2+1] complex64 (Projector::_projectee3D)
    const int *iCol, *iRow;       // host [nPxl]
    const float *datP, *ctfP, *sigP;   // host rows [nImg][nPxl] (complex64 / f32 / f32): Optimiser::_datP / _ctfP / _sigRcpP
    const double *quat, *tran;    // host [nImg][mLR][4], [nImg][mLT][2]: the filter's support points
    const float* attr;            // host [nImg][7] CTFAttr (phaseShift, amplitudeContrast are handed to ExpectLocalPreI3D)
    float* wR;                    // host out [nImg][mLR]: the rotation weights of the LAST phase (for checks); may be NULL
    float* wT;                    // host out [nImg][mLT]; may be NULL
    double seconds;               // out: wall time of the parallel loop (setup / teardown excluded)
    double callSeconds[4];        // out: host time inside ExpectLocalP / RTD / PreI3D / M, summed over the threads
};

#define H_RC(expr) do { int _rc = (expr); if (_rc) { std::fprintf(stderr, "harness: %s failed: %s\n", #expr, thx_last_error()); return _rc; } } while (0)

int thx_harness_expectation_local(thx_harness_args* a)
{
    const int P = a->N * a->pf, nPxl = a->nPxl, nR = a->mLR, nT = a->mLT, T = a->threads > 0 ? a->threads : 1;
    int *deviCol = nullptr, *deviRow = nullptr;
    float *devdatP = nullptr, *devctfP = nullptr, *devdefO = nullptr, *devsigP = nullptr, *devfreQ = nullptr;
    thx_texture* mgr = nullptr;
    H_RC(thx_ExpectPreidx_host(a->gpu, &deviCol, &deviRow, a->iCol, a->iRow, nPxl));
    H_RC(thx_ExpectLocalIn_host(a->gpu, &devdatP, &devctfP, &devdefO, &devsigP, nPxl, T, 0));
    H_RC(thx_texture_create(&mgr, 1, P, a->gpu));
    H_RC(thx_ExpectLocalV3D_host(a->gpu, mgr, a->volume, P));
    struct Ctx { thx_calpoint* mcp; float *wC, *wR, *wT, *wD; double *oldR, *oldT, *oldD, *trans, *rot, *dpara; };
    std::vector<Ctx> ctx(T);
    for (int t = 0; t < T; t++) {
        Ctx& c = ctx[t];
        H_RC(thx_calpoint_create(&c.mcp, 1, 0, a->gpu, nR, nT, 1, nPxl));
        H_RC(thx_ExpectLocalHostA_host(a->gpu, &c.wC, &c.wR, &c.wT, &c.wD, &c.oldR, &c.oldT, &c.oldD, &c.trans, &c.rot, &c.dpara, nR, nT, 1, 0));
        for (int i = 0; i < nR; i++) c.oldR[i] = 1.0 / nR;
        for (int i = 0; i < nT; i++) c.oldT[i] = 1.0 / nT;
        c.oldD[0] = 1.0;
    }
    omp_lock_t mtx;
    omp_init_lock(&mtx);
    int err = 0;
    double acc[4] = {0, 0, 0, 0};
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for num_threads(T) schedule(dynamic)
    for (int l = 0; l < a->nImg; l++) {
        if (err) continue;
        const int t = omp_get_thread_num();
        Ctx& c = ctx[t];
        double loc[4] = {0, 0, 0, 0};
        auto c0 = now();
        int rc = thx_ExpectLocalP_host(a->gpu, devdatP, devctfP, devdefO, devsigP, a->datP, a->ctfP, nullptr, a->sigP, t, l, nPxl, 0);
        loc[0] += secs(c0, now());
        for (int ph = 0; ph < a->phases && !rc; ph++) {
            std::memcpy(c.trans, a->tran + (size_t)l * nT * 2, (size_t)nT * 2 * sizeof(double));
            std::memcpy(c.rot, a->quat + (size_t)l * nR * 4, (size_t)nR * 4 * sizeof(double));
            if (a->lock) omp_set_lock(&mtx);
            auto c1 = now();
            rc = thx_ExpectLocalRTD_host(a->gpu, c.mcp, c.oldR, c.oldT, c.oldD, c.trans, c.rot, c.dpara);
            auto c2 = now();
            if (!rc) rc = thx_ExpectLocalPreI3D_host(a->gpu, t, mgr, c.mcp, devdefO, devfreQ, deviCol, deviRow, a->attr[(size_t)l * 7 + 6],
                                                     a->attr[(size_t)l * 7 + 5], 0.f, 0.f, a->pf, a->N, P, nPxl, 1);
            auto c3 = now();
            if (!rc) rc = thx_ExpectLocalM_host(a->gpu, t, c.mcp, devdatP, devctfP, devsigP, c.wC, c.wR, c.wT, c.wD, 1.0, nPxl);
            auto c4 = now();
            loc[1] += secs(c1, c2); loc[2] += secs(c2, c3); loc[3] += secs(c3, c4);
            if (a->lock) omp_unset_lock(&mtx);
        }
#pragma omp critical(thx_harness_acc)
        { for (int q = 0; q < 4; q++) acc[q] += loc[q]; }
        if (!rc) {
            if (a->wR) std::memcpy(a->wR + (size_t)l * nR, c.wR, nR * sizeof(float));
            if (a->wT) std::memcpy(a->wT + (size_t)l * nT, c.wT, nT * sizeof(float));
        } else {
#pragma omp critical
            { if (!err) { err = rc; std::fprintf(stderr, "harness: image %d: %s\n", l, thx_last_error()); } }
        }
    }
    a->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int q = 0; q < 4; q++) a->callSeconds[q] = acc[q];
    omp_destroy_lock(&mtx);
    for (int t = 0; t < T; t++) {
        Ctx& c = ctx[t];
        (void)thx_ExpectLocalHostF_host(a->gpu, &c.wC, &c.wR, &c.wT, &c.wD, &c.oldR, &c.oldT, &c.oldD, &c.trans, &c.rot, &c.dpara, 0);
        (void)thx_calpoint_destroy(c.mcp);
    }
    (void)thx_ExpectLocalFin_host(a->gpu, &devdatP, &devctfP, &devdefO, &devfreQ, &devsigP, 0);
    (void)thx_ExpectFreeIdx_host(a->gpu, &deviCol, &deviRow);
    (void)thx_texture_destroy(mgr);
    return err;
}

}  // extern "C"
