// classify.cpp -- a torch-free C++ caller driving a K-class classification through the C ABI only (include/thunder_amd.h):
// K synthetic references, images = CTF x slice of a random class at a scanned rotation x ramp of a scanned shift + noise (signal
// rows through the library's own project / CTF / translate), then thx_refine_create (nK = K, THX_SEARCH_GLOBAL) / set_particles /
// set_reference / set_grid / reset -> thx_refine_iterate: global scan over the K classes, class of every image, support points,
// local phases against the assigned reference, sigma update, multi-reference insertion, two reconstructions per class and half,
// per-class FSC, averaging, refresh (src/Optimiser.cpp:631-1660, 3405-3530, 7038-7760); then thx_refine_set_search_type(LOCAL)
// and a second iteration -- what `Optimiser::run` does per iteration of a 3-D classification.  Checks: the classes are
// recovered, every class map agrees with its own generating map.
// Build: g++ -std=c++17 -I include tests/cpp/classify.cpp -L thunder_amd/lib -lthunder_amd -Wl,-rpath,...
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "thunder_amd.h"

#define CK(expr)                                                                                       \
    do {                                                                                               \
        int _rc = (expr);                                                                              \
        if (_rc) { fprintf(stderr, "FAILED %s -> %d: %s\n", #expr, _rc, thx_last_error()); exit(2); } \
    } while (0)

typedef std::complex<float> cf;

template <typename T>
static T* dev_alloc(size_t n)
{
    void* p = nullptr;
    CK(thx_malloc_dev(&p, n * sizeof(T)));
    return reinterpret_cast<T*>(p);
}
template <typename T>
static T* dev_upload(const std::vector<T>& v)
{
    T* p = dev_alloc<T>(v.size());
    CK(thx_memcpy_h2d(p, v.data(), v.size() * sizeof(T)));
    return p;
}
template <typename T>
static std::vector<T> dev_download(const T* p, size_t n)
{
    std::vector<T> v(n);
    CK(thx_memcpy_d2h(v.data(), p, n * sizeof(T)));
    return v;
}

static void blob_map(float* m, int N, unsigned seed)
{
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::normal_distribution<float> G(0.f, 1.f);
    for (int b = 0; b < 8; b++) {
        float c[3] = {G(g), G(g), G(g)};
        const float nn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), rr = U(g) * 0.3f * N;
        for (int a = 0; a < 3; a++) c[a] = c[a] / nn * rr;
        const float s = 2.0f + 1.5f * U(g), amp = 0.5f + 0.5f * U(g);
        for (int k = 0; k < N; k++)
            for (int j = 0; j < N; j++)
                for (int i = 0; i < N; i++) {
                    const float z = (float)(k < N / 2 ? k : k - N), y = (float)(j < N / 2 ? j : j - N), x = (float)(i < N / 2 ? i : i - N);
                    const float d2 = (x - c[0]) * (x - c[0]) + (y - c[1]) * (y - c[1]) + (z - c[2]) * (z - c[2]);
                    m[((size_t)k * N + j) * N + i] += amp * std::exp(-d2 / (2 * s * s));
                }
    }
}

int main()
{
    int nDev = 0;
    CK(thx_device_count(&nDev));
    if (nDev < 1) { fprintf(stderr, "no GPU visible\n"); return 1; }
    CK(thx_set_device(0));
    const int N = 32, pf = 2, P = N * pf, nc = N / 2 + 1, K = 2, n = 240, nR = 200, nT = 6, rScan = 10, mLR = 32, mLT = 5, nPhase = 2, mReco = 10;
    const float pixelSize = 1.32f;
    const size_t mapN = (size_t)N * N * N, volN = (size_t)P * P * (P / 2 + 1);

    // ---- K generating maps and their projector volumes ----
    std::vector<float> refs(K * mapN, 0.f);
    for (int k = 0; k < K; k++) blob_map(refs.data() + k * mapN, N, 11 + 17 * k);
    float* refD = dev_upload(refs);
    thx_reco* plan = nullptr;
    CK(thx_reco_create(&plan, N, N, pf, 1.9f, 15.0f));
    float* vols = dev_alloc<float>(K * volN * 2);
    for (int k = 0; k < K; k++) CK(thx_reco_set_projectee_dev(plan, refD + k * mapN, vols + k * volN * 2, nullptr));
    const int rU = N / 2 - 2, cap = (rU + 2) * (2 * rU + 2);
    std::vector<int> iCol(cap), iRow(cap);
    int nPxl = 0;
    CK(thx_pixel_list_host(N, rU, 0, 0, iCol.data(), iRow.data(), nullptr, nullptr, &nPxl));
    iCol.resize(nPxl); iRow.resize(nPxl);
    int *iColD = dev_upload(iCol), *iRowD = dev_upload(iRow);

    // ---- the scanned grid; every image sits on one grid point of one class ----
    std::mt19937_64 g(2024);
    std::normal_distribution<double> G(0.0, 1.0);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<double> quat(nR * 4), shifts(nT * 2);
    for (int r = 0; r < nR; r++) {
        double q[4] = {G(g), G(g), G(g), G(g)}, nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int a = 0; a < 4; a++) quat[4 * r + a] = q[a] / nn;
    }
    for (int t = 0; t < nT; t++) { shifts[2 * t] = 1.5 * G(g); shifts[2 * t + 1] = 1.5 * G(g); }
    std::vector<int> clsTrue(n), rTrue(n), tTrue(n);
    std::vector<thx_ctf_attr> attr(n);
    for (int l = 0; l < n; l++) {
        clsTrue[l] = (int)(U(g) * K) % K; rTrue[l] = (int)(U(g) * nR) % nR; tTrue[l] = (int)(U(g) * nT) % nT;
        const float dU = (float)(1.0e4 + 2.0e4 * U(g));
        attr[l] = {3e5f, dU, dU + (float)(300 * G(g)), (float)(3.14159 * U(g)), 2.7e7f, 0.1f, 0.f};
    }
    double *quatD = dev_upload(quat), *shiftD = dev_upload(shifts);
    double* rotD = dev_alloc<double>((size_t)nR * 9);
    CK(thx_rotmat_dev(quatD, rotD, nR, nullptr));
    float* slD = dev_alloc<float>((size_t)nR * nPxl * 2);
    float* rampD = dev_alloc<float>((size_t)nT * nPxl * 2);
    CK(thx_translate_dev(rampD, shiftD, nT, iColD, iRowD, nPxl, N, nullptr));
    thx_ctf_attr* attrD = dev_upload(attr);
    float* ctfD = dev_alloc<float>((size_t)n * nPxl);
    CK(thx_ctf_dev(ctfD, attrD, nullptr, pixelSize, iColD, iRowD, nPxl, N, n, nullptr));
    CK(thx_device_sync());
    std::vector<cf> ramp = dev_download(reinterpret_cast<cf*>(rampD), (size_t)nT * nPxl);
    std::vector<float> ctf = dev_download(ctfD, (size_t)n * nPxl);
    std::vector<cf> dat((size_t)n * nPxl);
    for (int k = 0; k < K; k++) {
        CK(thx_project_dev(vols + k * volN * 2, slD, rotD, iColD, iRowD, nR, pf, P, nPxl, nullptr));
        CK(thx_device_sync());
        std::vector<cf> sl = dev_download(reinterpret_cast<cf*>(slD), (size_t)nR * nPxl);
        for (int l = 0; l < n; l++)
            if (clsTrue[l] == k)
                for (int p = 0; p < nPxl; p++)
                    dat[(size_t)l * nPxl + p] = sl[(size_t)rTrue[l] * nPxl + p] * ramp[(size_t)tTrue[l] * nPxl + p] * ctf[(size_t)l * nPxl + p];
    }
    double pSig = 0;
    for (const cf& x : dat) pSig += std::norm(x);
    pSig /= (double)dat.size();
    const double sigma2 = pSig / 2.0 / 2.0;   // SNR 2: variance per real component of a coefficient
    std::normal_distribution<float> Gf(0.f, (float)std::sqrt(sigma2));
    // full image FTs: noise everywhere, signal on the listed pixels (+ the Hermitian mirror of the kx = 0 column)
    std::vector<int> iPxl(nPxl);
    {
        std::vector<int> c2(nPxl), r2(nPxl), s2(nPxl);
        int np2 = 0;
        CK(thx_pixel_list_host(N, rU, 0, 0, c2.data(), r2.data(), iPxl.data(), s2.data(), &np2));
    }
    std::vector<cf> img((size_t)n * N * nc);
    for (int l = 0; l < n; l++) {
        cf* I = img.data() + (size_t)l * N * nc;
        for (size_t e = 0; e < (size_t)N * nc; e++) I[e] = cf(Gf(g), Gf(g));
        for (int p = 0; p < nPxl; p++) {
            I[iPxl[p]] += dat[(size_t)l * nPxl + p];
            if (iCol[p] == 0 && iRow[p] > 0) I[(size_t)(N - iRow[p]) * nc] = std::conj(I[iPxl[p]]);
        }
        I[0] = cf(I[0].real(), 0.f);
    }
    float* imgD = reinterpret_cast<float*>(dev_upload(img));
    // the filter's loaded support points are irrelevant to a global search (the scan replaces them): the true pose, repeated
    std::vector<double> q0((size_t)n * mLR * 4), t0((size_t)n * mLT * 2);
    for (int l = 0; l < n; l++) {
        for (int m = 0; m < mLR; m++) {
            const double ang = 0.02 * (m + 1), c = std::cos(ang / 2), s_ = std::sin(ang / 2);
            const double d[4] = {c, s_ * (m % 3 == 0), s_ * (m % 3 == 1), s_ * (m % 3 == 2)};
            const double* a = &quat[4 * rTrue[l]];
            double* o = &q0[((size_t)l * mLR + m) * 4];
            o[0] = a[0] * d[0] - a[1] * d[1] - a[2] * d[2] - a[3] * d[3];
            o[1] = a[0] * d[1] + a[1] * d[0] + a[2] * d[3] - a[3] * d[2];
            o[2] = a[0] * d[2] - a[1] * d[3] + a[2] * d[0] + a[3] * d[1];
            o[3] = a[0] * d[3] + a[1] * d[2] - a[2] * d[1] + a[3] * d[0];
        }
        for (int m = 0; m < mLT; m++) {
            t0[((size_t)l * mLT + m) * 2] = shifts[2 * tTrue[l]] + 0.3 * (m - 2);
            t0[((size_t)l * mLT + m) * 2 + 1] = shifts[2 * tTrue[l] + 1] - 0.2 * (m - 2);
        }
    }
    double *q0D = dev_upload(q0), *t0D = dev_upload(t0);
    std::vector<int> gid(n, 1);

    // ---- the iteration driver: K classes, first a global search ----
    thx_refine_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.N = N; cfg.pf = pf; cfg.nImg = n; cfg.halfOfRank = -1; cfg.nHalfA = n / 2;
    cfg.mLR = mLR; cfg.mLT = mLT; cfg.nPhase = nPhase; cfg.maxPhase = 0; cfg.mReco = mReco; cfg.batch = 50;
    cfg.rL = 1; cfg.nGroup = 1; cfg.groupSig = 1; cfg.pixelOrder = 1; cfg.wgPerCU = -1;
    cfg.pixelSize = pixelSize; cfg.maskRadiusPx = 0.45f * N; cfg.sigma2Init = (float)sigma2;
    cfg.transS = 2.0; cfg.transQ = 0.05; cfg.pfL = 2.0; cfg.pfS = 0.5; cfg.peakFactorR = 1e-3;
    cfg.coreFSC = 0; cfg.goldenAverage = 1; cfg.solventFlatten = 1;
    cfg.nK = K; cfg.searchType = THX_SEARCH_GLOBAL; cfg.nR = nR; cfg.nT = nT; cfg.rScan = rScan; cfg.scanBatch = 64;
    cfg.pfSGlobal = 0.5; cfg.peakFactorC = 1.0 - 1e-2; cfg.balanceClass = 1;
    cfg.scanMinK = std::pow(std::pow((double)nR, -1.0 / 3) / 0.5, 2.0); cfg.scanMinS = 0.3;
    cfg.seed = 424242;
    thx_refine* h = nullptr;
    CK(thx_refine_create(&h, &cfg, nullptr, nullptr));
    CK(thx_refine_set_particles(h, imgD, attrD, gid.data(), q0D, t0D, nullptr));
    CK(thx_refine_set_reference(h, refD, nullptr));
    CK(thx_refine_set_grid(h, quat.data(), shifts.data(), nullptr));   // host pointers are fine here
    CK(thx_refine_reset(h, nullptr));
    thx_refine_stats st;
    thx_refine_view v;
    CK(thx_refine_get_view(h, &v));
    std::vector<float> fsc((size_t)K * (N / 2));
    CK(thx_refine_iterate(h, fsc.data(), 1, nullptr));
    CK(thx_device_sync());
    {
        std::vector<int> cls1 = dev_download(v.cls, n);
        int hit1 = 0;
        for (int l = 0; l < n; l++) hit1 += cls1[l] == clsTrue[l];
        printf("iteration 1 (global search): classes recovered %d of %d\n", hit1, n);
        if (hit1 < 0.9 * n) return 1;
    }
    CK(thx_refine_get_stats(h, &st, 1));
    const int nScanBatches = 2 * ((n / 2 + 63) / 64);
    if (st.iterations != 1 || st.scanLaunches != (long)K * nScanBatches || st.scanImages != (long)K * n) {
        fprintf(stderr, "implausible scan statistics (%ld launches, %ld images)\n", st.scanLaunches, st.scanImages);
        return 3;
    }
    // iteration 2: a local search in the assigned classes against iteration 1's maps
    CK(thx_refine_set_search_type(h, THX_SEARCH_LOCAL));
    CK(thx_refine_iterate(h, fsc.data(), 1, nullptr));
    CK(thx_device_sync());
    CK(thx_refine_get_stats(h, &st, 0));
    if (st.iterations != 1 || st.scanLaunches != 0 || st.expectLaunches != 2 * nPhase * 3 || st.insertLaunches != 2 * 3 || st.balancingRounds <= 0 ||
        st.nPxlM != nPxl || v.nImg != n || v.nK != K || v.nVol != 2 * K) {
        fprintf(stderr, "implausible driver statistics (local %ld insert %ld rounds %ld)\n", st.expectLaunches, st.insertLaunches,
                st.balancingRounds);
        return 3;
    }
    std::vector<int> cls = dev_download(v.cls, n);
    int hit = 0;
    for (int l = 0; l < n; l++) hit += cls[l] == clsTrue[l];
    printf("iteration 2 (local search, references = iteration 1's maps): classes kept %d of %d; images per class %d / %d; balancing rounds %ld; FSC shell 2: %.3f / %.3f\n",
           hit, n, st.classCount[0], st.classCount[1], st.balancingRounds, fsc[2], fsc[N / 2 + 2]);
    bool ok = hit >= 0.9 * n && st.classCount[0] + st.classCount[1] == n;
    // ---- every class map against its own generating map and against the other's ----
    float *ftA = dev_alloc<float>((size_t)N * N * nc * 2), *ftB = dev_alloc<float>((size_t)N * N * nc * 2), *fscD = dev_alloc<float>(N / 2);
    for (int k = 0; k < K; k++) {
        float own[2] = {0, 0};
        for (int o = 0; o < 2; o++) {
            CK(thx_fft3d_fw_dev(const_cast<float*>(v.mapsMAP) + k * mapN, ftA, N, nullptr));   // half 0, class k
            CK(thx_fft3d_fw_dev(refD + ((k + o) % K) * mapN, ftB, N, nullptr));
            CK(thx_fsc_dev(fscD, N / 2, ftA, ftB, N, nullptr));
            CK(thx_device_sync());
            std::vector<float> f = dev_download(fscD, N / 2);
            own[o] = (f[1] + f[2] + f[3] + f[4]) / 4;
        }
        printf("class %d map: mean FSC over shells 1-4 with its own reference %.3f, with the other class's %.3f\n", k, own[0], own[1]);
        ok = ok && own[0] > 0.75f && own[0] > own[1] + 0.1f;   // (60 images per class and half at SNR 2; measured 0.80 against 0.59)
    }
    CK(thx_refine_destroy(h));
    CK(thx_reco_destroy(plan));
    for (void* p : {(void*)refD, (void*)vols, (void*)iColD, (void*)iRowD, (void*)quatD, (void*)shiftD, (void*)rotD, (void*)slD, (void*)rampD,
                    (void*)attrD, (void*)ctfD, (void*)imgD, (void*)q0D, (void*)t0D, (void*)ftA, (void*)ftB, (void*)fscD})
        CK(thx_free_dev(p));
    if (!ok) return 1;
    printf("OK (one rank, K = %d classes)\n", K);
    return 0;
}
