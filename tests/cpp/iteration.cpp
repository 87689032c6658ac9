// iteration.cpp -- a torch-free C++ caller driving the EM iteration through the C ABI only (include/thunder_amd.h):
// synthetic particles -> thx_refine_create / set_particles / set_reference / reset -> thx_refine_iterate x 2
// (rows -> particle-filter phases -> sigma update -> draws + insertion -> half-set reduce -> reconstruct -> FSC ->
// reconstruct -> projector refresh -> re-centre + re-mask), then checks the half-map FSC and the agreement of a half map
// with the generating map.  This is the reference's HOT LOOP B / HOT LOOP C sequencing (src/Optimiser.cpp:1162-1660,
// 7038-7241) as a C++ host program, what `Optimiser::run` would call per iteration.
// With THX_COMM_TRANSPORT=shm in the environment it forks two ranks that SHARE device 0 and exchange over the library's test-only
// shared-memory transport (what tests/test_native_gpu.py::test_cpp_iteration_driver_two_ranks runs on the 1-GPU box).
// With >= 2 visible GPUs it forks two ranks (one per GPU, one half-set each) whose communicators are bootstrapped from a
// unique id sent through a pipe (the reference: MPI_Bcast, gpu/src/cuthunder.cu:4192-4206) and which exchange the half maps
// over RCCL; with one GPU both halves live in one process.
// Build: g++ -std=c++17 -I include tests/cpp/iteration.cpp -L thunder_amd/lib -lthunder_amd -Wl,-rpath,...
#include <sys/wait.h>
#include <unistd.h>

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

static std::vector<float> blob_map(int N, unsigned seed)
{
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::normal_distribution<float> G(0.f, 1.f);
    std::vector<float> m((size_t)N * N * N, 0.f);
    for (int b = 0; b < 10; b++) {
        float c[3] = {G(g), G(g), G(g)};
        const float nn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), rr = U(g) * 0.3f * N;
        for (int a = 0; a < 3; a++) c[a] = c[a] / nn * rr;
        const float s = 1.5f + 2.5f * U(g) * N / 256.f + 1.f, amp = 0.5f + 0.5f * U(g);
        for (int k = 0; k < N; k++)
            for (int j = 0; j < N; j++)
                for (int i = 0; i < N; i++) {
                    const float z = (float)(k < N / 2 ? k : k - N), y = (float)(j < N / 2 ? j : j - N), x = (float)(i < N / 2 ? i : i - N);
                    const float d2 = (x - c[0]) * (x - c[0]) + (y - c[1]) * (y - c[1]) + (z - c[2]) * (z - c[2]);
                    m[((size_t)k * N + j) * N + i] += amp * std::exp(-d2 / (2 * s * s));
                }
    }
    return m;
}

struct Result { float fscHalf[4]; float fscTruth[4]; };
static bool g_oneDevice = false;   // THX_COMM_TRANSPORT=shm: both ranks on device 0 (the test-only shared-memory transport)

static Result run_rank(int rank, int world, const unsigned char* uid /*128 bytes or NULL*/)
{
    CK(thx_set_device(world > 1 && !g_oneDevice ? rank : 0));
    const int N = 32, pf = 2, P = N * pf, nc = N / 2 + 1, nTotal = 480, mLR = 40, mLT = 5, nPhase = 2, mReco = 10;
    const int n = world > 1 ? nTotal / world : nTotal;
    const float pixelSize = 1.32f;
    thx_comm* wcomm = nullptr;
    if (world > 1) CK(thx_comm_init(&wcomm, uid, rank, world));

    // ---- generating map, its projector volume, the rL = 0 pixel list ----
    std::vector<float> ref = blob_map(N, 7);
    float* refD = dev_upload(ref);
    thx_reco* plan = nullptr;
    CK(thx_reco_create(&plan, N, N, pf, 1.9f, 15.0f));
    float* vol = dev_alloc<float>((size_t)P * P * (P / 2 + 1) * 2);
    CK(thx_reco_set_projectee_dev(plan, refD, vol, nullptr));
    const int rU = N / 2 - 2, cap = (rU + 2) * (2 * rU + 2);
    std::vector<int> iCol(cap), iRow(cap), iPxl(cap), iSig(cap);
    int nPxl = 0;
    CK(thx_pixel_list_host(N, rU, 0, 0, iCol.data(), iRow.data(), iPxl.data(), iSig.data(), &nPxl));
    iCol.resize(nPxl); iRow.resize(nPxl); iPxl.resize(nPxl);
    int *iColD = dev_upload(iCol), *iRowD = dev_upload(iRow);

    // ---- particles of this rank: pose, shift, CTF; signal rows through the library's own project / CTF / translate ----
    std::mt19937_64 g(1000 + 7919 * rank);
    std::normal_distribution<double> G(0.0, 1.0);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<double> quat(n * 4), shift(n * 2);
    std::vector<thx_ctf_attr> attr(n);
    for (int l = 0; l < n; l++) {
        double q[4] = {G(g), G(g), G(g), G(g)}, nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int a = 0; a < 4; a++) quat[4 * l + a] = q[a] / nn;
        shift[2 * l] = 1.5 * G(g); shift[2 * l + 1] = 1.5 * G(g);
        const float dU = (float)(1.0e4 + 2.0e4 * U(g));
        attr[l] = {3e5f, dU, dU + (float)(300 * G(g)), (float)(3.14159 * U(g)), 2.7e7f, 0.1f, 0.f};
    }
    double* quatD = dev_upload(quat);
    double* rotD = dev_alloc<double>((size_t)n * 9);
    CK(thx_rotmat_dev(quatD, rotD, n, nullptr));
    float* slD = dev_alloc<float>((size_t)n * nPxl * 2);
    CK(thx_project_dev(vol, slD, rotD, iColD, iRowD, n, pf, P, nPxl, nullptr));
    thx_ctf_attr* attrD = dev_upload(attr);
    float* ctfD = dev_alloc<float>((size_t)n * nPxl);
    CK(thx_ctf_dev(ctfD, attrD, nullptr, pixelSize, iColD, iRowD, nPxl, N, n, nullptr));
    double* shiftD = dev_upload(shift);
    float* rampD = dev_alloc<float>((size_t)n * nPxl * 2);
    CK(thx_translate_dev(rampD, shiftD, n, iColD, iRowD, nPxl, N, nullptr));
    CK(thx_device_sync());
    std::vector<cf> sl = dev_download(reinterpret_cast<cf*>(slD), (size_t)n * nPxl), ramp = dev_download(reinterpret_cast<cf*>(rampD), (size_t)n * nPxl);
    std::vector<float> ctf = dev_download(ctfD, (size_t)n * nPxl);
    double pSig = 0;
    for (size_t e = 0; e < (size_t)n * nPxl; e++) { sl[e] = sl[e] * ramp[e] * ctf[e]; pSig += std::norm(sl[e]); }
    pSig /= (double)n * nPxl;
    const double snr = 0.5, sigma2 = pSig / snr / 2.0;   // variance per real component of an FT coefficient
    std::vector<cf> img((size_t)n * N * nc);
    std::normal_distribution<float> Gf(0.f, (float)std::sqrt(sigma2));
    for (int l = 0; l < n; l++) {
        cf* I = img.data() + (size_t)l * N * nc;
        for (size_t e = 0; e < (size_t)N * nc; e++) I[e] = cf(Gf(g), Gf(g));
        for (int p = 0; p < nPxl; p++) {
            I[iPxl[p]] += sl[(size_t)l * nPxl + p];
            if (iCol[p] == 0 && iRow[p] > 0) I[(size_t)(N - iRow[p]) * nc] = std::conj(I[iPxl[p]]);   // Hermitian mirror of kx = 0
        }
        I[0] = cf(I[0].real(), 0.f);
    }
    float* imgD = reinterpret_cast<float*>(dev_upload(img));

    // ---- initial support points of the particle filter: the pose +- small perturbations ----
    std::vector<double> q0((size_t)n * mLR * 4), t0((size_t)n * mLT * 2);
    for (int l = 0; l < n; l++) {
        for (int m = 0; m < mLR; m++) {
            double ax[3] = {G(g), G(g), G(g)}, an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            const double ang = m == 0 ? 0.0 : 0.04 * G(g), c = std::cos(ang / 2), s = std::sin(ang / 2);
            const double d[4] = {c, s * ax[0] / an, s * ax[1] / an, s * ax[2] / an};
            const double* a = &quat[4 * l];
            double* o = &q0[((size_t)l * mLR + m) * 4];
            o[0] = a[0] * d[0] - a[1] * d[1] - a[2] * d[2] - a[3] * d[3];
            o[1] = a[0] * d[1] + a[1] * d[0] + a[2] * d[3] - a[3] * d[2];
            o[2] = a[0] * d[2] - a[1] * d[3] + a[2] * d[0] + a[3] * d[1];
            o[3] = a[0] * d[3] + a[1] * d[2] - a[2] * d[1] + a[3] * d[0];
        }
        for (int m = 0; m < mLT; m++) {
            t0[((size_t)l * mLT + m) * 2] = shift[2 * l] + 0.5 * G(g);
            t0[((size_t)l * mLT + m) * 2 + 1] = shift[2 * l + 1] + 0.5 * G(g);
        }
    }
    double *q0D = dev_upload(q0), *t0D = dev_upload(t0);
    std::vector<int> gid(n, 1);

    // ---- the iteration driver ----
    thx_refine_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.N = N; cfg.pf = pf; cfg.nImg = n;
    cfg.halfOfRank = world > 1 ? rank % 2 : -1;
    cfg.nHalfA = (n + 1) / 2;
    cfg.mLR = mLR; cfg.mLT = mLT; cfg.nPhase = nPhase; cfg.maxPhase = 0; cfg.mReco = mReco; cfg.batch = 100;
    cfg.rL = 1; cfg.nGroup = 1; cfg.groupSig = 1; cfg.pixelOrder = 1; cfg.wgPerCU = -1;
    cfg.pixelSize = pixelSize; cfg.maskRadiusPx = 0.45f * N; cfg.sigma2Init = (float)sigma2;
    cfg.transS = 2.0; cfg.transQ = 0.05; cfg.pfL = 2.0; cfg.pfS = 0.5; cfg.peakFactorR = 1e-3;
    cfg.seed = 12345;   // ONE seed for the job (checked over `world`): the images are told apart by their index over all ranks
    thx_refine* h = nullptr;
    CK(thx_refine_create(&h, &cfg, nullptr /* one rank per half: nothing to reduce */, wcomm));
    CK(thx_refine_set_particles(h, imgD, attrD, gid.data(), q0D, t0D, nullptr));
    CK(thx_refine_set_reference(h, refD, nullptr));
    CK(thx_refine_reset(h, nullptr));
    std::vector<float> fsc(N / 2);
    for (int it = 0; it < 2; it++) CK(thx_refine_iterate(h, fsc.data(), 1, nullptr));
    thx_refine_stats st;
    CK(thx_refine_get_stats(h, &st, 0));
    if (st.iterations != 2 || st.expectLaunches <= 0 || st.insertLaunches <= 0 || st.balancingRounds <= 0) {
        fprintf(stderr, "rank %d: implausible driver statistics\n", rank);
        exit(3);
    }
    // ---- half map of this rank's (first) half against the generating map ----
    float* mapD = dev_alloc<float>((size_t)N * N * N);
    CK(thx_refine_get_map(h, world > 1 ? rank % 2 : 0, mapD, nullptr));
    float *ftA = dev_alloc<float>((size_t)N * N * nc * 2), *ftB = dev_alloc<float>((size_t)N * N * nc * 2), *fscD = dev_alloc<float>(N / 2);
    CK(thx_fft3d_fw_dev(mapD, ftA, N, nullptr));
    CK(thx_fft3d_fw_dev(refD, ftB, N, nullptr));
    CK(thx_fsc_dev(fscD, N / 2, ftA, ftB, N, nullptr));
    CK(thx_device_sync());
    std::vector<float> fscT = dev_download(fscD, N / 2);
    Result r;
    for (int s = 0; s < 4; s++) { r.fscHalf[s] = fsc[1 + s]; r.fscTruth[s] = fscT[1 + s]; }
    CK(thx_refine_destroy(h));
    CK(thx_reco_destroy(plan));
    if (wcomm) CK(thx_comm_destroy(wcomm));
    for (void* p : {(void*)refD, (void*)vol, (void*)iColD, (void*)iRowD, (void*)quatD, (void*)rotD, (void*)slD, (void*)attrD, (void*)ctfD,
                    (void*)shiftD, (void*)rampD, (void*)imgD, (void*)q0D, (void*)t0D, (void*)mapD, (void*)ftA, (void*)ftB, (void*)fscD})
        CK(thx_free_dev(p));
    return r;
}

static bool check(const Result& r, int rank)
{
    bool ok = true;
    // shells 1-3 of a 32^3 box: against the generating map, and half against half (240 particles per half: the third shell of the
    // independent halves sits around 0.8)
    for (int s = 0; s < 3; s++) ok = ok && r.fscHalf[s] > (s < 2 ? 0.8f : 0.6f) && r.fscTruth[s] > 0.8f;
    printf("rank %d  half-map FSC shells 1-4: %.3f %.3f %.3f %.3f   vs generating map: %.3f %.3f %.3f %.3f  %s\n", rank, r.fscHalf[0],
           r.fscHalf[1], r.fscHalf[2], r.fscHalf[3], r.fscTruth[0], r.fscTruth[1], r.fscTruth[2], r.fscTruth[3], ok ? "" : "<-- LOW");
    return ok;
}

int main()
{
    // the parent never touches the GPU runtime (a HIP context does not survive fork()): a short-lived child counts the devices
    const char* tp = getenv("THX_COMM_TRANSPORT");
    g_oneDevice = tp && strcmp(tp, "shm") == 0;
    int nDev = 0;
    {
        const pid_t c = fork();
        if (c == 0) {
            int n = 0;
            if (thx_device_count(&n) != 0) _exit(0);
            _exit(n > 100 ? 100 : n);
        }
        int stx = 0;
        waitpid(c, &stx, 0);
        nDev = WIFEXITED(stx) ? WEXITSTATUS(stx) : 0;
    }
    if (nDev < 1) { fprintf(stderr, "no GPU visible\n"); return 1; }
    if (nDev < 2 && !g_oneDevice) {
        const Result r = run_rank(0, 1, nullptr);
        if (!check(r, 0)) return 1;
        printf("OK (one rank, both half-sets on one GPU)\n");
        return 0;
    }
    // two ranks, one GPU and one half-set each.  The parent never touches the GPU runtime; rank 0 creates the unique id
    // (its process hosts RCCL's bootstrap listener, so it must be a participant) and hands it to rank 1 through a pipe.
    int pfd[2];
    if (pipe(pfd)) return 1;
    pid_t kids[2];
    for (int rank = 0; rank < 2; rank++) {
        kids[rank] = fork();
        if (kids[rank] == 0) {
            unsigned char uid[128];
            if (rank == 0) {
                CK(thx_set_device(0));
                CK(thx_comm_unique_id(uid));
                if (write(pfd[1], uid, 128) != 128) _exit(1);
            } else {
                if (read(pfd[0], uid, 128) != 128) _exit(1);
            }
            const Result r = run_rank(rank, 2, uid);
            _exit(check(r, rank) ? 0 : 1);
        }
    }
    bool ok = true;
    for (int rank = 0; rank < 2; rank++) {
        int stx = 0;
        waitpid(kids[rank], &stx, 0);
        ok = ok && WIFEXITED(stx) && WEXITSTATUS(stx) == 0;
    }
    if (!ok) return 1;
    printf(g_oneDevice ? "OK (two ranks on one GPU over the shared-memory transport)\n" : "OK (two ranks over RCCL)\n");
    return 0;
}
