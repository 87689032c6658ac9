// thx_classify.hip -- one 3-D classification iteration (K references) sequenced in native code: host side C++, device work
// through the C ABI of this library only.  Reference control flow, restricted to the path in scope (BASELINE configs[3]):
//   Optimiser::expectation, global search  src/Optimiser.cpp:631-1140
//       scan of every image against K classes x nR rotations x nT shifts at r = rScan                       :756-894
//       class of the image: keepHalfHeightPeak(PAR_C) / resample(k, PAR_C) / rand(cls)                      :925-952
//       support points of the local search from the scan posterior of that class, minimum spread            :953-1079
//   Optimiser::expectation, local phases against the assigned reference (HOT LOOP B)                         :1141-1660
//   Optimiser::reconstructRef: mReco draws per image routed to the image's class (HOT LOOP C)               :7038-7241
//       prepareTF, then per class reconstruct with MAP off and MAP on (Reconstructor::reconstruct)          :7248-7760
//   Model::refreshProj for every class (src/Model.cpp:1013-1044) when cfg.refresh != 0
// One thx_classify handle = one rank's HBM-resident shard of images (one process per GPU), all of ONE half-set; the only
// exchange between the ranks of the half is the reduction of the K pairs of fixed-point accumulators (thx_reco_allreduce_acc,
// RCCL over xGMI).  The handle BORROWS the caller's rows on the rL = 0 pixel list and owns everything else.
// thunder_amd/native.py:NativeClassify is the front-end; bench.py --classification times this driver.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "thx_common.h"

using namespace thx;

struct thx_comm;
extern "C" {
size_t thx_reco_allreduce_acc_workspace(int dim, int maxRadius, int pf);
int thx_reco_allreduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, double* O, int* counter, int dim, int maxRadius, int pf,
                                 void* workspace, void* stream);
int thx_pixel_list_host(int N, int rU, int rL, int order, int* iCol, int* iRow, int* iPxl, int* iSig, int* nPxl);
int thx_draw_reco_dev(double* recoRot, double* recoTran, const double* r, const double* t, int nImg, int nR, int nT, int mReco,
                      unsigned long long seed, unsigned call, unsigned img0, void* stream);
}

namespace {

// rows of a sub-list cut from the rows of the rL = 0 list: dst [nImg][nSub] <- src [nImg][nSrc] through idx [nSub]
template <typename T>
__global__ void k_cut_rows(T* __restrict__ dst, const T* __restrict__ src, const int* __restrict__ idx, int nSub, int nSrc)
{
    const int l = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nSub) return;
    dst[(size_t)l * nSub + p] = src[(size_t)l * nSrc + idx[p]];
}

__global__ void k_fill_f64(double* __restrict__ p, double v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_fill_nan(float* __restrict__ p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __int_as_float(0x7fc00000);
}

// cls [nImg] -> clsD [nImg][mReco]: every draw of an image goes to the image's class (src/Optimiser.cpp:7129-7150)
__global__ void k_expand_cls(int* __restrict__ clsD, const int* __restrict__ cls, int nImg, int mReco)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (size_t)nImg * mReco) clsD[e] = cls[e / mReco];
}

// images per class (Model's class distribution of the iteration); one thread per image
__global__ void k_count_cls(int* __restrict__ count, const int* __restrict__ cls, int nImg, int nK)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < nImg && cls[l] >= 0 && cls[l] < nK) atomicAdd(count + cls[l], 1);
}

enum { CS_SCAN = 0, CS_SELECT, CS_LOCAL, CS_INSERT, CS_RECO, CS_COUNT };
enum { CE_SCAN = 16, CE_LOCAL, CE_INSERT };

}  // namespace

struct thx_classify {
    thx_classify_config cfg;
    thx_comm* hemi = nullptr;
    int N, pf, P, rU, nK, nImg, batch;
    int nPxlS = 0, nPxlE = 0, nPxlM = 0;
    std::vector<void*> owned;
    int *iColS, *iRowS, *iColE, *iRowE, *iColM, *iRowM, *e2m, *s2m;
    // borrowed rows on the rL = 0 list
    const float *datM = nullptr, *ctfM = nullptr, *sigM = nullptr, *w = nullptr;
    float *datE, *ctfE, *sigE, *datS, *ctfS, *sigS, *bounds;
    // scan grid
    double *gridR = nullptr, *gridT = nullptr, *mats = nullptr;
    float* traS = nullptr;
    bool haveGrid = false, haveRefs = false;
    // references and accumulators
    float *vols, *cells, *F, *T, *maps, *mapsX, *rotP;
    void* accInt;
    int* gexp;
    thx_reco* plan = nullptr;
    // scan posteriors, filter state
    double *pR, *pT, *r, *t, *wR, *wT, *k123, *s01, *topR, *topT, *rotB, *recoRot, *recoTran, *pD;
    float *uC, *uR, *uT, *baseL, *lwC, *luR, *luT, *lwD, *lbase;
    int *cls, *clsD, *clsCount;
    void *wsGlobal, *wsLocal, *wsReduce;
    unsigned pfCall = 0;
    long nImgHemi = 0;
    std::vector<float> fscMAP;
    thx_classify_capture cap = {};
    // timing
    bool timed = false;
    struct Ev { hipEvent_t a, b; int kind; int images; };
    std::vector<Ev> events;
    double stageMs[CS_COUNT] = {0, 0, 0, 0, 0};
    double scanMs = 0, localMs = 0, insertMs = 0;
    long scanLaunches = 0, localLaunches = 0, localImages = 0, insertLaunches = 0, insertImages = 0, rounds = 0, iterations = 0;
    int lastRounds[2 * 16];
    int clsCountHost[16];
};

namespace {

template <typename T>
int dalloc(thx_classify* h, T** p, size_t n)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed in the classification driver: %s", n * sizeof(T), hipGetErrorString(e));
        return (int)e;
    }
    h->owned.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}

int upload(thx_classify* h, int** p, const std::vector<int>& v)
{
    THX_RC(dalloc(h, p, v.size()));
    THX_CHECK(hipMemcpy(*p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

struct Scope {
    thx_classify* h; hipStream_t st; int kind, images; hipEvent_t a{}, b{}; bool on;
    Scope(thx_classify* h_, hipStream_t st_, int kind_, int images_ = 0) : h(h_), st(st_), kind(kind_), images(images_), on(h_->timed)
    {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    ~Scope()
    {
        if (on) { (void)hipEventRecord(b, st); h->events.push_back({a, b, kind, images}); }
    }
};

int resolve_events(thx_classify* h)
{
    for (auto& e : h->events) {
        THX_CHECK(hipEventSynchronize(e.b));
        float ms = 0;
        THX_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
        if (e.kind == CE_SCAN) { h->scanMs += ms; h->scanLaunches++; }
        else if (e.kind == CE_LOCAL) { h->localMs += ms; h->localLaunches++; h->localImages += e.images; }
        else if (e.kind == CE_INSERT) { h->insertMs += ms; h->insertLaunches++; h->insertImages += e.images; }
        else h->stageMs[e.kind] += ms;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    h->events.clear();
    return 0;
}

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

struct HostList {
    std::vector<int> iCol, iRow;
    int n = 0;
};

int host_list(HostList& pl, int N, int rU, int rL, int order)
{
    int n = 0;
    THX_RC(thx_pixel_list_host(N, rU, rL, order, nullptr, nullptr, nullptr, nullptr, &n));
    pl.iCol.resize(n); pl.iRow.resize(n); pl.n = n;
    std::vector<int> iPxl(n), iSig(n);
    THX_RC(thx_pixel_list_host(N, rU, rL, order, pl.iCol.data(), pl.iRow.data(), iPxl.data(), iSig.data(), &n));
    return 0;
}

// position of every pixel of `sub` in `full` (both lists of (iCol, iRow) pairs; sub is a subset of full)
int sub_index(std::vector<int>& idx, const HostList& sub, const HostList& full, int N)
{
    std::vector<int> pos((size_t)(N / 2 + 1) * N, -1);
    for (int k = 0; k < full.n; k++) pos[(size_t)(full.iRow[k] + N / 2) * (N / 2 + 1) + full.iCol[k]] = k;
    idx.resize(sub.n);
    for (int k = 0; k < sub.n; k++) {
        const int p = pos[(size_t)(sub.iRow[k] + N / 2) * (N / 2 + 1) + sub.iCol[k]];
        THX_REQUIRE(p >= 0, "a scan / local-search pixel is missing from the rL = 0 list");
        idx[k] = p;
    }
    return 0;
}

template <typename T>
int cut_rows(T* dst, const T* src, const int* idx, int nSub, int nSrc, int nImg, hipStream_t st)
{
    for (int l0 = 0; l0 < nImg; l0 += 65535) {
        const int nl = std::min(65535, nImg - l0);
        hipLaunchKernelGGL(k_cut_rows<T>, dim3((nSub + 255) / 256, nl), dim3(256), 0, st, dst + (size_t)l0 * nSub,
                           src + (size_t)l0 * nSrc, idx, nSub, nSrc);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

// ---- Optimiser::expectation, global search: scan, class, support points (src/Optimiser.cpp:756-1079) ----
int scan_and_select(thx_classify* h, hipStream_t st)
{
    const thx_classify_config& c = h->cfg;
    const size_t volStride = (size_t)h->P * h->P * (h->P / 2 + 1) * 2;
    {
        Scope s(h, st, CS_SCAN);
        THX_CHECK(hipMemsetAsync(h->uC, 0, (size_t)h->nImg * h->nK * sizeof(float), st));
        THX_CHECK(hipMemsetAsync(h->uR, 0, (size_t)h->nK * h->nImg * c.nR * sizeof(float), st));
        THX_CHECK(hipMemsetAsync(h->uT, 0, (size_t)h->nK * h->nImg * c.nT * sizeof(float), st));
        hipLaunchKernelGGL(k_fill_nan, dim3(blocks_for(h->nImg)), dim3(256), 0, st, h->baseL, (size_t)h->nImg);   // "unset", :737-745
        THX_LAUNCH_CHECK();
        for (int k = 0; k < h->nK; k++) {
            // the class's slices at every scanned rotation (:775-781), then the contraction against every image and shift
            THX_RC(thx_project_dev(h->vols + (size_t)k * volStride, h->rotP, h->mats, h->iColS, h->iRowS, c.nR, h->pf, h->P, h->nPxlS, st));
            Scope e(h, st, CE_SCAN, h->nImg);
            THX_RC(thx_expect_global_dev(h->rotP, h->traS, h->datS, h->ctfS, h->sigS, h->pR, h->pT, h->uC, h->uR, h->uT, h->baseL, k,
                                         h->nK, c.nR, c.nT, h->nPxlS, h->nImg, h->wsGlobal, st));
        }
    }
    {
        Scope s(h, st, CS_SELECT);
        h->pfCall++;
        THX_RC(thx_pf_class_select_dev(h->cls, h->uC, nullptr, h->nImg, h->nK, c.peakFactorC, c.seed, h->pfCall, st));
        h->pfCall++;
        THX_RC(thx_pf_scan_support_dev(h->r, h->t, h->wR, h->wT, h->k123, h->s01, h->topR, h->topT, h->gridR, h->gridT, h->uR, h->uT,
                                       h->cls, h->nImg, c.nR, c.nT, c.mLR, c.mLT, c.peakFactorR, c.scanMinK, c.scanMinS, c.seed,
                                       h->pfCall, st));
        // optional trace for the stage-level parity test: the support points as the scan left them
        if (h->cap.r0) THX_CHECK(hipMemcpyAsync(h->cap.r0, h->r, (size_t)h->nImg * c.mLR * 4 * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (h->cap.t0) THX_CHECK(hipMemcpyAsync(h->cap.t0, h->t, (size_t)h->nImg * c.mLT * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

// ---- HOT LOOP B against the assigned reference (src/Optimiser.cpp:1141-1660) ----
int local_phases(thx_classify* h, hipStream_t st)
{
    const thx_classify_config& c = h->cfg;
    Scope s(h, st, CS_LOCAL);
    for (int p = 0; p < c.nPhase; p++)
        for (int b0 = 0; b0 < h->nImg; b0 += h->batch) {
            const int nb = std::min(h->batch, h->nImg - b0);
            double* r = h->r + (size_t)b0 * c.mLR * 4;
            double* t = h->t + (size_t)b0 * c.mLT * 2;
            double* wR = h->wR + (size_t)b0 * c.mLR;
            double* wT = h->wT + (size_t)b0 * c.mLT;
            double* k = h->k123 + (size_t)b0 * 3;
            double* s01 = h->s01 + (size_t)b0 * 2;
            const double f = p == 0 ? c.pfL : c.pfS;
            h->pfCall++;
            THX_RC(thx_pf_perturb_dev(r, t, wR, wT, k, s01, nb, c.mLR, c.mLT, f, f, c.transS, c.transQ, c.seed, h->pfCall, nullptr, st));
            THX_RC(thx_rotmat_dev(r, h->rotB, nb * c.mLR, st));
            {
                Scope e(h, st, CE_LOCAL, nb);
                THX_RC(thx_expect_local_packed_dev(h->cells, h->cls + b0, h->P, h->pf, h->N, h->iColE, h->iRowE, h->nPxlE, nb,
                                                   h->datE + (size_t)b0 * h->nPxlE * 2, h->ctfE + (size_t)b0 * h->nPxlE,
                                                   h->sigE + (size_t)b0 * h->nPxlE, h->rotB, c.mLR, t, c.mLT, 1, nullptr, wR, wT, h->pD,
                                                   h->lwC, h->luR, h->luT, h->lwD, h->lbase, nullptr, h->wsLocal, c.wgPerCU, nullptr, st));
            }
            h->pfCall++;
            THX_RC(thx_pf_update_dev(r, t, wR, wT, h->luR, h->luT, k, s01, h->topR + (size_t)b0 * 4, h->topT + (size_t)b0 * 2, nb, c.mLR,
                                     c.mLT, c.peakFactorR, c.seed, h->pfCall, nullptr, st));
        }
    return 0;
}

// ---- HOT LOOP C: the draws of every image into the F / T of its class, one insertion session over all batches ----
int insertion(thx_classify* h, hipStream_t st)
{
    const thx_classify_config& c = h->cfg;
    const size_t volN = (size_t)h->P * h->P * (h->P / 2 + 1);
    Scope s(h, st, CS_INSERT);
    THX_CHECK(hipMemsetAsync(h->F, 0, (size_t)h->nK * volN * 2 * sizeof(float), st));
    THX_CHECK(hipMemsetAsync(h->T, 0, (size_t)h->nK * volN * sizeof(float), st));
    THX_RC(thx_insert_scale_dev(h->gexp, h->bounds, h->w, h->nImg, c.mReco, 0, h->nImgHemi, h->hemi, st));
    THX_CHECK(hipMemsetAsync(h->accInt, 0, thx_insert_acc_bytes(h->P, h->nK), st));
    for (int b0 = 0; b0 < h->nImg; b0 += h->batch) {
        const int nb = std::min(h->batch, h->nImg - b0);
        h->pfCall++;
        THX_RC(thx_draw_reco_dev(h->recoRot, h->recoTran, h->r + (size_t)b0 * c.mLR * 4, h->t + (size_t)b0 * c.mLT * 2, nb, c.mLR, c.mLT,
                                 c.mReco, c.seed, h->pfCall, (unsigned)b0, st));
        hipLaunchKernelGGL(k_expand_cls, dim3(blocks_for((size_t)nb * c.mReco)), dim3(256), 0, st, h->clsD, h->cls + b0, nb, c.mReco);
        THX_LAUNCH_CHECK();
        Scope e(h, st, CE_INSERT, nb);
        THX_RC(thx_insert_accumulate_dev(h->accInt, h->gexp, h->bounds + (size_t)b0 * 2, nullptr, nullptr, h->P, h->nK,
                                         h->datM + (size_t)b0 * h->nPxlM * 2, h->ctfM + (size_t)b0 * h->nPxlM, h->w + b0, h->recoRot,
                                         h->recoTran, nullptr, h->clsD, nullptr, nullptr, 0, c.pixelSize, h->iColM, h->iRowM, h->pf,
                                         h->nPxlM, c.mReco, h->N, nb, st));
    }
    // the half-set reduce on the integers, class by class through one workspace (N ranks == 1 rank, bit for bit)
    for (int k = 0; k < h->nK && h->hemi; k++)
        THX_RC(thx_reco_allreduce_acc_class(h->hemi, h->accInt, h->nK, k, nullptr, nullptr, h->P, h->rU, h->pf, h->wsReduce, st));
    THX_RC(thx_insert_finish_dev(h->F, h->T, h->accInt, h->gexp, h->P, h->nK, st));
    if (h->cap.Fraw) THX_CHECK(hipMemcpyAsync(h->cap.Fraw, h->F, (size_t)h->nK * volN * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (h->cap.Traw) THX_CHECK(hipMemcpyAsync(h->cap.Traw, h->T, (size_t)h->nK * volN * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

// ---- Optimiser::reconstructRef after the insertion: prepareTF's normalisation, MAP off then MAP on per class ----
int reconstruct_classes(thx_classify* h, hipStream_t st)
{
    const thx_classify_config& c = h->cfg;
    const size_t volN = (size_t)h->P * h->P * (h->P / 2 + 1), mapN = (size_t)h->N * h->N * h->N;
    const size_t cellStride = thx_projector_packed_bytes(h->P) / sizeof(float);
    Scope s(h, st, CS_RECO);
    THX_CHECK(hipMemsetAsync(h->clsCount, 0, 16 * sizeof(int), st));
    hipLaunchKernelGGL(k_count_cls, dim3(blocks_for(h->nImg)), dim3(256), 0, st, h->clsCount, h->cls, h->nImg, h->nK);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpyAsync(h->clsCountHost, h->clsCount, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
    // T(0,0,0) of every class after the half-set reduce: the sum of w ctf(0)^2 over every draw any rank of the half sent there
    float t0[16];
    THX_CHECK(hipMemcpy2DAsync(t0, sizeof(float), h->T, volN * sizeof(float), sizeof(float), (size_t)h->nK, hipMemcpyDeviceToHost, st));
    THX_CHECK(hipStreamSynchronize(st));
    for (int k = 0; k < h->nK; k++) {
        h->lastRounds[2 * k] = h->lastRounds[2 * k + 1] = 0;
        if (!(t0[k] > 0.f)) continue;   // a class no image of the half went to keeps its reference (sf = 1 / T(0,0,0) has nothing to normalise)
        float* F = h->F + (size_t)k * volN * 2;
        float* T = h->T + (size_t)k * volN;
        THX_RC(thx_normalise_tf_dev(F, T, h->P, st));
        int iters = 0;
        float diffC = 0;
        // setMAP(false) / setMAP(true) with the FSC the caller handed over (all ones before the first comparison of half maps);
        // joinHalf off: one half-set per handle, the comparison of the two halves is the caller's
        THX_RC(thx_reco_reconstruct_dev(h->plan, F, T, h->rU, nullptr, 0, 0, 0, 1, h->maps + (size_t)k * mapN, &iters, &diffC, st));
        h->rounds += iters; h->lastRounds[2 * k] = iters;
        THX_RC(thx_reco_reconstruct_dev(h->plan, F, T, h->rU, h->fscMAP.data() + (size_t)k * h->rU, h->rU, 0, 1, 1,
                                        h->mapsX + (size_t)k * mapN, &iters, &diffC, st));
        h->rounds += iters; h->lastRounds[2 * k + 1] = iters;
        if (c.refresh) {   // Model::refreshProj: the MAP-on map is the class's next reference
            THX_RC(thx_reco_set_projectee_dev(h->plan, h->mapsX + (size_t)k * mapN, h->vols + (size_t)k * volN * 2, st));
            THX_RC(thx_projector_pack_dev(h->cells + (size_t)k * cellStride, h->vols + (size_t)k * volN * 2, h->P, 1, st));
        }
    }
    return 0;
}

}  // namespace

extern "C" {

int thx_classify_destroy(thx_classify* h)
{
    if (!h) return 0;
    for (auto& e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (h->plan) thx_reco_destroy(h->plan);
    for (void* p : h->owned) (void)hipFree(p);
    delete h;
    return 0;
}

int thx_classify_create(thx_classify** out, const thx_classify_config* cfg, thx_comm* hemi)
{
    THX_REQUIRE(out && cfg, "null argument");
    const thx_classify_config& c = *cfg;
    THX_REQUIRE(c.N > 0 && c.N % 2 == 0 && c.pf >= 1, "bad box / padding factor");
    THX_REQUIRE(c.nK >= 1 && c.nK <= 16, "1 <= nK <= 16 classes");
    THX_REQUIRE(c.nImg > 0 && c.nR > 0 && c.nT > 0 && c.nR <= 16384 && c.nT <= 16384, "bad image count / scan grid (nR, nT <= 16384)");
    THX_REQUIRE(c.mLR > 0 && c.mLR <= 256 && c.mLT > 0 && c.mLT <= 32, "mLR <= 256, mLT <= 32 support points");
    THX_REQUIRE(c.nPhase >= 1 && c.mReco > 0 && c.mReco < 4096, "nPhase >= 1, 0 < mReco < 4096");
    THX_REQUIRE(c.rL >= 0 && c.rScan > c.rL && c.rScan <= c.N / 2 - 2, "rL < rScan <= N / 2 - 2");
    thx_classify* h = new thx_classify();
    h->cfg = c;
    h->hemi = hemi;
    h->N = c.N; h->pf = c.pf; h->P = c.N * c.pf; h->rU = c.N / 2 - 2; h->nK = c.nK; h->nImg = c.nImg;
    h->batch = c.batch > 0 ? std::min(c.batch, c.nImg) : c.nImg;
    h->nImgHemi = c.nImgHemi > 0 ? c.nImgHemi : c.nImg;
    h->fscMAP.assign((size_t)c.nK * h->rU, 1.0f);
    memset(h->lastRounds, 0, sizeof(h->lastRounds));
    memset(h->clsCountHost, 0, sizeof(h->clsCountHost));
    int rc = [&]() -> int {
        HostList S, E, M;
        THX_RC(host_list(S, c.N, c.rScan, c.rL, c.pixelOrder));
        THX_RC(host_list(E, c.N, h->rU, c.rL, c.pixelOrder));
        THX_RC(host_list(M, c.N, h->rU, 0, 0));
        h->nPxlS = S.n; h->nPxlE = E.n; h->nPxlM = M.n;
        std::vector<int> e2m, s2m;
        THX_RC(sub_index(e2m, E, M, c.N));
        THX_RC(sub_index(s2m, S, M, c.N));
        THX_RC(upload(h, &h->iColS, S.iCol)); THX_RC(upload(h, &h->iRowS, S.iRow));
        THX_RC(upload(h, &h->iColE, E.iCol)); THX_RC(upload(h, &h->iRowE, E.iRow));
        THX_RC(upload(h, &h->iColM, M.iCol)); THX_RC(upload(h, &h->iRowM, M.iRow));
        THX_RC(upload(h, &h->e2m, e2m)); THX_RC(upload(h, &h->s2m, s2m));
        const size_t n = c.nImg, volN = (size_t)h->P * h->P * (h->P / 2 + 1), mapN = (size_t)c.N * c.N * c.N;
        THX_RC(dalloc(h, &h->datE, n * E.n * 2)); THX_RC(dalloc(h, &h->ctfE, n * E.n)); THX_RC(dalloc(h, &h->sigE, n * E.n));
        THX_RC(dalloc(h, &h->datS, n * S.n * 2)); THX_RC(dalloc(h, &h->ctfS, n * S.n)); THX_RC(dalloc(h, &h->sigS, n * S.n));
        THX_RC(dalloc(h, &h->bounds, n * 2));
        THX_RC(dalloc(h, &h->gridR, (size_t)c.nR * 4)); THX_RC(dalloc(h, &h->gridT, (size_t)c.nT * 2));
        THX_RC(dalloc(h, &h->mats, (size_t)c.nR * 9)); THX_RC(dalloc(h, &h->traS, (size_t)c.nT * S.n * 2));
        THX_RC(dalloc(h, &h->rotP, (size_t)c.nR * S.n * 2));
        THX_RC(dalloc(h, &h->vols, (size_t)c.nK * volN * 2));
        {
            float* cells = nullptr;
            void* q = nullptr;
            hipError_t e = hipMalloc(&q, (size_t)c.nK * thx_projector_packed_bytes(h->P));
            if (e != hipSuccess) { set_error("hipMalloc of %d cell-packed references failed: %s", c.nK, hipGetErrorString(e)); return (int)e; }
            h->owned.push_back(q);
            cells = reinterpret_cast<float*>(q);
            h->cells = cells;
        }
        THX_RC(dalloc(h, &h->F, (size_t)c.nK * volN * 2)); THX_RC(dalloc(h, &h->T, (size_t)c.nK * volN));
        THX_RC(dalloc(h, &h->maps, (size_t)c.nK * mapN)); THX_RC(dalloc(h, &h->mapsX, (size_t)c.nK * mapN));
        THX_CHECK(hipMemset(h->maps, 0, (size_t)c.nK * mapN * sizeof(float)));    // (the maps of a class no image went to stay zero)
        THX_CHECK(hipMemset(h->mapsX, 0, (size_t)c.nK * mapN * sizeof(float)));
        {
            void* q = nullptr;
            hipError_t e = hipMalloc(&q, thx_insert_acc_bytes(h->P, c.nK));
            if (e != hipSuccess) { set_error("hipMalloc of the fixed-point accumulators failed: %s", hipGetErrorString(e)); return (int)e; }
            h->owned.push_back(q);
            h->accInt = q;
        }
        THX_RC(dalloc(h, &h->gexp, 2));
        THX_RC(dalloc(h, &h->pR, n * c.nR)); THX_RC(dalloc(h, &h->pT, n * c.nT));
        THX_RC(dalloc(h, &h->uC, n * c.nK)); THX_RC(dalloc(h, &h->uR, (size_t)c.nK * n * c.nR)); THX_RC(dalloc(h, &h->uT, (size_t)c.nK * n * c.nT));
        THX_RC(dalloc(h, &h->baseL, n));
        THX_RC(dalloc(h, &h->cls, n)); THX_RC(dalloc(h, &h->clsD, (size_t)h->batch * c.mReco)); THX_RC(dalloc(h, &h->clsCount, 16));
        THX_RC(dalloc(h, &h->r, n * c.mLR * 4)); THX_RC(dalloc(h, &h->t, n * c.mLT * 2));
        THX_RC(dalloc(h, &h->wR, n * c.mLR)); THX_RC(dalloc(h, &h->wT, n * c.mLT));
        THX_RC(dalloc(h, &h->k123, n * 3)); THX_RC(dalloc(h, &h->s01, n * 2));
        THX_RC(dalloc(h, &h->topR, n * 4)); THX_RC(dalloc(h, &h->topT, n * 2));
        const size_t nb = h->batch;
        THX_RC(dalloc(h, &h->rotB, nb * c.mLR * 9));
        THX_RC(dalloc(h, &h->recoRot, nb * c.mReco * 9)); THX_RC(dalloc(h, &h->recoTran, nb * c.mReco * 2));
        THX_RC(dalloc(h, &h->pD, nb));
        THX_RC(dalloc(h, &h->lwC, nb)); THX_RC(dalloc(h, &h->luR, nb * c.mLR)); THX_RC(dalloc(h, &h->luT, nb * c.mLT));
        THX_RC(dalloc(h, &h->lwD, nb)); THX_RC(dalloc(h, &h->lbase, nb));
        {
            unsigned char* q = nullptr;
            THX_RC(dalloc(h, &q, thx_expect_global_workspace(c.nImg, c.nR, c.nT))); h->wsGlobal = q;
            THX_RC(dalloc(h, &q, thx_expect_local_workspace(h->batch, c.mLR, c.mLT, 1))); h->wsLocal = q;
            THX_RC(dalloc(h, &q, hemi ? thx_reco_allreduce_acc_workspace(h->P, h->rU, h->pf) : 1)); h->wsReduce = q;
        }
        // uniform priors of the scanned grid (Particle::reset(nR, nT): every support point 1 / n), defocus prior 1
        hipLaunchKernelGGL(k_fill_f64, dim3(blocks_for(n * c.nR)), dim3(256), 0, 0, h->pR, 1.0 / c.nR, n * c.nR);
        hipLaunchKernelGGL(k_fill_f64, dim3(blocks_for(n * c.nT)), dim3(256), 0, 0, h->pT, 1.0 / c.nT, n * c.nT);
        hipLaunchKernelGGL(k_fill_f64, dim3(blocks_for(nb)), dim3(256), 0, 0, h->pD, 1.0, nb);
        THX_LAUNCH_CHECK();
        THX_CHECK(hipDeviceSynchronize());
        THX_RC(thx_reco_create(&h->plan, c.N, c.N, c.pf, 1.9f, 15.0f));
        return 0;
    }();
    if (rc) { thx_classify_destroy(h); return rc; }
    *out = h;
    return 0;
}

int thx_classify_set_grid(thx_classify* h, const double* quat, const double* shifts, void* stream)
{
    THX_REQUIRE(h && quat && shifts, "null argument");
    hipStream_t st = as_stream(stream);
    const thx_classify_config& c = h->cfg;
    THX_CHECK(hipMemcpyAsync(h->gridR, quat, (size_t)c.nR * 4 * sizeof(double), hipMemcpyDefault, st));
    THX_CHECK(hipMemcpyAsync(h->gridT, shifts, (size_t)c.nT * 2 * sizeof(double), hipMemcpyDefault, st));
    THX_RC(thx_rotmat_dev(h->gridR, h->mats, c.nR, st));
    THX_RC(thx_translate_dev(h->traS, h->gridT, c.nT, h->iColS, h->iRowS, h->nPxlS, h->N, st));
    h->haveGrid = true;
    return 0;
}

int thx_classify_set_particles(thx_classify* h, const float* datM, const float* ctfM, const float* sigRcpM, const float* w, void* stream)
{
    THX_REQUIRE(h && datM && ctfM && sigRcpM && w, "null argument");
    hipStream_t st = as_stream(stream);
    h->datM = datM; h->ctfM = ctfM; h->sigM = sigRcpM; h->w = w;
    typedef float2 cplx;
    THX_RC(cut_rows<cplx>((cplx*)h->datE, (const cplx*)datM, h->e2m, h->nPxlE, h->nPxlM, h->nImg, st));
    THX_RC(cut_rows<float>(h->ctfE, ctfM, h->e2m, h->nPxlE, h->nPxlM, h->nImg, st));
    THX_RC(cut_rows<float>(h->sigE, sigRcpM, h->e2m, h->nPxlE, h->nPxlM, h->nImg, st));
    THX_RC(cut_rows<cplx>((cplx*)h->datS, (const cplx*)datM, h->s2m, h->nPxlS, h->nPxlM, h->nImg, st));
    THX_RC(cut_rows<float>(h->ctfS, ctfM, h->s2m, h->nPxlS, h->nPxlM, h->nImg, st));
    THX_RC(cut_rows<float>(h->sigS, sigRcpM, h->s2m, h->nPxlS, h->nPxlM, h->nImg, st));
    THX_RC(thx_insert_bounds_dev(h->bounds, datM, ctfM, h->nPxlM, h->nImg, st));
    return 0;
}

int thx_classify_set_references(thx_classify* h, const float* refRL, void* stream)
{
    THX_REQUIRE(h && refRL, "null argument");
    hipStream_t st = as_stream(stream);
    const size_t volN = (size_t)h->P * h->P * (h->P / 2 + 1), mapN = (size_t)h->N * h->N * h->N;
    for (int k = 0; k < h->nK; k++)
        THX_RC(thx_reco_set_projectee_dev(h->plan, refRL + (size_t)k * mapN, h->vols + (size_t)k * volN * 2, st));
    THX_RC(thx_projector_pack_dev(h->cells, h->vols, h->P, h->nK, st));
    h->haveRefs = true;
    return 0;
}

int thx_classify_set_fsc(thx_classify* h, const float* fscHost, int n)
{
    THX_REQUIRE(h && fscHost && n == h->nK * h->rU, "fsc: [nK][N / 2 - 2] floats");
    h->fscMAP.assign(fscHost, fscHost + n);
    return 0;
}

int thx_classify_set_capture(thx_classify* h, const thx_classify_capture* capture)
{
    THX_REQUIRE(h, "null handle");
    if (capture) h->cap = *capture; else memset(&h->cap, 0, sizeof(h->cap));
    return 0;
}

int thx_classify_iterate(thx_classify* h, int timed, void* stream)
{
    THX_REQUIRE(h && h->datM && h->haveGrid && h->haveRefs, "thx_classify_set_grid / set_particles / set_references first");
    hipStream_t st = as_stream(stream);
    h->timed = timed != 0;
    if (h->events.size() > 4096) THX_RC(resolve_events(h));   // (a caller that never asks for the statistics must not pile events up)
    THX_RC(scan_and_select(h, st));
    THX_RC(local_phases(h, st));
    THX_RC(insertion(h, st));
    THX_RC(reconstruct_classes(h, st));
    h->iterations++;
    return 0;
}

int thx_classify_get_view(thx_classify* h, thx_classify_view* v)
{
    THX_REQUIRE(h && v, "null argument");
    memset(v, 0, sizeof(*v));
    v->nImg = h->nImg; v->nK = h->nK; v->nPxlS = h->nPxlS; v->nPxlE = h->nPxlE; v->nPxlM = h->nPxlM; v->vdim = h->P;
    v->cls = h->cls; v->uC = h->uC; v->uR = h->uR; v->uT = h->uT;
    v->r = h->r; v->t = h->t; v->wR = h->wR; v->wT = h->wT; v->topR = h->topR; v->topT = h->topT;
    v->vols = h->vols; v->cells = h->cells; v->F = h->F; v->T = h->T; v->maps = h->maps; v->mapsMAP = h->mapsX;
    return 0;
}

int thx_classify_get_stats(thx_classify* h, thx_classify_stats* out, int reset)
{
    THX_REQUIRE(h && out, "null argument");
    THX_RC(resolve_events(h));
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < CS_COUNT; i++) out->stageMs[i] = h->stageMs[i];
    out->scanMs = h->scanMs; out->localMs = h->localMs; out->insertMs = h->insertMs;
    out->scanLaunches = h->scanLaunches; out->localLaunches = h->localLaunches; out->localImages = h->localImages;
    out->insertLaunches = h->insertLaunches; out->insertImages = h->insertImages;
    out->balancingRounds = h->rounds; out->iterations = h->iterations;
    out->nPxlS = h->nPxlS; out->nPxlE = h->nPxlE; out->nPxlM = h->nPxlM; out->batch = h->batch;
    memcpy(out->lastRounds, h->lastRounds, sizeof(out->lastRounds));
    memcpy(out->classCount, h->clsCountHost, sizeof(out->classCount));
    if (reset) {
        for (int i = 0; i < CS_COUNT; i++) h->stageMs[i] = 0;
        h->scanMs = h->localMs = h->insertMs = 0;
        h->scanLaunches = h->localLaunches = h->localImages = h->insertLaunches = h->insertImages = h->rounds = h->iterations = 0;
    }
    return 0;
}

}  // extern "C"
