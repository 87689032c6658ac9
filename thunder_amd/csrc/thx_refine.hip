// thx_refine.hip -- the per-iteration E/M loop sequenced in native code (host side C++, device work through the C ABI of
// this library only): one driver for refinement and K-class classification, local / global / CTF search, any point group.
// Reference control flow, restricted to the path in scope:
//   Optimiser::expectation   src/Optimiser.cpp:631-1140 (global search: scan :756-894, class :925-952, support points :953-1079)
//                            src/Optimiser.cpp:1141-1660 (HOT LOOP B: local particle-filter phases)
//   Optimiser::maximization  src/Optimiser.cpp:3405-3480 -> allReduceSigma :6395-6710, reconstructRef :6711-7766
//                            (HOT LOOP C :7038-7241, prepareTF, reconstruct x 2 per half)
//   Model::compareTwoHemispheres FSC (src/Model.cpp:424-551, src/Functions/Spectrum.cpp:302-337), Model::refreshProj
//   (src/Model.cpp:1013-1044), Optimiser::reCentreImg / reMaskImg (:6065-6149), allocPreCalIdx / allocPreCal (:7991-8171)
// One thx_refine handle = one rank's HBM-resident shard of particles (one process per GPU).  The exchanges between ranks: the
// half-set reduction of the 64-bit F / T accumulators of every class towards the rank that reconstructs it
// (thx_reco_reduce_acc_class, RCCL over xGMI), three small sigma tables, the norm vector and the class histogram, and the N^3 half
// maps of every class from their reconstructing ranks -- all through thx_comm (thx_comm.hip).  The handle BORROWS the caller's image stack
// (_imgOri) and owns everything else.
#include <algorithm>
#include <cmath>
#include <vector>

#include "thx_common.h"
#include "thx_philox.h"

struct thx_comm;
extern "C" {
int thx_comm_rank(const thx_comm* c);
int thx_comm_size(const thx_comm* c);
int thx_comm_allreduce_f32(thx_comm* c, float* buf, size_t count, void* stream);
int thx_comm_allreduce_f64(thx_comm* c, double* buf, size_t count, void* stream);
int thx_comm_allreduce_i32(thx_comm* c, int* buf, size_t count, void* stream);
int thx_comm_allreduce_max_f64(thx_comm* c, double* buf, size_t count, void* stream);
int thx_reco_reconstruct_async_dev(thx_reco* r, const float* F, float* T, int maxRadius, const float* FSC_host, int nFSC, int joinHalf,
                                   int MAP, int gridCorr, float* dstRL, void* result, void* stream);
int thx_reco_reduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, int root, int dim, int maxRadius, int pf, void* workspace,
                              void* stream);
int thx_reco_allreduce_acc_class(thx_comm* hemi, void* acc, int nK, int k, double* O, int* counter, int dim, int maxRadius, int pf,
                                 void* workspace, void* stream);
int thx_comm_broadcast(thx_comm* c, void* buf, size_t bytes, int root, void* stream);
size_t thx_reco_allreduce_workspace(int dim, int maxRadius, int pf);
int thx_compare_hemispheres_dev(float* A, float* B, int N, int rU, float* fscHost, const float* maskRL, float coreR, float ew,
                                int avgFlag, int avgR, unsigned long long seed, unsigned call, int* randomPhaseThresOut,
                                void* stream);
int thx_soft_mask_volume_dev(float* vol, int N, float r, float ew, float bg, void* stream);
int thx_reco_allreduce(thx_comm* hemi, float* F, float* T, double* O, int* counter, int dim, int maxRadius, int pf,
                       void* workspace, void* stream);
int thx_insert_groups_total(unsigned long long* out, int reset, void* stream);
size_t thx_reco_allreduce_acc_workspace(int dim, int maxRadius, int pf);
int thx_reco_allreduce_acc(thx_comm* hemi, void* acc, double* O, int* counter, int dim, int maxRadius, int pf, void* workspace,
                           void* stream);
}

namespace thx {

// ---------------------------------------------------------------------------------------------
// host integer work: Optimiser::allocPreCalIdx (src/Optimiser.cpp:7991-8041) and the pixel-visit order of the E-step
// ---------------------------------------------------------------------------------------------
struct PixelList {
    std::vector<int> iCol, iRow, iPxl, iSig;
    int nPxl = 0;
};

static PixelList pixel_list_host(int N, int rU, int rL)
{
    PixelList pl;
    const float rU2 = pow2f_((float)rU), rL2 = pow2f_((float)rL);   // TSGSL_pow_2 on RFLOAT
    const int lim = rU + 1;
    for (int j = -lim; j < lim; j++)
        for (int i = 0; i <= lim; i++) {   // IMAGE_FOR_PIXEL_R_FT(rU + 1)
            if (i == 0 && j < 0) continue;
            const float u = (float)((double)i * i + (double)j * j);   // QUAD(i, j) narrowed to RFLOAT
            if (u < rU2 && u >= rL2) {
                const int v = (int)rint(gsl_hypot_((double)i, (double)j));   // AROUND(NORM(i, j))
                if (v < rU && v >= rL) {
                    pl.iPxl.push_back((j >= 0 ? j : j + N) * (N / 2 + 1) + i);
                    pl.iCol.push_back(i);
                    pl.iRow.push_back(j);
                    pl.iSig.push_back(v);
                }
            }
        }
    pl.nPxl = (int)pl.iCol.size();
    return pl;
}

static unsigned spread_bits(unsigned v)
{
    unsigned o = 0;
    for (int b = 0; b < 12; b++) o |= ((v >> b) & 1u) << (2 * b);
    return o;
}

// Morton (Z-order) visit order over (iCol, iRow + N): neighbouring pixels -- and the volume cells their rotations touch --
// stay close together in time (thunder_amd/refine.py:pixel_visit_order is the Python twin; tests compare the two)
static void morton_order(PixelList& pl, int N)
{
    std::vector<int> idx(pl.nPxl);
    for (int i = 0; i < pl.nPxl; i++) idx[i] = i;
    std::vector<unsigned> key(pl.nPxl);
    for (int i = 0; i < pl.nPxl; i++) key[i] = spread_bits((unsigned)pl.iCol[i]) | (spread_bits((unsigned)(pl.iRow[i] + N)) << 1);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] < key[b]; });
    PixelList o;
    o.nPxl = pl.nPxl;
    for (int i : idx) {
        o.iCol.push_back(pl.iCol[i]); o.iRow.push_back(pl.iRow[i]); o.iPxl.push_back(pl.iPxl[i]); o.iSig.push_back(pl.iSig[i]);
    }
    pl = o;
}

// ---------------------------------------------------------------------------------------------
// small device kernels of the driver
// ---------------------------------------------------------------------------------------------
// allocPreCal, src/Optimiser.cpp:8077-8081: _sigRcpP[l][p] = _sigRcp(groupID[l] - 1, iSig[p]);  grid (ceil(nPxl/256), nImg)
__global__ void k_sigrcp_rows(float* __restrict__ sigRcpP, const float* __restrict__ sigRcp, const int* __restrict__ gid0,
                              const int* __restrict__ iSig, int nPxl, int rSig)
{
    const int l = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPxl) return;
    sigRcpP[(size_t)l * nPxl + p] = sigRcp[(size_t)gid0[l] * rSig + iSig[p]];
}

// The mReco draws of the insertion (src/Optimiser.cpp:7129-7150): the filter has been resampled (thx_pf_update_dev), so
// Particle::rand(quat) / rand(tran) [/ rand(d) under CTF search] (src/Particle.cpp:2109-2178) is a uniform pick among the
// support points.  One thread per draw; rotation matrix as rotate3D (src/Geometry/Euler.cpp:181-189).
__global__ void k_draw_reco(double* __restrict__ recoRot, double* __restrict__ recoTran, double* __restrict__ recoD,
                            const double* __restrict__ r, const double* __restrict__ t, const double* __restrict__ d, int nImg, int nR,
                            int nT, int nD, int mReco, unsigned long long seed, unsigned call, unsigned img0)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)nImg * mReco) return;
    const int l = (int)(e / mReco), m = (int)(e - (size_t)l * mReco);
    double u[4];
    draw_u4(u, seed, img0 + (unsigned)l, call, 7u, (unsigned)m);
    int iR = (int)(u[0] * nR), iT = (int)(u[1] * nT);
    iR = iR >= nR ? nR - 1 : iR;
    iT = iT >= nT ? nT - 1 : iT;
    const double* q = r + ((size_t)l * nR + iR) * 4;
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const double A[3][3] = {{0, -q3, q2}, {q3, 0, -q1}, {-q2, q1, 0}};
    double* mat = recoRot + e * 9;
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[rr][k] * A[k][c];
            mat[c * 3 + rr] = (rr == c ? 1.0 : 0.0) + 2 * q0 * A[rr][c] + 2 * s;
        }
    recoTran[2 * e] = t[((size_t)l * nT + iT) * 2];
    recoTran[2 * e + 1] = t[((size_t)l * nT + iT) * 2 + 1];
    if (recoD && nD > 0) {
        int iD = (int)(u[2] * nD);
        iD = iD >= nD ? nD - 1 : iD;
        recoD[e] = d[(size_t)l * nD + iD];
    }
}

// reCentreImg bookkeeping, src/Optimiser.cpp:6065-6090: _offset[l] -= tran; _par[l].setT(t - tran); setTopT(topT - tran)
__global__ void k_recentre_state(double* __restrict__ offset, double* __restrict__ t, double* __restrict__ topT,
                                 const double* __restrict__ tranTop, int nImg, int nT)
{
    const int l = blockIdx.x;
    if (l >= nImg) return;
    const double tx = tranTop[2 * l], ty = tranTop[2 * l + 1];
    for (int i = threadIdx.x; i < nT; i += blockDim.x) {
        t[((size_t)l * nT + i) * 2] -= tx;
        t[((size_t)l * nT + i) * 2 + 1] -= ty;
    }
    if (threadIdx.x == 0) {
        offset[2 * l] -= tx; offset[2 * l + 1] -= ty;
        topT[2 * l] -= tx; topT[2 * l + 1] -= ty;
    }
}

// Particle::load -> calVari for the shifts (src/Particle.cpp:1291-1330): per-column standard deviation (n - 1 form)
__global__ void k_shift_sd(double* __restrict__ s01, const double* __restrict__ t, int nImg, int nT)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nImg) return;
    for (int c = 0; c < 2; c++) {
        double m = 0;
        for (int i = 0; i < nT; i++) m += t[((size_t)l * nT + i) * 2 + c];
        m /= nT;
        double v = 0;
        for (int i = 0; i < nT; i++) { const double d = t[((size_t)l * nT + i) * 2 + c] - m; v += d * d; }
        s01[2 * l + c] = sqrt(v / (nT > 1 ? nT - 1 : 1));
    }
}

template <typename T>
__global__ void k_fill(T* __restrict__ p, T v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// first support point of every image: topR [n][4] <- r [n][nR][4] (stride copy)
__global__ void k_take_first(double* __restrict__ dst, const double* __restrict__ src, int n, int stride, int width)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n * width) return;
    const size_t l = e / width, c = e - l * width;
    dst[e] = src[l * stride + c];
}

// dst [n] <- src [n * stride] (T(0,0,0) of the n classes of a half)
__global__ void k_take_first_f32(float* __restrict__ dst, const float* __restrict__ src, int n, size_t stride)
{
    const int k = threadIdx.x;
    if (k < n) dst[k] = src[(size_t)k * stride];
}

// T(0,0,0) of the classes this rank does not reconstruct are partial sums: zeroed before the sum over the half's ranks
__global__ void k_mask_owned(float* __restrict__ t0, int n, unsigned ownedMask)
{
    const int k = threadIdx.x;
    if (k < n && !((ownedMask >> k) & 1u)) t0[k] = 0.f;
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

// images per class; one thread per image
__global__ void k_count_cls(int* __restrict__ count, const int* __restrict__ cls, int nImg, int nK)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < nImg && cls[l] >= 0 && cls[l] < nK) atomicAdd(count + cls[l], 1);
}

}  // namespace thx

using namespace thx;


// ---- the C ABI this driver sequences (declared in include/thunder_amd.h, which thx_common.h includes) ----

struct thx_refine_stats_acc {
    double ms = 0;
    long launches = 0, images = 0;
};

struct thx_refine {
    thx_refine_config cfg;
    thx_comm* hemi = nullptr;
    thx_comm* world = nullptr;
    int N, pf, P, nc, rU, rSig, nPxl, nPxlM, nPxlS = 0, nV, nK, nImg, batch, scanBatch = 0;
    // frequency cut-offs of the next iteration (thx_refine_set_cutoff; Nyquist by default): rE = Optimiser::_r (expectation's list,
    // Projector::_maxRadius), rU = Model::_rU (reconstruction's list, Reconstructor::_maxRadius), rS = the scan's radius = min(cfg.rScan,
    // rE); size = Reconstructor::_size = min(N, (rU + ceil(a)) * 2) after resizeSpace, PF = pf * size = the grid of F / T / W / C.
    // Capacities (what the buffers were allocated for): the lists of rCap = N / 2 - 1, the scan's list of cfg.rScan, the grid P.
    int rE = 0, rS = 0, rCap = 0, size = 0, PF = 0, capPxl = 0, capPxlM = 0, capPxlS = 0;
    bool haveParticles = false;
    int searchType = THX_SEARCH_LOCAL;
    int halves[2];
    int lo[2], hi[2];
    std::vector<void*> owned;
    std::vector<int> gidHost;
    // pixel lists (E-step, M-step, scan)
    int *iCol, *iRow, *iPxl, *iSig, *iColM, *iRowM, *iPxlM;
    int *iColS = nullptr, *iRowS = nullptr, *iPxlS = nullptr, *iSigS = nullptr;
    // particles
    const float* imgOri = nullptr;
    float* img = nullptr;
    thx_ctf_attr* attr = nullptr;
    int* gid0 = nullptr;
    float *datM, *ctfM, *datP, *ctfP, *sigRcpP, *w;
    double* offset;
    // particle-filter state and its initial copy
    double *r, *t, *wR, *wT, *k123, *s01, *topR, *topT, *r0, *t0, *pD;
    int *cls = nullptr, *clsD = nullptr, *clsCount = nullptr;   // class of every image; of every draw of a half; histogram
    int* cls0 = nullptr;         // the classes the images were loaded with (thx_refine_set_classes; all 0 otherwise): what reset restores
    int clsCountHost[16];
    // point group
    int nSym = 0;
    std::vector<double> symMatHost;
    double* symQ = nullptr;      // DEVICE [nSym][4]
    float* symTmp = nullptr;     // one complex half grid: SYMMETRIZE_FT's destination
    // references [local half][class], accumulators
    float* refRL = nullptr;
    float *vols, *cells, *F, *T, *maps, *mapsX, *ftA, *ftB, *fscDev;
    float *sig, *sigRcp, *acc;
    thx_reco* plans[2] = {nullptr, nullptr};
    // global search: the scanned grid, its slices / ramps, one batch of scan rows and weights
    double *gridR = nullptr, *gridT = nullptr, *mats = nullptr, *pRs = nullptr, *pTs = nullptr;
    float *traS = nullptr, *rotP = nullptr, *datS = nullptr, *ctfS = nullptr, *sigS = nullptr;
    float *uC = nullptr, *uRs = nullptr, *uTs = nullptr, *baseS = nullptr;
    void* wsGlobal = nullptr;
    bool haveGrid = false;
    // CTF search
    double *dD = nullptr, *wDD = nullptr, *sD = nullptr, *topD = nullptr, *recoD = nullptr;
    float *freqD = nullptr, *defD = nullptr, *k1D = nullptr, *k2D = nullptr, *ctfD = nullptr, *uD = nullptr;
    // per-batch / per-half scratch
    double *rotB, *recoRot, *recoTran, *rotTop, *tranTop;
    float *uR, *uT, *wC, *wD, *baseL, *spec;
    void *wsExpect, *wsReduce;
    void* accInt = nullptr;      // 64-bit fixed-point accumulators of the insertion session (one half, nK classes, at a time)
    float* bounds = nullptr;     // [nImg][2] per-image bounds of the M-step rows
    int* gexp = nullptr;         // [2] the session's quanta
    float* t0Dev = nullptr;      // [16] T(0,0,0) of the classes of a half
    long nImgHemi = 0;           // images of this rank's half over all its ranks (head-room of the sums)
    unsigned iterCount = 0;
    int *active = nullptr, *nP = nullptr, *nActiveDev = nullptr;   // per-image stop rule
    double* stopState = nullptr;
    long imagePhases = 0;
    std::vector<float> fscReco;          // [nK][N / 2], fscRecoN entries each: Reconstructor::_FSC, what Model::resetReco handed over at the
                                         // end of the last iteration (the rU of THAT iteration: setFSC(_FSC.col(l)), src/Model.cpp:1122)
    int fscRecoN = 0;
    thx_refine_capture cap = {};
    // timing (HIP events on the launch stream, resolved in thx_refine_stats)
    bool timed = false;
    struct Ev { hipEvent_t a, b; int kind; int images; };
    std::vector<Ev> events;
    thx_refine_stats_acc accExpect, accInsert, accScan;
    double stageMs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long recoRounds = 0, iterations = 0;
    int lastRounds[4] = {0, 0, 0, 0};
    int lastRoundsK[64];
    int balanced[16];
    int* recoRes = nullptr;      // page-locked HOST [2][2][16][8]: what the queued reconstructions report (thx_reco_reconstruct_async_dev)
    float* norm = nullptr;       // [nImg] normCorrection: the local norms; normAll [world total] + the median behind it
    float* normAll = nullptr;
    long nImgWorld = 0, worldOffset = 0;   // particles of the whole job; of the ranks before this one
    long imgBase = -1;                     // Philox number of this rank's first image (-1: worldOffset; thx_refine_set_image_base)
    std::vector<long> worldCount;          // particles of every rank
    float lastNormMedian = 0.f, lastNormRadius = 0.f;
};

namespace {

template <typename T>
int dalloc(thx_refine* h, T** p, size_t n)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed in the refine driver: %s", n * sizeof(T), hipGetErrorString(e));
        return (int)e;
    }
    h->owned.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}

template <typename T>
int upload(thx_refine* h, T** p, const std::vector<T>& v)
{
    THX_RC(dalloc(h, p, v.size()));
    THX_CHECK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

enum { EV_EXPECT = 0, EV_INSERT = 1, EV_SCAN = 2, EV_STAGE0 = 8 };
enum { ST_ROWS = 0, ST_EXPECT, ST_SIGMA, ST_INSERT, ST_RECO, ST_RECENTRE, ST_NORM, ST_SCAN, ST_COUNT };
// Philox call numbering: call = iteration * kCallsPerIter + slot (include/thunder_amd.h, thx_refine_iterate)
enum { kCallsPerIter = 1024, SLOT_CLASS = 1, SLOT_SUPPORT = 2, SLOT_RESET = 3, SLOT_PHASE0 = 8, SLOT_DRAWS = 1000, SLOT_BALANCE = 1001 };

struct Scope {
    thx_refine* h; hipStream_t st; int kind, images; hipEvent_t a{}, b{}; bool on;
    Scope(thx_refine* h_, hipStream_t st_, int kind_, int images_ = 0) : h(h_), st(st_), kind(kind_), images(images_), on(h_->timed)
    {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    ~Scope()
    {
        if (on) { (void)hipEventRecord(b, st); h->events.push_back({a, b, kind, images}); }
    }
};

int resolve_events(thx_refine* h)
{
    for (auto& e : h->events) {
        THX_CHECK(hipEventSynchronize(e.b));
        float ms = 0;
        THX_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
        if (e.kind == EV_EXPECT) { h->accExpect.ms += ms; h->accExpect.launches++; h->accExpect.images += e.images; }
        else if (e.kind == EV_INSERT) { h->accInsert.ms += ms; h->accInsert.launches++; h->accInsert.images += e.images; }
        else if (e.kind == EV_SCAN) { h->accScan.ms += ms; h->accScan.launches++; h->accScan.images += e.images; }
        else h->stageMs[e.kind - EV_STAGE0] += ms;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    h->events.clear();
    return 0;
}

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }
unsigned call_id(const thx_refine* h, unsigned slot) { return h->iterCount * (unsigned)kCallsPerIter + slot; }
long img_base(const thx_refine* h) { return h->imgBase >= 0 ? h->imgBase : h->worldOffset; }
thx_pf_ctx pf_ctx(const thx_refine* h, int b0)
{
    thx_pf_ctx c;
    c.symQuat = h->symQ; c.nSym = h->nSym; c.img0 = (unsigned)(img_base(h) + b0);
    return c;
}
size_t vol_n(const thx_refine* h) { return (size_t)h->P * h->P * (h->P / 2 + 1); }       // projector volumes: (pf N)^3 half grid
size_t volF_n(const thx_refine* h) { return (size_t)h->PF * h->PF * (h->PF / 2 + 1); }   // F / T: (pf size)^3 half grid (resizeSpace)
size_t cell_stride(const thx_refine* h) { return thx_projector_packed_bytes(h->P) / sizeof(float); }
// volumes are indexed [local half][class]
float* vol_of(thx_refine* h, int vi, int k) { return h->vols + ((size_t)vi * h->nK + k) * vol_n(h) * 2; }
float* cells_of(thx_refine* h, int vi, int k) { return h->cells + ((size_t)vi * h->nK + k) * cell_stride(h); }
const int* vol_idx(const thx_refine* h, int b0) { return h->nK > 1 ? h->cls + b0 : nullptr; }

// ---- who reconstructs what.  The reference's ranks of a hemisphere all-reduce F / T and then ALL run the same reconstruction
// (src/Reconstructor.cpp:2383,2436).  Here class k of a half is reduced to ONE rank of the half -- rank k mod (ranks of the
// half), numbered inside the half -- reconstructed there and the N^3 map broadcast: K classes on min(K, ranks) GPUs at once and
// about half the ring traffic of an all-reduce.  THX_RECO_OWNERS=0 restores the reference's replicated form (A/B).
int ranks_of_half(const thx_refine* h, int half)
{
    const int W = h->world ? thx_comm_size(h->world) : 1;
    return W <= 1 ? 1 : (W - half + 1) / 2;
}
// which = 0: the MAP-off reconstruction of class k, 1: the MAP-on one.  The two are independent -- the MAP-on pass uses the FSC
// Model::resetReco handed over at the end of the PREVIOUS iteration -- so where a half has at least two ranks per class they go to
// two different ranks (2 k and 2 k + 1) and run at the same time; otherwise rank k mod H does both.
int owner_in_half(const thx_refine* h, int half, int k, int which)
{
    const int H = ranks_of_half(h, half);
    if (H <= 1 || knobs().recoReplicate) return 0;
    return H >= 2 * h->nK ? 2 * k + which : k % H;
}
bool owns_class(const thx_refine* h, int vi, int k, int which)
{
    const int H = h->hemi ? thx_comm_size(h->hemi) : 1;
    if (H <= 1 || knobs().recoReplicate) return true;
    return owner_in_half(h, h->halves[vi], k, which) == thx_comm_rank(h->hemi);
}
// where the sums of class k have to arrive: one rank (>= 0: ncclReduce) or, with two reconstructing ranks, everybody (-1: all-reduce)
int reduce_root(const thx_refine* h, int k)
{
    const int H = h->hemi ? thx_comm_size(h->hemi) : 1;
    if (H <= 1 || knobs().recoReplicate) return -1;
    const int half = h->halves[0], a = owner_in_half(h, half, k, 0), b = owner_in_half(h, half, k, 1);
    return a == b ? a : -1;
}

// Optimiser::allocPreCal rows of local half vi: _datP from the masked stack, _sigRcpP from the group's sigma table
int sigrcp_rows(thx_refine* h, float* dst, int vi, int l0, int n, const int* iSig, int nPxl, hipStream_t st)
{
    for (int a = 0; a < n; a += 65535) {
        const int nl = std::min(65535, n - a);
        hipLaunchKernelGGL(k_sigrcp_rows, dim3((nPxl + 255) / 256, nl), dim3(256), 0, st, dst + (size_t)a * nPxl,
                           h->sigRcp + (size_t)vi * h->cfg.nGroup * h->rSig, h->gid0 + l0 + a, iSig, nPxl, h->rSig);
    }
    THX_LAUNCH_CHECK();
    return 0;
}

int refresh_rows(thx_refine* h, int vi, hipStream_t st)
{
    const int lo = h->lo[vi], n = h->hi[vi] - lo;
    if (n <= 0) return 0;
    const size_t imgSize = (size_t)h->N * h->nc * 2;
    THX_RC(thx_gather_pixels_dev(h->datP + (size_t)lo * h->nPxl * 2, h->img + (size_t)lo * imgSize, h->iPxl, h->nPxl, h->N, n, st));
    THX_RC(sigrcp_rows(h, h->sigRcpP + (size_t)lo * h->nPxl, vi, lo, n, h->iSig, h->nPxl, st));
    return 0;
}

// Everything cut from the image stacks / CTF parameters along the CURRENT pixel lists that does not change from iteration to
// iteration: the M-step rows (the unmasked images on the rL = 0 list of rU, src/Optimiser.cpp:6722-6741) with their bounds, the CTF
// rows of both lists, the defocus search's pre-calculated rows.  (datP / sigRcpP are cut at the head of every expectation.)
int recut_rows(thx_refine* h, hipStream_t st)
{
    const thx_refine_config& c = h->cfg;
    const int n = h->nImg;
    THX_RC(thx_gather_pixels_dev(h->datM, h->imgOri, h->iPxlM, h->nPxlM, h->N, n, st));
    THX_RC(thx_ctf_dev(h->ctfM, h->attr, nullptr, c.pixelSize, h->iColM, h->iRowM, h->nPxlM, h->N, n, st));
    THX_RC(thx_ctf_dev(h->ctfP, h->attr, nullptr, c.pixelSize, h->iCol, h->iRow, h->nPxl, h->N, n, st));
    THX_RC(thx_insert_bounds_dev(h->bounds, h->datM, h->ctfM, h->nPxlM, n, st));
    if (c.mLD > 0)   // allocPreCal(.., ctf = true), src/Optimiser.cpp:8124-8169
        THX_RC(thx_expect_precal_dev(h->freqD, h->defD, h->k1D, h->k2D, h->attr, h->N, c.pixelSize, h->iCol, h->iRow, h->nPxl, n, st));
    return 0;
}

// The frequency cut-offs of the next iteration (thx_refine_set_cutoff; also how thx_refine_create sets up Nyquist).
//   r  = Optimiser::_r : allocPreCalIdx(_r, _rL) for the expectation (src/Optimiser.cpp:631,1693) -- a global search scans on it as
//        well: the scan's radius is min(cfg.rScan, r) --, Projector::_maxRadius (Model::refreshProj, src/Model.cpp:1042: where
//        allReduceSigma's and normCorrection's slices end), normCorrection's rNorm = min(_r, .) (:6203);
//   rU = Model::_rU   : allocPreCalIdx(rU, 0) for the reconstruction (:6722-6741), Reconstructor::setMaxRadius(rU) and
//        resizeSpace(min(_size, (rU + ceil(a)) * 2)) (Model::resetReco, src/Model.cpp:1100-1125; src/Reconstructor.cpp:184-198):
//        F / T / W / C and the gridding loop live on the (pf size)^3 grid, the last step pads F W into (pf N)^3 (:1677-1701);
//        the FSC of compareTwoHemispheres has rU shells.
int apply_cutoff(thx_refine* h, int r, int rU, hipStream_t st)
{
    const thx_refine_config& c = h->cfg;
    THX_REQUIRE(r > c.rL && r <= h->rCap && rU > 0 && rU <= h->rCap, "cut-offs: rL < r <= N / 2 - 1, 0 < rU <= N / 2 - 1");
    PixelList pl = pixel_list_host(c.N, r, c.rL), plM = pixel_list_host(c.N, rU, 0);
    if (c.pixelOrder == 1) morton_order(pl, c.N);
    THX_REQUIRE(pl.nPxl > 0 && pl.nPxl <= h->capPxl && plM.nPxl > 0 && plM.nPxl <= h->capPxlM, "pixel list outside the allocated capacity");
    THX_CHECK(hipStreamSynchronize(st));   // (queued kernels may still read the lists)
    auto put = [](int* d, const std::vector<int>& v) { return hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice); };
    THX_CHECK(put(h->iCol, pl.iCol)); THX_CHECK(put(h->iRow, pl.iRow)); THX_CHECK(put(h->iPxl, pl.iPxl)); THX_CHECK(put(h->iSig, pl.iSig));
    THX_CHECK(put(h->iColM, plM.iCol)); THX_CHECK(put(h->iRowM, plM.iRow)); THX_CHECK(put(h->iPxlM, plM.iPxl));
    h->rE = r; h->rU = rU; h->nPxl = pl.nPxl; h->nPxlM = plM.nPxl;
    if (c.nR > 0) {   // the scan's list: allocPreCalIdx(_r, _rL) as well, never beyond the radius the scan buffers were sized for
        h->rS = std::min(c.rScan, r);
        PixelList plS = pixel_list_host(c.N, h->rS, c.rL);
        if (c.pixelOrder == 1) morton_order(plS, c.N);
        THX_REQUIRE(plS.nPxl > 0 && plS.nPxl <= h->capPxlS, "scan pixel list outside the allocated capacity");
        THX_CHECK(put(h->iColS, plS.iCol)); THX_CHECK(put(h->iRowS, plS.iRow)); THX_CHECK(put(h->iPxlS, plS.iPxl)); THX_CHECK(put(h->iSigS, plS.iSig));
        h->nPxlS = plS.nPxl;
        if (h->haveGrid) THX_RC(thx_translate_dev(h->traS, h->gridT, c.nT, h->iColS, h->iRowS, h->nPxlS, h->N, st));
    }
    // Reconstructor::resizeSpace: _size = min(N, (rU + CEIL(_a)) * 2), a = 1.9; the plans (W, C, scratch, FFT plans) follow the size
    const int size = std::min(c.N, (rU + 2) * 2);
    if (size != h->size) {
        for (int v = 0; v < h->nV; v++) {
            if (h->plans[v]) { THX_RC(thx_reco_destroy(h->plans[v])); h->plans[v] = nullptr; }
            THX_RC(thx_reco_create(&h->plans[v], size, c.N, c.pf, 1.9f, 15.0f));
        }
        h->size = size;
        h->PF = c.pf * size;
    }
    if (h->haveParticles) THX_RC(recut_rows(h, st));
    if (h->iterCount == 0) {   // Model::initProjReco: _reco[l]->setFSC(vec::Constant(_rU, 1)), src/Model.cpp:1086
        h->fscReco.assign((size_t)h->nK * (c.N / 2), 1.0f);
        h->fscRecoN = rU;
    }
    return 0;
}

// Optimiser::expectation's global search over the images of local half vi (src/Optimiser.cpp:631-1140), batch by batch:
// rows at the scan radius; per class the nR slices of its reference and the contraction against every image and shift
// (weights and running baseline carried from class to class, :756-894); the class of every image (:925-952); the support
// points of its local search from the scan posterior of that class, with the scanning phase's minimum spread (:953-1079).
int scan_and_select(thx_refine* h, int vi, hipStream_t st)
{
    const thx_refine_config& c = h->cfg;
    const int lo = h->lo[vi];
    const size_t imgSize = (size_t)h->N * h->nc * 2;
    THX_REQUIRE(h->haveGrid, "thx_refine_set_grid has not been called (global search)");
    for (int b0 = lo; b0 < h->hi[vi]; b0 += h->scanBatch) {
        const int nb = std::min(h->scanBatch, h->hi[vi] - b0);
        // allocPreCal(true, true, false) at the scan's radius: masked images, CTF rows, sigma rows (:635-638)
        THX_RC(thx_gather_pixels_dev(h->datS, h->img + (size_t)b0 * imgSize, h->iPxlS, h->nPxlS, h->N, nb, st));
        THX_RC(thx_ctf_dev(h->ctfS, h->attr + b0, nullptr, c.pixelSize, h->iColS, h->iRowS, h->nPxlS, h->N, nb, st));
        THX_RC(sigrcp_rows(h, h->sigS, vi, b0, nb, h->iSigS, h->nPxlS, st));
        THX_CHECK(hipMemsetAsync(h->uC, 0, (size_t)nb * h->nK * sizeof(float), st));
        THX_CHECK(hipMemsetAsync(h->uRs, 0, (size_t)h->nK * nb * c.nR * sizeof(float), st));
        THX_CHECK(hipMemsetAsync(h->uTs, 0, (size_t)h->nK * nb * c.nT * sizeof(float), st));
        hipLaunchKernelGGL(k_fill_nan, dim3(blocks_for(nb)), dim3(256), 0, st, h->baseS, (size_t)nb);   // "unset", :737-745
        THX_LAUNCH_CHECK();
        for (int k = 0; k < h->nK; k++) {
            THX_RC(thx_project_dev(vol_of(h, vi, k), h->rotP, h->mats, h->iColS, h->iRowS, c.nR, h->pf, h->P, h->nPxlS, st));   // :775-781
            Scope e(h, st, EV_SCAN, nb);
            THX_RC(thx_expect_global_dev(h->rotP, h->traS, h->datS, h->ctfS, h->sigS, h->pRs, h->pTs, h->uC, h->uRs, h->uTs, h->baseS, k,
                                         h->nK, c.nR, c.nT, h->nPxlS, nb, h->wsGlobal, st));
        }
        if (h->cap.scanUC) THX_CHECK(hipMemcpyAsync(h->cap.scanUC + (size_t)b0 * h->nK, h->uC, (size_t)nb * h->nK * sizeof(float), hipMemcpyDeviceToDevice, st));
        for (int k = 0; k < h->nK; k++) {   // [nK][nb][.] of the batch -> [nImg][nK][.] of the trace
            if (h->cap.scanUR)
                THX_CHECK(hipMemcpy2DAsync(h->cap.scanUR + ((size_t)b0 * h->nK + k) * c.nR, (size_t)h->nK * c.nR * sizeof(float),
                                           h->uRs + (size_t)k * nb * c.nR, (size_t)c.nR * sizeof(float), (size_t)c.nR * sizeof(float), nb,
                                           hipMemcpyDeviceToDevice, st));
            if (h->cap.scanUT)
                THX_CHECK(hipMemcpy2DAsync(h->cap.scanUT + ((size_t)b0 * h->nK + k) * c.nT, (size_t)h->nK * c.nT * sizeof(float),
                                           h->uTs + (size_t)k * nb * c.nT, (size_t)c.nT * sizeof(float), (size_t)c.nT * sizeof(float), nb,
                                           hipMemcpyDeviceToDevice, st));
        }
        const thx_pf_ctx ctx = pf_ctx(h, b0);
        THX_RC(thx_pf_class_select_ex_dev(h->cls + b0, h->uC, nullptr, nb, h->nK, c.peakFactorC, c.seed, call_id(h, SLOT_CLASS), ctx.img0, st));
        THX_RC(thx_pf_scan_support_ex_dev(h->r + (size_t)b0 * c.mLR * 4, h->t + (size_t)b0 * c.mLT * 2, h->wR + (size_t)b0 * c.mLR,
                                          h->wT + (size_t)b0 * c.mLT, h->k123 + (size_t)b0 * 3, h->s01 + (size_t)b0 * 2,
                                          h->topR + (size_t)b0 * 4, h->topT + (size_t)b0 * 2, h->gridR, h->gridT, h->uRs, h->uTs, h->cls + b0,
                                          nb, nb, c.nR, c.nT, c.mLR, c.mLT, c.peakFactorR, c.scanMinK, c.scanMinS, c.seed,
                                          call_id(h, SLOT_SUPPORT), &ctx, st));
    }
    const int n = h->hi[vi] - lo;
    if (h->cap.r0) THX_CHECK(hipMemcpyAsync(h->cap.r0 + (size_t)lo * c.mLR * 4, h->r + (size_t)lo * c.mLR * 4, (size_t)n * c.mLR * 4 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (h->cap.t0) THX_CHECK(hipMemcpyAsync(h->cap.t0 + (size_t)lo * c.mLT * 2, h->t + (size_t)lo * c.mLT * 2, (size_t)n * c.mLT * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (h->cap.k0) THX_CHECK(hipMemcpyAsync(h->cap.k0 + (size_t)lo * 3, h->k123 + (size_t)lo * 3, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (h->cap.s0) THX_CHECK(hipMemcpyAsync(h->cap.s0 + (size_t)lo * 2, h->s01 + (size_t)lo * 2, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    return 0;
}

// HOT LOOP B: the particle-filter phases over the images of local half vi.  cfg.maxPhase <= nPhase: exactly nPhase phases
// per image (the fixed-work iteration).  Otherwise the reference's per-image stop rule (src/Optimiser.cpp:1510-1615): from
// phase index nPhase (= MIN_N_PHASE_PER_ITER_LOCAL / _GLOBAL) on every image's variances are checked after the phase, images
// without a decrease drop out (device mask: their workgroups return at once), and the loop ends when no image of the half is
// left.  After a global scan the phase index starts at 1 and every phase uses perturbFactorSGlobal (:1185-1212).
int expectation(thx_refine* h, int vi, hipStream_t st)
{
    const thx_refine_config& c = h->cfg;
    const int lo = h->lo[vi], n = h->hi[vi] - lo;
    if (n <= 0) return 0;
    const bool global = h->searchType == THX_SEARCH_GLOBAL, ctf = h->searchType == THX_SEARCH_CTF;
    const int nD = ctf ? c.mLD : 1;
    const bool rule = c.maxPhase > c.nPhase;
    const int p0 = global ? 1 : 0;
    const int pEnd = rule ? c.maxPhase : p0 + c.nPhase;   // phase indices [p0, pEnd)
    if (rule) THX_RC(thx_pf_stop_init_dev(h->active + lo, h->nP + lo, h->stopState + (size_t)lo * 8, c.transS, ctf ? c.ctfRefineS : 0.01, n, st));
    const int capP = h->cap.phases > 0 ? h->cap.phases : c.nPhase;   // depth of the optional per-phase trace
    long imagePhases = 0;
    int nActive = n;
    for (int p = p0; p < pEnd && nActive > 0; p++) {
        const int pi = p - p0;
        for (int b0 = lo; b0 < h->hi[vi]; b0 += h->batch) {
            const int nb = std::min(h->batch, h->hi[vi] - b0);
            double* r = h->r + (size_t)b0 * c.mLR * 4;
            double* t = h->t + (size_t)b0 * c.mLT * 2;
            double* wR = h->wR + (size_t)b0 * c.mLR;
            double* wT = h->wT + (size_t)b0 * c.mLT;
            double* k = h->k123 + (size_t)b0 * 3;
            double* s = h->s01 + (size_t)b0 * 2;
            const int* act = rule ? h->active + b0 : nullptr;
            const double f = p == 0 ? c.pfL : (global ? c.pfSGlobal : c.pfS);
            const thx_pf_ctx ctx = pf_ctx(h, b0);
            const unsigned callP = call_id(h, SLOT_PHASE0 + 2 * p), callU = callP + 1;
            // Particle::perturb, then the phase's support points are the filter's own (src/Optimiser.cpp:1186-1208)
            THX_RC(thx_pf_perturb_ex_dev(r, t, wR, wT, k, s, nb, c.mLR, c.mLT, f, f, c.transS, c.transQ, c.seed, callP, act, &ctx, st));
            const double* pDb = h->pD;
            const float* ctfRows = h->ctfP + (size_t)b0 * h->nPxl;
            if (ctf) {   // initD in phase 0, perturb(perturbFactorSCTF, PAR_D) afterwards (:1196-1209); CTF rows per defocus factor (:1246-1272)
                double* d = h->dD + (size_t)b0 * nD;
                THX_RC(thx_pf_perturb_d_ex_dev(d, h->wDD + (size_t)b0 * nD, h->sD + b0, nb, nD, p == 0 ? c.ctfRefineS : c.pfSCTF, p == 0 ? 1 : 0,
                                               c.seed, callP, act, ctx.img0, st));
                THX_RC(thx_ctf_dsearch_dev(h->ctfD, h->freqD, h->defD + (size_t)b0 * h->nPxl, h->k1D + b0, h->k2D + b0, h->attr + b0, d, nD,
                                           h->nPxl, nb, st));
                pDb = h->wDD + (size_t)b0 * nD;
                ctfRows = h->ctfD;
                if (pi < capP && h->cap.dP)
                    THX_CHECK(hipMemcpyAsync(h->cap.dP + ((size_t)pi * h->nImg + b0) * nD, d, (size_t)nb * nD * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
            THX_RC(thx_rotmat_dev(r, h->rotB, nb * c.mLR, st));
            if (pi < capP && (h->cap.rP || h->cap.tP || h->cap.wRP || h->cap.wTP)) {
                const size_t at = (size_t)pi * h->nImg + b0;
                if (h->cap.rP) THX_CHECK(hipMemcpyAsync(h->cap.rP + at * c.mLR * 4, r, (size_t)nb * c.mLR * 4 * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (h->cap.tP) THX_CHECK(hipMemcpyAsync(h->cap.tP + at * c.mLT * 2, t, (size_t)nb * c.mLT * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (h->cap.wRP) THX_CHECK(hipMemcpyAsync(h->cap.wRP + at * c.mLR, wR, (size_t)nb * c.mLR * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (h->cap.wTP) THX_CHECK(hipMemcpyAsync(h->cap.wTP + at * c.mLT, wT, (size_t)nb * c.mLT * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
            {
                Scope ev(h, st, EV_EXPECT, nb);
                const int wg = (p > 0 && knobs().expectWgLater >= 0) ? knobs().expectWgLater : c.wgPerCU;   // (A/B knob: the clouds of later phases are tighter)
                if (!ctf)
                    THX_RC(thx_expect_local_packed_dev(cells_of(h, vi, 0), vol_idx(h, b0), h->P, h->pf, h->N, h->iCol, h->iRow, h->nPxl, nb,
                                                       h->datP + (size_t)b0 * h->nPxl * 2, ctfRows, h->sigRcpP + (size_t)b0 * h->nPxl, h->rotB,
                                                       c.mLR, t, c.mLT, 1, nullptr, wR, wT, pDb, h->wC, h->uR, h->uT, h->wD, h->baseL, nullptr,
                                                       h->wsExpect, wg, act, st));
                else   // (CTF search: the fused defocus kernel, on the cell-packed references as well since round 5)
                    THX_RC(thx_expect_local_packed_dev(cells_of(h, vi, 0), vol_idx(h, b0), h->P, h->pf, h->N, h->iCol, h->iRow, h->nPxl, nb,
                                                h->datP + (size_t)b0 * h->nPxl * 2, ctfRows, h->sigRcpP + (size_t)b0 * h->nPxl, h->rotB, c.mLR, t,
                                                c.mLT, nD, nullptr, wR, wT, pDb, h->wC, h->uR, h->uT, h->uD, h->baseL, nullptr, h->wsExpect,
                                                c.wgPerCU, act, st));
            }
            THX_RC(thx_pf_update_ex_dev(r, t, wR, wT, h->uR, h->uT, k, s, h->topR + (size_t)b0 * 4, h->topT + (size_t)b0 * 2, nb, c.mLR,
                                        c.mLT, c.peakFactorR, c.seed, callU, act, &ctx, st));
            if (ctf) {
                if (pi < capP && h->cap.uD)
                    THX_CHECK(hipMemcpyAsync(h->cap.uD + ((size_t)pi * h->nImg + b0) * nD, h->uD, (size_t)nb * nD * sizeof(float), hipMemcpyDeviceToDevice, st));
                THX_RC(thx_pf_update_d_ex_dev(h->dD + (size_t)b0 * nD, h->wDD + (size_t)b0 * nD, h->uD, h->sD + b0, h->topD + b0, nb, nD, c.seed,
                                              callU, act, ctx.img0, st));
                if (pi < capP && h->cap.dR)
                    THX_CHECK(hipMemcpyAsync(h->cap.dR + ((size_t)pi * h->nImg + b0) * nD, h->dD + (size_t)b0 * nD, (size_t)nb * nD * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
            if (pi < capP) {   // optional trace for the chain-level parity tests
                const size_t at = (size_t)pi * h->nImg + b0;
                const thx_refine_capture& cp = h->cap;
                if (cp.uR) THX_CHECK(hipMemcpyAsync(cp.uR + at * c.mLR, h->uR, (size_t)nb * c.mLR * sizeof(float), hipMemcpyDeviceToDevice, st));
                if (cp.uT) THX_CHECK(hipMemcpyAsync(cp.uT + at * c.mLT, h->uT, (size_t)nb * c.mLT * sizeof(float), hipMemcpyDeviceToDevice, st));
                if (cp.r) THX_CHECK(hipMemcpyAsync(cp.r + at * c.mLR * 4, r, (size_t)nb * c.mLR * 4 * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (cp.t) THX_CHECK(hipMemcpyAsync(cp.t + at * c.mLT * 2, t, (size_t)nb * c.mLT * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (cp.k123) THX_CHECK(hipMemcpyAsync(cp.k123 + at * 3, k, (size_t)nb * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (cp.s01) THX_CHECK(hipMemcpyAsync(cp.s01 + at * 2, s, (size_t)nb * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
        }
        imagePhases += nActive;
        if (rule && p >= c.nPhase) {
            THX_CHECK(hipMemsetAsync(h->nActiveDev, 0, sizeof(int), st));
            THX_RC(thx_pf_stop_rule_dev(h->active + lo, h->nP + lo, h->stopState + (size_t)lo * 8, h->k123 + (size_t)lo * 3,
                                        h->s01 + (size_t)lo * 2, ctf ? h->sD + lo : nullptr, p, n, h->nActiveDev, st));
            THX_CHECK(hipMemcpyAsync(&nActive, h->nActiveDev, sizeof(int), hipMemcpyDeviceToHost, st));
            THX_CHECK(hipStreamSynchronize(st));
        }
    }
    h->imagePhases += imagePhases;
    return 0;
}

// Optimiser::allReduceSigma (src/Optimiser.cpp:6395-6710) for local half vi, from the top pose of the last phase against the
// reference of the image's class
int sigma_update(thx_refine* h, int vi, hipStream_t st)
{
    const thx_refine_config& c = h->cfg;
    const int lo = h->lo[vi], n = h->hi[vi] - lo;
    if (n <= 0) return 0;
    const size_t imgSize = (size_t)h->N * h->nc * 2;
    THX_RC(thx_rotmat_dev(h->topR + (size_t)lo * 4, h->rotTop, n, st));
    THX_CHECK(hipMemcpyAsync(h->tranTop, h->topT + (size_t)lo * 2, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    const double* dfac = h->searchType == THX_SEARCH_CTF ? h->topD + lo : nullptr;
    // (the slice is cut at Projector::_maxRadius = _r, Model::refreshProj src/Model.cpp:1042; the spectra run over the whole frequency range)
    THX_RC(thx_sigma_spectra_packed_dev(h->spec, cells_of(h, vi, 0), vol_idx(h, lo), h->P, h->pf, h->N, h->rE, h->rSig,
                                 h->img + (size_t)lo * imgSize, h->imgOri + (size_t)lo * imgSize, h->attr + lo, dfac, c.pixelSize,
                                 h->rotTop, h->tranTop, h->offset + (size_t)lo * 2, n, st));
    const size_t tab = (size_t)c.nGroup * (h->rSig + 1);
    THX_CHECK(hipMemsetAsync(h->acc, 0, 3 * tab * sizeof(float), st));
    THX_RC(thx_sigma_accum_dev(h->acc, h->acc + tab, h->acc + 2 * tab, h->spec, h->gidHost.data() + lo, n, c.nGroup, h->rSig,
                               c.groupSig, st));
    THX_RC(thx_comm_allreduce_f32(h->hemi, h->acc, 3 * tab, st));   // :6608-6650, the three tables in one collective
    THX_RC(thx_sigma_final_dev(h->sig + (size_t)vi * c.nGroup * h->rSig, h->sigRcp + (size_t)vi * c.nGroup * h->rSig, h->acc,
                               h->acc + tab, h->acc + 2 * tab, c.nGroup, h->rSig, c.groupSig, c.maskRadiusPx * c.pixelSize, h->N,
                               c.pixelSize, st));
    return 0;
}

// Optimiser::normCorrection (src/Optimiser.cpp:6201-6394) at the head of Optimiser::maximization (:3405-3413): the residual power
// of every image against its top pose's slice over rL <= r < rNorm, the median over ALL particles of the job, both stacks
// rescaled in place; then the rows and bounds that were cut from _imgOri are cut again.
int norm_correction(thx_refine* h, hipStream_t st)
{
    const thx_refine_config& c = h->cfg;
    const size_t imgSize = (size_t)h->N * h->nc * 2;
    // rNorm = min(_r, _model.resolutionP(0.75, false)): the largest over the classes of resP(_FSC.col(k), 0.75, 1, 1, false)
    // (src/Model.cpp:977-994, src/Functions/Spectrum.cpp:339-363) on the FSC the previous iteration left in the model
    int res = 0;
    for (int k = 0; k < h->nK; k++) {
        const float* f = h->fscReco.data() + (size_t)k * (h->N / 2);
        int rk = 1;
        for (; rk < h->fscRecoN; rk++)
            if (f[rk] < 0.75f) break;
        res = std::max(res, rk - 1);
    }
    const float rNorm = std::min((float)h->rE, (float)res);   // TSGSL_MIN_RFLOAT(_r, _model.resolutionP(0.75, false)), :6203
    for (int vi = 0; vi < h->nV; vi++) {
        const int lo = h->lo[vi], n = h->hi[vi] - lo;
        if (n <= 0) continue;
        THX_RC(thx_rotmat_dev(h->topR + (size_t)lo * 4, h->rotTop, n, st));
        THX_RC(thx_norm_residual_packed_dev(h->norm + lo, cells_of(h, vi, 0), vol_idx(h, lo), h->P, h->pf, h->N, h->rE, (float)c.rL, rNorm,
                                     h->img + (size_t)lo * imgSize, h->attr + lo, h->searchType == THX_SEARCH_CTF ? h->topD + lo : nullptr,
                                     c.pixelSize, h->rotTop, h->topT + (size_t)lo * 2, n, st));
    }
    // norm of every particle of the job on every rank: each rank's norms at its offset of a zeroed vector, summed over world
    // (MPI_Allreduce(MPI_IN_PLACE, norm.data(), norm.size(), .., MPI_SUM, MPI_COMM_WORLD), :6362-6367)
    const float* all = h->norm;
    long nAll = h->nImg;
    if (h->world && thx_comm_size(h->world) > 1) {
        THX_CHECK(hipMemsetAsync(h->normAll, 0, (size_t)h->nImgWorld * sizeof(float), st));
        THX_CHECK(hipMemcpyAsync(h->normAll + h->worldOffset, h->norm, (size_t)h->nImg * sizeof(float), hipMemcpyDeviceToDevice, st));
        THX_RC(thx_comm_allreduce_f32(h->world, h->normAll, (size_t)h->nImgWorld, st));
        all = h->normAll;
        nAll = h->nImgWorld;
    }
    float* med = h->normAll + h->nImgWorld;
    THX_RC(thx_median_f32_dev(med, all, (int)nAll, st));
    THX_RC(thx_norm_scale_dev(h->img, const_cast<float*>(h->imgOri), h->norm, med, h->N, h->nImg, st));
    THX_RC(thx_gather_pixels_dev(h->datM, h->imgOri, h->iPxlM, h->nPxlM, h->N, h->nImg, st));
    THX_RC(thx_insert_bounds_dev(h->bounds, h->datM, h->ctfM, h->nPxlM, h->nImg, st));
    THX_CHECK(hipMemcpyAsync(&h->lastNormMedian, med, sizeof(float), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipStreamSynchronize(st));
    h->lastNormRadius = rNorm;
    return 0;
}

// HOT LOOP C: mReco draws per image, trilinear insertion into the F / T of the image's class
int insertion(thx_refine* h, int vi, hipStream_t st)
{
    const thx_refine_config& c = h->cfg;
    const int lo = h->lo[vi], n = h->hi[vi] - lo;
    const size_t volN = volF_n(h);   // F / T live on the reconstructors' grid PF = pf * size (Reconstructor::allocSpace after resizeSpace)
    const bool ctf = h->searchType == THX_SEARCH_CTF;
    float* F = h->F + (size_t)vi * h->nK * volN * 2;
    float* T = h->T + (size_t)vi * h->nK * volN;
    THX_CHECK(hipMemsetAsync(F, 0, (size_t)h->nK * volN * 2 * sizeof(float), st));
    THX_CHECK(hipMemsetAsync(T, 0, (size_t)h->nK * volN * sizeof(float), st));
    // one insertion SESSION per half: common quanta over every image of the half (all ranks), 64-bit accumulators zeroed once,
    // all batches accumulate into them, the half-set reduce runs on the integers (N ranks == 1 rank, bit for bit), and only
    // then do they become the float F / T that prepareTF normalises
    THX_RC(thx_insert_scale_dev(h->gexp, h->bounds + (size_t)lo * 2, h->w + lo, n > 0 ? n : 0, c.mReco, ctf ? 1 : 0, h->nImgHemi, h->hemi, st));
    THX_CHECK(hipMemsetAsync(h->accInt, 0, thx_insert_acc_bytes(h->PF, h->nK), st));
    if (n > 0) {
        hipLaunchKernelGGL(k_draw_reco, dim3(blocks_for((size_t)n * c.mReco)), dim3(256), 0, st, h->recoRot, h->recoTran,
                           ctf ? h->recoD : nullptr, h->r + (size_t)lo * c.mLR * 4, h->t + (size_t)lo * c.mLT * 2,
                           ctf ? h->dD + (size_t)lo * c.mLD : nullptr, n, c.mLR, c.mLT, ctf ? c.mLD : 0, c.mReco, c.seed,
                           call_id(h, SLOT_DRAWS), (unsigned)(img_base(h) + lo));
        if (h->nK > 1)   // every draw of an image goes to the image's class (src/Optimiser.cpp:7129-7150)
            hipLaunchKernelGGL(k_expand_cls, dim3(blocks_for((size_t)n * c.mReco)), dim3(256), 0, st, h->clsD, h->cls + lo, n, c.mReco);
        THX_LAUNCH_CHECK();
    }
    for (int b0 = lo; b0 < h->hi[vi]; b0 += h->batch) {
        const int nb = std::min(h->batch, h->hi[vi] - b0);
        Scope ev(h, st, EV_INSERT, nb);
        THX_RC(thx_insert_accumulate_dev(h->accInt, h->gexp, h->bounds + (size_t)b0 * 2, nullptr, nullptr, h->PF, h->nK,
                                         h->datM + (size_t)b0 * h->nPxlM * 2, h->ctfM + (size_t)b0 * h->nPxlM, h->w + b0,
                                         h->recoRot + (size_t)(b0 - lo) * c.mReco * 9, h->recoTran + (size_t)(b0 - lo) * c.mReco * 2,
                                         h->offset + (size_t)b0 * 2, h->nK > 1 ? h->clsD + (size_t)(b0 - lo) * c.mReco : nullptr,
                                         ctf ? h->attr + b0 : nullptr, ctf ? h->recoD + (size_t)(b0 - lo) * c.mReco : nullptr, ctf ? 1 : 0,
                                         c.pixelSize, h->iColM, h->iRowM, h->pf, h->nPxlM, c.mReco, h->N, nb, st));
    }
    // the half-set reduce on the integers, class by class through one workspace, towards the rank that reconstructs the class
    for (int k = 0; k < h->nK && h->hemi; k++)
        THX_RC(thx_reco_reduce_acc_class(h->hemi, h->accInt, h->nK, k, reduce_root(h, k), h->PF, h->rU, h->pf, h->wsReduce, st));
    THX_RC(thx_insert_finish_dev(F, T, h->accInt, h->gexp, h->PF, h->nK, st));
    return 0;
}

int refresh_projector(thx_refine* h, int vi, int k, const float* mapRL, hipStream_t st)
{
    THX_RC(thx_reco_set_projectee_dev(h->plans[vi], mapRL, vol_of(h, vi, k), st));   // Model::refreshProj
    THX_RC(thx_projector_pack_dev(cells_of(h, vi, k), vol_of(h, vi, k), h->P, 1, st));
    return 0;
}

// images per class over all ranks (Optimiser::refreshClassDistr, src/Optimiser.cpp:5484-5516: with one support class per image
// Particle::rand(cls) is the image's class); this rank's counts go to the statistics
int class_distribution(thx_refine* h, std::vector<double>& distr, hipStream_t st)
{
    THX_CHECK(hipMemsetAsync(h->clsCount, 0, 16 * sizeof(int), st));
    hipLaunchKernelGGL(k_count_cls, dim3(blocks_for(h->nImg)), dim3(256), 0, st, h->clsCount, h->cls, h->nImg, h->nK);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpyAsync(h->clsCountHost, h->clsCount, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
    if (h->world && thx_comm_size(h->world) > 1) THX_RC(thx_comm_allreduce_i32(h->world, h->clsCount, 16, st));
    int tot[16];
    THX_CHECK(hipMemcpyAsync(tot, h->clsCount, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipStreamSynchronize(st));
    double sum = 0;
    for (int k = 0; k < h->nK; k++) sum += tot[k];
    distr.assign(h->nK, 0.0);
    for (int k = 0; k < h->nK; k++) distr[k] = sum > 0 ? tot[k] / sum : 0.0;
    return 0;
}

// Optimiser::determineBalanceClass (src/Optimiser.cpp:5518-5584): classes below thres / K of the images are "empty"; each is
// reassigned to a class drawn from the cumulative distribution of (cDistr - thres / K) over the others.  The reference draws
// on the master and broadcasts; here every rank evaluates the same Philox stream on the same all-reduced distribution.
void determine_balance(const thx_refine* h, const std::vector<double>& distr, int* bm)
{
    const int K = h->nK;
    const double thres = 0.05 / K;   // CLASS_BALANCE_FACTOR / _para.k
    std::vector<double> cum(K, 0.0);
    double sum = 0;
    for (int t = 0; t < K; t++) { cum[t] = distr[t] < thres ? 0.0 : distr[t] - thres; sum += cum[t]; }
    double acc = 0;
    for (int t = 0; t < K; t++) { acc += cum[t] / sum; cum[t] = acc; }
    for (int t = 0; t < K; t++) {
        bm[t] = -1;
        if (!(distr[t] < thres) || !(sum > 0)) continue;
        double u[4];
        draw_u4(u, h->cfg.seed, 0u, call_id(h, SLOT_BALANCE), 14u, (unsigned)t);
        const float indice = (float)u[0];   // RFLOAT indice = TSGSL_ran_flat(engine, 0, 1)
        int j = 0;
        while (j < K - 1 && cum[j] < indice) j++;
        bm[t] = j;
    }
}

}  // namespace

extern "C" {

int thx_pixel_list_host(int N, int rU, int rL, int order, int* iCol, int* iRow, int* iPxl, int* iSig, int* nPxl)
{
    THX_REQUIRE(nPxl && N > 0 && rU > 0 && rL >= 0, "bad arguments");
    PixelList pl = pixel_list_host(N, rU, rL);
    if (order == 1) morton_order(pl, N);
    *nPxl = pl.nPxl;
    if (iCol) memcpy(iCol, pl.iCol.data(), pl.nPxl * sizeof(int));
    if (iRow) memcpy(iRow, pl.iRow.data(), pl.nPxl * sizeof(int));
    if (iPxl) memcpy(iPxl, pl.iPxl.data(), pl.nPxl * sizeof(int));
    if (iSig) memcpy(iSig, pl.iSig.data(), pl.nPxl * sizeof(int));
    return 0;
}

int thx_view_order_host(const double* quat, int n, int* perm)
{
    THX_REQUIRE(quat && perm && n >= 0, "bad arguments");
    std::vector<unsigned> key(n);
    for (int l = 0; l < n; l++) {
        const double q0 = quat[4 * (size_t)l], q1 = quat[4 * (size_t)l + 1], q2 = quat[4 * (size_t)l + 2], q3 = quat[4 * (size_t)l + 3];
        double nx = 2 * (q1 * q3 + q0 * q2), ny = 2 * (q2 * q3 - q0 * q1), nz = 1 - 2 * (q1 * q1 + q2 * q2);   // R e_z
        if (nz < 0) { nx = -nx; ny = -ny; nz = -nz; }
        const double s = sqrt(1.0 / (1.0 + nz));   // Lambert azimuthal equal-area, scaled to the unit disc
        double X = floor((nx * s + 1) * 0.5 * 1023), Y = floor((ny * s + 1) * 0.5 * 1023);
        X = X < 0 ? 0 : (X > 1023 ? 1023 : X);
        Y = Y < 0 ? 0 : (Y > 1023 ? 1023 : Y);
        key[l] = spread_bits((unsigned)X) | (spread_bits((unsigned)Y) << 1);
    }
    for (int l = 0; l < n; l++) perm[l] = l;
    std::stable_sort(perm, perm + n, [&](int a, int b) { return key[a] < key[b]; });
    return 0;
}

int thx_draw_reco_dev(double* recoRot, double* recoTran, const double* r, const double* t, int nImg, int nR, int nT, int mReco,
                      unsigned long long seed, unsigned call, unsigned img0, void* stream)
{
    if (nImg <= 0 || mReco <= 0) return 0;
    THX_REQUIRE(recoRot && recoTran && r && t && nR > 0 && nT > 0, "bad arguments");
    hipLaunchKernelGGL(k_draw_reco, dim3(blocks_for((size_t)nImg * mReco)), dim3(256), 0, as_stream(stream), recoRot, recoTran,
                       (double*)nullptr, r, t, (const double*)nullptr, nImg, nR, nT, 0, mReco, seed, call, img0);
    THX_LAUNCH_CHECK();
    return 0;
}

int thx_refine_destroy(thx_refine* h)
{
    if (!h) return 0;
    (void)hipDeviceSynchronize();
    for (auto& e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int v = 0; v < 2; v++)
        if (h->plans[v]) (void)thx_reco_destroy(h->plans[v]);
    for (void* p : h->owned) (void)hipFree(p);
    if (h->recoRes) (void)hipHostFree(h->recoRes);
    delete h;
    return 0;
}

int thx_refine_create(thx_refine** out, const thx_refine_config* cfg, thx_comm* hemi, thx_comm* world)
{
    THX_REQUIRE(out && cfg, "NULL argument");
    thx_refine_config c = *cfg;
    if (c.nK <= 0) c.nK = 1;
    THX_REQUIRE(c.N > 0 && (c.N % 2) == 0 && c.pf >= 1 && c.nImg > 0, "bad box / particle count");
    THX_REQUIRE(c.mLR >= 5 && c.mLR <= 256 && c.mLT >= 2 && c.mLT <= 32 && c.nPhase > 0 && c.mReco > 0 && c.mReco < 3200,
                "bad search parameters (5 <= mLR <= 256, 2 <= mLT <= 32, 0 < mReco < 3200)");
    THX_REQUIRE(c.nGroup > 0 && c.batch > 0, "bad nGroup / batch");
    THX_REQUIRE(c.halfOfRank >= -1 && c.halfOfRank <= 1, "halfOfRank must be -1 (both halves here), 0 or 1");
    THX_REQUIRE(c.nK <= 16, "at most 16 classes");
    THX_REQUIRE(c.searchType >= THX_SEARCH_LOCAL && c.searchType <= THX_SEARCH_CTF, "searchType: THX_SEARCH_LOCAL / GLOBAL / CTF");
    THX_REQUIRE(c.nSym >= 0 && (c.nSym == 0 || (c.symMat && c.symQuat)), "nSym > 0 needs symMat and symQuat");
    const bool scans = c.nR > 0;
    if (scans) {
        THX_REQUIRE(c.nR <= 16384 && c.nT > 0 && c.nT <= 16384 && c.mLR <= c.nR && c.mLT <= c.nT, "scan grid: mLR <= nR <= 16384, mLT <= nT <= 16384");
        THX_REQUIRE(c.rScan > c.rL && c.rScan <= c.N / 2 - 2, "rL < rScan <= N / 2 - 2");
    }
    THX_REQUIRE(c.searchType != THX_SEARCH_GLOBAL || scans, "a global search needs the scan grid sizes nR, nT, rScan");
    THX_REQUIRE(c.mLD >= 0 && c.mLD <= 9 && (c.searchType != THX_SEARCH_CTF || c.mLD >= 1), "0 <= mLD <= 9 (a CTF search needs mLD >= 1)");
    // thx_refine_iterate broadcasts half map h from world rank h: with several ranks the caller must follow the reference's
    // odd / even convention (src/Parallel.cpp:26-36) -- rank r owns half r mod 2 -- and no rank may hold both halves
    if (world && thx_comm_size(world) > 1) {
        THX_REQUIRE(c.halfOfRank >= 0, "halfOfRank = -1 (both halves on one rank) needs a one-rank world");
        THX_REQUIRE(c.halfOfRank == thx_comm_rank(world) % 2, "halfOfRank must equal the world rank mod 2 (rank r owns half r mod 2)");
        // the ranks of a half are numbered in world order: world rank r is rank r / 2 of half r mod 2 (the maps of class k travel
        // from world rank 2 x (k mod ranks of the half) + half)
        const int W = thx_comm_size(world), wr = thx_comm_rank(world), Hh = (W - c.halfOfRank + 1) / 2;
        THX_REQUIRE((hemi ? thx_comm_size(hemi) : 1) == Hh || (!hemi && Hh == 1), "hemi must span the world ranks of this rank's parity");
        THX_REQUIRE(!hemi || thx_comm_rank(hemi) == wr / 2, "hemi rank must be world rank / 2");
    }
    // the owners of the classes are derived from `world` (ranks_of_half / owner_in_half) and the reduce's root from `hemi`: a half
    // that spans several ranks needs the world they are numbered in, or nothing would be broadcast (round-5 advisor)
    THX_REQUIRE(!hemi || thx_comm_size(hemi) <= 1 || (world && thx_comm_size(world) > 1),
                "a hemisphere communicator of more than one rank needs the world communicator as well");
    thx_refine* h = new thx_refine;
    h->cfg = c;
    h->cfg.symMat = nullptr; h->cfg.symQuat = nullptr;   // (copied below; the caller's arrays are not kept)
    h->hemi = hemi;
    h->world = world;
    h->N = c.N; h->pf = c.pf; h->P = c.N * c.pf; h->nc = c.N / 2 + 1;
    h->rU = h->rE = c.N / 2 - 2;   // the cut-offs until thx_refine_set_cutoff says otherwise: Nyquist, _size = _N
    h->rCap = c.N / 2 - 1;         // Optimiser::maxR(): the largest cut-off a caller can ask for; the buffers are sized for its lists
    h->rSig = c.N / 2 - 1;
    h->nImg = c.nImg;
    h->nK = c.nK;
    h->nSym = c.nSym;
    h->searchType = c.searchType;
    memset(h->clsCountHost, 0, sizeof(h->clsCountHost));
    memset(h->lastRoundsK, 0, sizeof(h->lastRoundsK));
    for (int k = 0; k < 16; k++) h->balanced[k] = -1;
    if (c.halfOfRank < 0) {   // both halves on this rank: [0, nHalfA) is half 0, the rest half 1
        if (!(c.nHalfA >= 0 && c.nHalfA <= c.nImg)) { delete h; set_error("nHalfA out of range"); return -1; }
        h->nV = 2;
        h->halves[0] = 0; h->halves[1] = 1;
        h->lo[0] = 0; h->hi[0] = c.nHalfA; h->lo[1] = c.nHalfA; h->hi[1] = c.nImg;
    } else {
        h->nV = 1;
        h->halves[0] = c.halfOfRank; h->halves[1] = -1;
        h->lo[0] = 0; h->hi[0] = c.nImg; h->lo[1] = h->hi[1] = 0;
    }
    int nmax = 0;
    for (int v = 0; v < h->nV; v++) nmax = std::max(nmax, h->hi[v] - h->lo[v]);
    // balanced batches of at most cfg.batch images (a short last batch leaves the chip half empty)
    const int nb = std::max(1, (nmax + c.batch - 1) / c.batch);
    h->batch = std::min(65535, std::max(1, (nmax + nb - 1) / nb));
#define RC_OR_FREE(expr) do { int _rc = (expr); if (_rc) { thx_refine_destroy(h); return _rc; } } while (0)
    // the pixel lists live in device arrays sized for the largest cut-off; apply_cutoff (below, and thx_refine_set_cutoff) fills them
    h->capPxl = pixel_list_host(c.N, h->rCap, c.rL).nPxl;
    h->capPxlM = pixel_list_host(c.N, h->rCap, 0).nPxl;
    h->nPxl = h->capPxl; h->nPxlM = h->capPxlM;     // (row buffers below are allocated through these two)
    RC_OR_FREE(dalloc(h, &h->iCol, (size_t)h->capPxl)); RC_OR_FREE(dalloc(h, &h->iRow, (size_t)h->capPxl));
    RC_OR_FREE(dalloc(h, &h->iPxl, (size_t)h->capPxl)); RC_OR_FREE(dalloc(h, &h->iSig, (size_t)h->capPxl));
    RC_OR_FREE(dalloc(h, &h->iColM, (size_t)h->capPxlM)); RC_OR_FREE(dalloc(h, &h->iRowM, (size_t)h->capPxlM)); RC_OR_FREE(dalloc(h, &h->iPxlM, (size_t)h->capPxlM));
    const size_t n = c.nImg, imgSize = (size_t)c.N * h->nc * 2, volN = vol_n(h), mapN = (size_t)c.N * c.N * c.N;
    const size_t nVol = (size_t)h->nV * h->nK;
    RC_OR_FREE(dalloc(h, &h->img, n * imgSize));
    RC_OR_FREE(dalloc(h, &h->attr, n));
    RC_OR_FREE(dalloc(h, &h->gid0, n));
    RC_OR_FREE(dalloc(h, &h->datM, n * h->nPxlM * 2)); RC_OR_FREE(dalloc(h, &h->ctfM, n * h->nPxlM));
    RC_OR_FREE(dalloc(h, &h->datP, n * h->nPxl * 2)); RC_OR_FREE(dalloc(h, &h->ctfP, n * h->nPxl));
    RC_OR_FREE(dalloc(h, &h->sigRcpP, n * h->nPxl)); RC_OR_FREE(dalloc(h, &h->w, n));
    RC_OR_FREE(dalloc(h, &h->offset, n * 2));
    RC_OR_FREE(dalloc(h, &h->r, n * c.mLR * 4)); RC_OR_FREE(dalloc(h, &h->t, n * c.mLT * 2));
    RC_OR_FREE(dalloc(h, &h->r0, n * c.mLR * 4)); RC_OR_FREE(dalloc(h, &h->t0, n * c.mLT * 2));
    RC_OR_FREE(dalloc(h, &h->wR, n * c.mLR)); RC_OR_FREE(dalloc(h, &h->wT, n * c.mLT));
    RC_OR_FREE(dalloc(h, &h->k123, n * 3)); RC_OR_FREE(dalloc(h, &h->s01, n * 2));
    RC_OR_FREE(dalloc(h, &h->topR, n * 4)); RC_OR_FREE(dalloc(h, &h->topT, n * 2));
    RC_OR_FREE(dalloc(h, &h->active, n)); RC_OR_FREE(dalloc(h, &h->nP, n)); RC_OR_FREE(dalloc(h, &h->nActiveDev, (size_t)1));
    RC_OR_FREE(dalloc(h, &h->stopState, n * 8));
    RC_OR_FREE(dalloc(h, &h->cls, n)); RC_OR_FREE(dalloc(h, &h->cls0, n)); RC_OR_FREE(dalloc(h, &h->clsCount, (size_t)16));
    if (hipMemset(h->cls, 0, n * sizeof(int)) != hipSuccess || hipMemset(h->cls0, 0, n * sizeof(int)) != hipSuccess) {
        set_error("refine driver: hipMemset failed"); thx_refine_destroy(h); return -1;
    }
    if (h->nK > 1) RC_OR_FREE(dalloc(h, &h->clsD, (size_t)nmax * c.mReco));
    RC_OR_FREE(dalloc(h, &h->refRL, (size_t)h->nK * mapN));
    RC_OR_FREE(dalloc(h, &h->vols, nVol * volN * 2));
    {
        void* q = nullptr;   // (byte-sized: K cell-packed references of a 512^3 box exceed 2^32 floats)
        hipError_t e = hipMalloc(&q, nVol * thx_projector_packed_bytes(h->P));
        if (e != hipSuccess) { set_error("hipMalloc of %zu cell-packed references failed: %s", nVol, hipGetErrorString(e)); thx_refine_destroy(h); return (int)e; }
        h->owned.push_back(q);
        h->cells = reinterpret_cast<float*>(q);
    }
    RC_OR_FREE(dalloc(h, &h->F, nVol * volN * 2)); RC_OR_FREE(dalloc(h, &h->T, nVol * volN));
    RC_OR_FREE(dalloc(h, &h->maps, 2 * h->nK * mapN)); RC_OR_FREE(dalloc(h, &h->mapsX, 2 * h->nK * mapN));
    if (hipMemset(h->maps, 0, 2 * h->nK * mapN * sizeof(float)) != hipSuccess ||
        hipMemset(h->mapsX, 0, 2 * h->nK * mapN * sizeof(float)) != hipSuccess) {
        set_error("refine driver: hipMemset of the half maps failed"); thx_refine_destroy(h); return -1;
    }
    RC_OR_FREE(dalloc(h, &h->ftA, imgSize * c.N)); RC_OR_FREE(dalloc(h, &h->ftB, imgSize * c.N));
    RC_OR_FREE(dalloc(h, &h->fscDev, (size_t)c.N / 2));
    RC_OR_FREE(dalloc(h, &h->sig, (size_t)h->nV * c.nGroup * h->rSig)); RC_OR_FREE(dalloc(h, &h->sigRcp, (size_t)h->nV * c.nGroup * h->rSig));
    RC_OR_FREE(dalloc(h, &h->acc, 3 * (size_t)c.nGroup * (h->rSig + 1)));
    RC_OR_FREE(dalloc(h, &h->t0Dev, (size_t)16));
    const size_t B = h->batch;
    const int nDmax = std::max(1, c.mLD);
    RC_OR_FREE(dalloc(h, &h->rotB, B * c.mLR * 9)); RC_OR_FREE(dalloc(h, &h->pD, B));
    RC_OR_FREE(dalloc(h, &h->uR, B * c.mLR)); RC_OR_FREE(dalloc(h, &h->uT, B * c.mLT));
    RC_OR_FREE(dalloc(h, &h->wC, B)); RC_OR_FREE(dalloc(h, &h->wD, B * nDmax)); RC_OR_FREE(dalloc(h, &h->baseL, B));
    RC_OR_FREE(dalloc(h, &h->recoRot, (size_t)nmax * c.mReco * 9)); RC_OR_FREE(dalloc(h, &h->recoTran, (size_t)nmax * c.mReco * 2));
    RC_OR_FREE(dalloc(h, &h->rotTop, (size_t)nmax * 9)); RC_OR_FREE(dalloc(h, &h->tranTop, (size_t)nmax * 2));
    RC_OR_FREE(dalloc(h, &h->spec, (size_t)nmax * 4 * h->rSig));
    {
        char* ws = nullptr;
        RC_OR_FREE(dalloc(h, &ws, thx_expect_local_workspace((int)B, c.mLR, c.mLT, nDmax)));
        h->wsExpect = ws;
        char* wr = nullptr;
        RC_OR_FREE(dalloc(h, &wr, hemi ? thx_reco_allreduce_acc_workspace(h->P, h->rCap, c.pf) : 16));
        h->wsReduce = wr;
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, thx_insert_acc_bytes(h->P, h->nK));
        if (e != hipSuccess) { set_error("hipMalloc of the fixed-point accumulators failed: %s", hipGetErrorString(e)); thx_refine_destroy(h); return (int)e; }
        h->owned.push_back(q);
        h->accInt = q;
        RC_OR_FREE(dalloc(h, &h->bounds, n * 2));
        RC_OR_FREE(dalloc(h, &h->gexp, (size_t)2));
    }
    if (h->nSym > 0) {   // Symmetry::quat / the R matrices of the point group; SYMMETRIZE_FT's destination
        h->symMatHost.assign(c.symMat, c.symMat + (size_t)9 * c.nSym);
        std::vector<double> q(c.symQuat, c.symQuat + (size_t)4 * c.nSym);
        RC_OR_FREE(upload(h, &h->symQ, q));
        RC_OR_FREE(dalloc(h, &h->symTmp, volN * 2));
    }
    if (scans) {   // the scan's pixel list and one batch of its rows, weights and workspace
        PixelList plS = pixel_list_host(c.N, c.rScan, c.rL);   // capacity: the scan never runs beyond cfg.rScan (its radius = min(rScan, r))
        h->capPxlS = h->nPxlS = plS.nPxl;
        RC_OR_FREE(dalloc(h, &h->iColS, (size_t)plS.nPxl)); RC_OR_FREE(dalloc(h, &h->iRowS, (size_t)plS.nPxl));
        RC_OR_FREE(dalloc(h, &h->iPxlS, (size_t)plS.nPxl)); RC_OR_FREE(dalloc(h, &h->iSigS, (size_t)plS.nPxl));
        const int sb0 = c.scanBatch > 0 ? c.scanBatch : 2048;
        const int nsb = std::max(1, (nmax + sb0 - 1) / sb0);
        h->scanBatch = std::min(65535, std::max(1, (nmax + nsb - 1) / nsb));
        const size_t SB = h->scanBatch, nS = plS.nPxl;
        RC_OR_FREE(dalloc(h, &h->gridR, (size_t)c.nR * 4)); RC_OR_FREE(dalloc(h, &h->gridT, (size_t)c.nT * 2));
        RC_OR_FREE(dalloc(h, &h->mats, (size_t)c.nR * 9)); RC_OR_FREE(dalloc(h, &h->traS, (size_t)c.nT * nS * 2));
        RC_OR_FREE(dalloc(h, &h->rotP, (size_t)c.nR * nS * 2));
        RC_OR_FREE(dalloc(h, &h->datS, SB * nS * 2)); RC_OR_FREE(dalloc(h, &h->ctfS, SB * nS)); RC_OR_FREE(dalloc(h, &h->sigS, SB * nS));
        RC_OR_FREE(dalloc(h, &h->pRs, SB * c.nR)); RC_OR_FREE(dalloc(h, &h->pTs, SB * c.nT));
        RC_OR_FREE(dalloc(h, &h->uC, SB * h->nK)); RC_OR_FREE(dalloc(h, &h->uRs, (size_t)h->nK * SB * c.nR));
        RC_OR_FREE(dalloc(h, &h->uTs, (size_t)h->nK * SB * c.nT)); RC_OR_FREE(dalloc(h, &h->baseS, SB));
        unsigned char* q = nullptr;
        RC_OR_FREE(dalloc(h, &q, thx_expect_global_workspace((int)SB, c.nR, c.nT)));
        h->wsGlobal = q;
        // uniform priors of the scanned grid (Particle::reset(k, nR, nT, 1): every support point 1 / n)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(SB * c.nR)), dim3(256), 0, nullptr, h->pRs, 1.0 / c.nR, SB * c.nR);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(SB * c.nT)), dim3(256), 0, nullptr, h->pTs, 1.0 / c.nT, SB * c.nT);
    }
    if (c.mLD > 0) {   // CTF search: defocus factors of every image, the pre-calculated rows of allocPreCal's ctf = true branch
        RC_OR_FREE(dalloc(h, &h->dD, n * c.mLD)); RC_OR_FREE(dalloc(h, &h->wDD, n * c.mLD));
        RC_OR_FREE(dalloc(h, &h->sD, n)); RC_OR_FREE(dalloc(h, &h->topD, n));
        RC_OR_FREE(dalloc(h, &h->recoD, (size_t)nmax * c.mReco));
        RC_OR_FREE(dalloc(h, &h->freqD, (size_t)h->nPxl)); RC_OR_FREE(dalloc(h, &h->defD, n * h->nPxl));
        RC_OR_FREE(dalloc(h, &h->k1D, n)); RC_OR_FREE(dalloc(h, &h->k2D, n));
        RC_OR_FREE(dalloc(h, &h->ctfD, B * c.mLD * h->nPxl)); RC_OR_FREE(dalloc(h, &h->uD, B * c.mLD));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(n)), dim3(256), 0, nullptr, h->topD, 1.0, n);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(n * c.mLD)), dim3(256), 0, nullptr, h->dD, 1.0, n * c.mLD);
    }
    RC_OR_FREE(apply_cutoff(h, h->rE, h->rU, nullptr));   // the lists at Nyquist, the reconstruction plans at _size = _N
    if (hipHostMalloc(reinterpret_cast<void**>(&h->recoRes), 2 * 2 * 16 * 8 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        set_error("refine driver: hipHostMalloc failed"); h->recoRes = nullptr; thx_refine_destroy(h); return -1;
    }
    memset(h->recoRes, 0, 2 * 2 * 16 * 8 * sizeof(int));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(B)), dim3(256), 0, nullptr, h->pD, 1.0, B);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<float>), dim3(blocks_for(n)), dim3(256), 0, nullptr, h->w, 1.0f / c.mReco, n);
    if (hipDeviceSynchronize() != hipSuccess) { set_error("refine driver: device error during create"); thx_refine_destroy(h); return -1; }
#undef RC_OR_FREE
    *out = h;
    return 0;
}

int thx_refine_set_particles(thx_refine* h, const float* imgOri, const thx_ctf_attr* attr, const int* groupID_host,
                             const double* quat0, const double* tran0, void* stream)
{
    THX_REQUIRE(h && imgOri && attr && groupID_host && quat0 && tran0, "NULL argument");
    hipStream_t st = as_stream(stream);
    const thx_refine_config& c = h->cfg;
    const size_t n = h->nImg;
    h->imgOri = imgOri;
    h->gidHost.assign(groupID_host, groupID_host + n);
    std::vector<int> g0(n);
    for (size_t l = 0; l < n; l++) {
        THX_REQUIRE(groupID_host[l] >= 1 && groupID_host[l] <= c.nGroup, "groupID out of range (1-based)");
        g0[l] = groupID_host[l] - 1;
    }
    THX_CHECK(hipMemcpyAsync(h->gid0, g0.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
    THX_CHECK(hipMemcpyAsync(h->attr, attr, n * sizeof(thx_ctf_attr), hipMemcpyDeviceToDevice, st));
    THX_CHECK(hipMemcpyAsync(h->r0, quat0, n * c.mLR * 4 * sizeof(double), hipMemcpyDeviceToDevice, st));
    THX_CHECK(hipMemcpyAsync(h->t0, tran0, n * c.mLT * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    h->haveParticles = true;
    THX_RC(recut_rows(h, st));
    {   // images of this rank's half over all of its ranks (the 64-bit sums' head-room, thx_insert_scale_dev)
        double* cnt = reinterpret_cast<double*>(scratch(st, 7, sizeof(double)));
        THX_REQUIRE(cnt, "device scratch allocation failed");
        const double mine = (double)(h->nV == 2 ? std::max(h->hi[0] - h->lo[0], h->hi[1] - h->lo[1]) : h->nImg);
        THX_CHECK(hipMemcpyAsync(cnt, &mine, sizeof(double), hipMemcpyHostToDevice, st));
        THX_CHECK(hipStreamSynchronize(st));
        THX_RC(thx_comm_allreduce_f64(h->hemi, cnt, 1, st));
        double tot = mine;
        THX_CHECK(hipMemcpyAsync(&tot, cnt, sizeof(double), hipMemcpyDeviceToHost, st));
        THX_CHECK(hipStreamSynchronize(st));
        h->nImgHemi = (long)tot;
    }
    if (h->worldCount.empty()) {   // particles of every rank: an image's index over all ranks numbers its Philox streams, and
                                   // normCorrection's median runs over the norms of every rank's particles
        const int ws = h->world ? thx_comm_size(h->world) : 1, wr = h->world ? thx_comm_rank(h->world) : 0;
        std::vector<int> cnt(ws, 0);
        cnt[wr] = h->nImg;
        if (ws > 1) {
            int* d = reinterpret_cast<int*>(scratch(st, 7, (size_t)ws * sizeof(int)));
            THX_REQUIRE(d, "device scratch allocation failed");
            THX_CHECK(hipMemcpyAsync(d, cnt.data(), (size_t)ws * sizeof(int), hipMemcpyHostToDevice, st));
            THX_CHECK(hipStreamSynchronize(st));
            THX_RC(thx_comm_allreduce_i32(h->world, d, (size_t)ws, st));
            THX_CHECK(hipMemcpyAsync(cnt.data(), d, (size_t)ws * sizeof(int), hipMemcpyDeviceToHost, st));
            THX_CHECK(hipStreamSynchronize(st));
        }
        if (ws > 1) {   // determineBalanceClass and the FSC's random phases are evaluated on every rank from the SAME Philox streams
                        // (the reference draws on the master and broadcasts, src/Optimiser.cpp:5518-5584): the seed must not depend on the rank
            double* d = reinterpret_cast<double*>(scratch(st, 7, 4 * sizeof(double)));
            THX_REQUIRE(d, "device scratch allocation failed");
            const double lo = (double)(c.seed & 0xffffffffull), hi = (double)(c.seed >> 32);
            double v[4] = {lo, -lo, hi, -hi};
            THX_CHECK(hipMemcpyAsync(d, v, sizeof(v), hipMemcpyHostToDevice, st));
            THX_CHECK(hipStreamSynchronize(st));
            THX_RC(thx_comm_allreduce_max_f64(h->world, d, 4, st));
            THX_CHECK(hipMemcpyAsync(v, d, sizeof(v), hipMemcpyDeviceToHost, st));
            THX_CHECK(hipStreamSynchronize(st));
            THX_REQUIRE(v[0] == lo && v[1] == -lo && v[2] == hi && v[3] == -hi,
                        "thx_refine_config.seed differs between the ranks: it must be one number for the whole job (images are told apart by their index over all ranks)");
        }
        h->nImgWorld = 0; h->worldOffset = 0;
        for (int r = 0; r < ws; r++) { if (r < wr) h->worldOffset += cnt[r]; h->nImgWorld += cnt[r]; h->worldCount.push_back(cnt[r]); }
        if (c.normCorrection) {
            THX_RC(dalloc(h, &h->norm, (size_t)h->nImg));
            THX_RC(dalloc(h, &h->normAll, (size_t)h->nImgWorld + 1));
        }
    }
    THX_CHECK(hipStreamSynchronize(st));   // g0 is a host temporary
    return 0;
}

int thx_refine_set_image_base(thx_refine* h, long long base)
{
    THX_REQUIRE(h && base >= -1 && base + h->nImg <= 0xffffffffll, "bad arguments");
    h->imgBase = (long)base;
    return 0;
}

int thx_refine_set_reference(thx_refine* h, const float* refRL, void* stream)
{
    THX_REQUIRE(h && refRL, "NULL argument");
    THX_CHECK(hipMemcpyAsync(h->refRL, refRL, (size_t)h->nK * h->N * h->N * h->N * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}

int thx_refine_set_classes(thx_refine* h, const int* cls, void* stream)
{
    THX_REQUIRE(h && cls, "NULL argument");
    THX_CHECK(hipMemcpyAsync(h->cls, cls, (size_t)h->nImg * sizeof(int), hipMemcpyDefault, as_stream(stream)));
    THX_CHECK(hipMemcpyAsync(h->cls0, h->cls, (size_t)h->nImg * sizeof(int), hipMemcpyDeviceToDevice, as_stream(stream)));
    THX_CHECK(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

int thx_refine_set_grid(thx_refine* h, const double* quat, const double* shifts, void* stream)
{
    THX_REQUIRE(h && quat && shifts, "NULL argument");
    THX_REQUIRE(h->gridR, "the handle was created without a scan grid (cfg.nR = 0)");
    hipStream_t st = as_stream(stream);
    const thx_refine_config& c = h->cfg;
    THX_CHECK(hipMemcpyAsync(h->gridR, quat, (size_t)c.nR * 4 * sizeof(double), hipMemcpyDefault, st));
    THX_CHECK(hipMemcpyAsync(h->gridT, shifts, (size_t)c.nT * 2 * sizeof(double), hipMemcpyDefault, st));
    // Particle::reset ends with symmetrise() (src/Particle.cpp:168): the drawn rotations move next to ANCHOR_POINT_2
    THX_RC(thx_pf_symmetrise_dev(h->gridR, nullptr, 1, c.nR, h->symQ, h->nSym, st));
    THX_RC(thx_rotmat_dev(h->gridR, h->mats, c.nR, st));
    THX_RC(thx_translate_dev(h->traS, h->gridT, c.nT, h->iColS, h->iRowS, h->nPxlS, h->N, st));
    THX_CHECK(hipStreamSynchronize(st));
    h->haveGrid = true;
    return 0;
}

int thx_refine_set_cutoff(thx_refine* h, int r, int rU, void* stream)
{
    THX_REQUIRE(h, "NULL handle");
    return apply_cutoff(h, r, rU, as_stream(stream));
}

int thx_refine_get_cutoff(const thx_refine* h, int* r, int* rU, int* size, int* rScan)
{
    THX_REQUIRE(h, "NULL handle");
    if (r) *r = h->rE;
    if (rU) *rU = h->rU;
    if (size) *size = h->size;
    if (rScan) *rScan = h->rS;
    return 0;
}

int thx_refine_set_search_type(thx_refine* h, int searchType)
{
    THX_REQUIRE(h, "NULL handle");
    THX_REQUIRE(searchType == THX_SEARCH_LOCAL || (searchType == THX_SEARCH_GLOBAL && h->gridR) || (searchType == THX_SEARCH_CTF && h->cfg.mLD > 0),
                "search type not available: a global search needs cfg.nR / nT / rScan, a CTF search cfg.mLD, at create");
    h->searchType = searchType;
    return 0;
}

// the state before the first iteration: initial references, no re-centring offset, masked copies of the images as read
// (Optimiser::initImg masks them on load), flat initial noise model, initial support points (Particle::load)
int thx_refine_reset(thx_refine* h, void* stream)
{
    THX_REQUIRE(h && h->imgOri, "thx_refine_set_particles has not been called");
    hipStream_t st = as_stream(stream);
    const thx_refine_config& c = h->cfg;
    const size_t n = h->nImg, imgSize = (size_t)h->N * h->nc * 2, mapN = (size_t)h->N * h->N * h->N;
    for (int v = 0; v < h->nV; v++)
        for (int k = 0; k < h->nK; k++) THX_RC(refresh_projector(h, v, k, h->refRL + (size_t)k * mapN, st));
    THX_CHECK(hipMemsetAsync(h->offset, 0, n * 2 * sizeof(double), st));
    THX_CHECK(hipMemcpyAsync(h->img, h->imgOri, n * imgSize * sizeof(float), hipMemcpyDeviceToDevice, st));
    THX_RC(thx_remask_dev(h->img, (int)n, h->N, c.maskRadiusPx, 6.0f, st));
    const size_t nsig = (size_t)h->nV * c.nGroup * h->rSig;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<float>), dim3(blocks_for(nsig)), dim3(256), 0, st, h->sig, c.sigma2Init, nsig);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<float>), dim3(blocks_for(nsig)), dim3(256), 0, st, h->sigRcp, -0.5f / c.sigma2Init, nsig);
    THX_CHECK(hipMemcpyAsync(h->r, h->r0, n * c.mLR * 4 * sizeof(double), hipMemcpyDeviceToDevice, st));
    THX_CHECK(hipMemcpyAsync(h->t, h->t0, n * c.mLT * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(n * c.mLR)), dim3(256), 0, st, h->wR, 1.0 / c.mLR, n * c.mLR);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(n * c.mLT)), dim3(256), 0, st, h->wT, 1.0 / c.mLT, n * c.mLT);
    // the classes the particles were loaded with and the defocus search's state (Particle::load: d = 1)
    THX_CHECK(hipMemcpyAsync(h->cls, h->cls0, n * sizeof(int), hipMemcpyDeviceToDevice, st));
    if (c.mLD > 0) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(n)), dim3(256), 0, st, h->topD, 1.0, n);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fill<double>), dim3(blocks_for(n * c.mLD)), dim3(256), 0, st, h->dD, 1.0, n * c.mLD);
        THX_CHECK(hipMemsetAsync(h->sD, 0, n * sizeof(double), st));
        THX_CHECK(hipMemsetAsync(h->wDD, 0, n * c.mLD * sizeof(double), st));
    }
    THX_LAUNCH_CHECK();
    h->iterCount = 0;
    // Particle::load -> calVari: ACG concentration of the rotations (with a point group: of their counterparts next to a random
    // one of them), per-column sd of the shifts
    for (size_t b0 = 0; b0 < n; b0 += 65535) {
        const int nb = (int)std::min<size_t>(65535, n - b0);
        const thx_pf_ctx ctx = pf_ctx(h, (int)b0);
        THX_RC(thx_pf_cal_vari_dev(h->r + b0 * c.mLR * 4, h->t + b0 * c.mLT * 2, h->k123 + b0 * 3, h->s01 + b0 * 2, nb, c.mLR, c.mLT, c.seed,
                                   (unsigned)SLOT_RESET, &ctx, st));
    }
    hipLaunchKernelGGL(k_take_first, dim3(blocks_for(n * 4)), dim3(256), 0, st, h->topR, h->r, (int)n, c.mLR * 4, 4);
    hipLaunchKernelGGL(k_take_first, dim3(blocks_for(n * 2)), dim3(256), 0, st, h->topT, h->t, (int)n, c.mLT * 2, 2);
    THX_LAUNCH_CHECK();
    h->fscReco.assign((size_t)h->nK * (h->N / 2), 1.0f);   // Model::initProjReco: _reco[l]->setFSC(vec::Constant(_rU, 1)), src/Model.cpp:1086
    h->fscRecoN = h->rU;
    for (int v = 0; v < h->nV; v++) THX_RC(refresh_rows(h, v, st));
    return 0;
}

int thx_refine_iterate(thx_refine* h, float* fscHost, int timed, void* stream)
{
    THX_REQUIRE(h && h->imgOri, "thx_refine_set_particles has not been called");
    hipStream_t st = as_stream(stream);
    const thx_refine_config& c = h->cfg;
    const size_t mapN = (size_t)h->N * h->N * h->N, volN = volF_n(h);   // (volN: one F / T volume on the reconstructors' grid PF)
    const size_t imgSize = (size_t)h->N * h->nc * 2;
    const int K = h->nK;
    const bool global = h->searchType == THX_SEARCH_GLOBAL;
    h->timed = timed != 0;
    if (h->events.size() > 8192) THX_RC(resolve_events(h));   // (a caller that never asks for the statistics must not pile events up)
    // ---- E and M per local half: rows -> [scan] -> expectation -> sigma update -> draws + insertion ----
    // With normCorrection the M-step starts with a statistic over ALL particles (src/Optimiser.cpp:3405-3413), so every local
    // half's expectation runs first; without it each half goes through E and M in turn.  (The Philox numbering depends on the
    // iteration and the phase only, so both orders draw the same numbers.)
    const bool normOn = c.normCorrection != 0;
    h->lastNormMedian = 0.f; h->lastNormRadius = 0.f;
    auto e_step = [&](int vi) -> int {
        { Scope s(h, st, EV_STAGE0 + ST_ROWS); THX_RC(refresh_rows(h, vi, st)); }
        if (global) { Scope s(h, st, EV_STAGE0 + ST_SCAN); THX_RC(scan_and_select(h, vi, st)); }
        { Scope s(h, st, EV_STAGE0 + ST_EXPECT); THX_RC(expectation(h, vi, st)); }
        return 0;
    };
    if (normOn) {
        for (int vi = 0; vi < h->nV; vi++) THX_RC(e_step(vi));
        if (h->iterCount != 0 && !global) { Scope s(h, st, EV_STAGE0 + ST_NORM); THX_RC(norm_correction(h, st)); }   // (_iter != 0) && not global
    }
    for (int vi = 0; vi < h->nV; vi++) {
        if (!normOn) THX_RC(e_step(vi));
        { Scope s(h, st, EV_STAGE0 + ST_SIGMA); THX_RC(sigma_update(h, vi, st)); }
        { Scope s(h, st, EV_STAGE0 + ST_INSERT); THX_RC(insertion(h, vi, st)); }
        if (h->cap.Fraw) THX_CHECK(hipMemcpyAsync(h->cap.Fraw + (size_t)vi * K * volN * 2, h->F + (size_t)vi * K * volN * 2, (size_t)K * volN * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (h->cap.Traw) THX_CHECK(hipMemcpyAsync(h->cap.Traw + (size_t)vi * K * volN, h->T + (size_t)vi * K * volN, (size_t)K * volN * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    // ---- Optimiser::reconstructRef after the insertion (src/Optimiser.cpp:7248-7760) and the run loop up to Model::resetReco
    // (:3900-4073): prepareTF; reconstruct with MAP off -> [balanceClass] -> compareTwoHemispheres(fsc) -> Model::_FSC; reconstruct
    // with MAP on and the reconstructor's OWN FSC (set by resetReco at the end of the previous iteration) -> [balanceClass] ->
    // compareTwoHemispheres(avg); solventFlatten; refreshProj; resetReco ----
    std::vector<float> fsc((size_t)K * (h->N / 2), 0.f);
    {
        Scope s(h, st, EV_STAGE0 + ST_RECO);
        const bool multi = h->world && thx_comm_size(h->world) > 1;
        const double symR = (double)h->rU * h->pf + 1;   // _maxRadius * _pf + 1, src/Reconstructor.cpp:2676-2690
        // class distribution of the iteration and, after a global search, which empty class takes over which reference
        int bm[16];
        for (int k = 0; k < 16; k++) bm[k] = -1;
        if (K > 1) {
            std::vector<double> distr;
            THX_RC(class_distribution(h, distr, st));
            if (global && c.balanceClass) determine_balance(h, distr, bm);
        } else {
            h->clsCountHost[0] = h->nImg;
        }
        for (int k = 0; k < 16; k++) h->balanced[k] = bm[k];
        const int Hh = h->hemi ? thx_comm_size(h->hemi) : 1;
        const bool owners = Hh > 1 && !knobs().recoReplicate;
        float t0[2][16];
        for (int vi = 0; vi < h->nV; vi++) {
            // T(0,0,0) of every class after the half-set reduce: a class no image of the half went to has nothing to normalise.
            // With owners only the reconstructing rank holds the sum: the others' partial sums are zeroed and the half's ranks add up
            hipLaunchKernelGGL(k_take_first_f32, dim3(1), dim3(64), 0, st, h->t0Dev, h->T + (size_t)vi * K * volN, K, volN);
            if (owners) {   // (counted once: from the rank of the class's MAP-off reconstruction)
                unsigned mask = 0;
                for (int k = 0; k < K; k++) mask |= owns_class(h, vi, k, 0) ? (1u << k) : 0u;
                hipLaunchKernelGGL(k_mask_owned, dim3(1), dim3(64), 0, st, h->t0Dev, K, mask);
            }
            THX_LAUNCH_CHECK();
            if (owners) THX_RC(thx_comm_allreduce_f32(h->hemi, h->t0Dev, (size_t)K, st));
            THX_CHECK(hipMemcpyAsync(t0[vi], h->t0Dev, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, st));
        }
        THX_CHECK(hipStreamSynchronize(st));
        memset(h->lastRoundsK, 0, sizeof(h->lastRoundsK));
        memset(h->recoRes, 0, 2 * 2 * 16 * 8 * sizeof(int));
        // the reconstructions are QUEUED (the gridding loop's stop rule runs on the device, thx_reco.hip): their round counts
        // land in page-locked memory and are read once the stream has been synchronised
        auto res_of = [&](int map, int vi, int k) { return h->recoRes + (((size_t)map * 2 + vi) * 16 + k) * 8; };
        for (int vi = 0; vi < h->nV; vi++)
            for (int k = 0; k < K; k++) {
                const bool off = owns_class(h, vi, k, 0), on = owns_class(h, vi, k, 1);
                if (!(t0[vi][k] > 0.f) || !(off || on)) continue;
                float* F = h->F + ((size_t)vi * K + k) * volN * 2;
                float* T = h->T + ((size_t)vi * K + k) * volN;
                // prepareTF, src/Reconstructor.cpp:1056-1091: [allReduceT: done on the integers] normalise T and F by 1 / T(0,0,0)
                // (:2455-2476), symmetrizeT, [allReduceF], symmetrizeF
                THX_RC(thx_normalise_tf_dev(F, T, h->PF, st));
                if (h->nSym > 0) {
                    THX_RC(thx_symmetrize_dev(h->symTmp, T, h->PF, 0, h->symMatHost.data(), h->nSym, symR, st));
                    THX_CHECK(hipMemcpyAsync(T, h->symTmp, volN * sizeof(float), hipMemcpyDeviceToDevice, st));
                    THX_RC(thx_symmetrize_dev(h->symTmp, F, h->PF, 1, h->symMatHost.data(), h->nSym, symR, st));
                    THX_CHECK(hipMemcpyAsync(F, h->symTmp, volN * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
                }
                if (h->cap.Fsym) THX_CHECK(hipMemcpyAsync(h->cap.Fsym + ((size_t)vi * K + k) * volN * 2, F, volN * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
                if (h->cap.Tsym) THX_CHECK(hipMemcpyAsync(h->cap.Tsym + ((size_t)vi * K + k) * volN, T, volN * sizeof(float), hipMemcpyDeviceToDevice, st));
                // setMAP(false); setJoinHalf(true) (OPTIMISER_RECONSTRUCT_JOIN_HALF); setGridCorr(true) (OPTIMISER_3D_GRID_CORR), :7326-7352
                if (off)
                    THX_RC(thx_reco_reconstruct_async_dev(h->plans[vi], F, T, h->rU, nullptr, 0, 1, 0, 1, h->maps + ((size_t)h->halves[vi] * K + k) * mapN,
                                                          res_of(0, vi, k), st));
                else   // (the MAP-off pass runs on another rank: its one side effect on T, the 1e-25 floor, is applied here)
                    THX_RC(thx_reco_floor_T_dev(h->plans[vi], T, h->rU, st));
                // setMAP(true); setJoinHalf(true); setGridCorr(true), :7574-7600; Reconstructor::_FSC is LAST iteration's (Model::resetReco),
                // so this pass does not wait for the FSC of the maps above: it is queued right behind them -- on another rank of
                // the half where there is one to spare
                if (on)
                    THX_RC(thx_reco_reconstruct_async_dev(h->plans[vi], F, T, h->rU, h->fscReco.data() + (size_t)k * (h->N / 2), h->fscRecoN, 1, 1, 1,
                                                          h->mapsX + ((size_t)h->halves[vi] * K + k) * mapN, res_of(1, vi, k), st));
            }
        // every rank ends up with every class's map of both halves (the reference sends them to the master, src/Model.cpp:375-391):
        // class k of half hf is broadcast from the world rank that reconstructed it -- rank 2 x (its number inside the half) + hf
        // (rank r owns half r mod 2, checked in thx_refine_create)
        auto exchange = [&](float* m, int which) -> int {
            if (!multi) return 0;
            for (int hf = 0; hf < 2; hf++)
                for (int k = 0; k < K; k++)
                    THX_RC(thx_comm_broadcast(h->world, m + ((size_t)hf * K + k) * mapN, mapN * sizeof(float), 2 * owner_in_half(h, hf, k, which) + hf, st));
            return 0;
        };
        // balanceClass(bm), :5586-5593, 7510-7523: _model.ref(t) = _model.ref(j).copyVolume(), for every half whose maps are here
        auto balance = [&](float* m) -> int {
            for (int hf = 0; hf < 2; hf++) {
                if (!multi && !(h->halves[0] == hf || (h->nV > 1 && h->halves[1] == hf))) continue;
                for (int t = 0; t < K; t++)
                    if (bm[t] >= 0 && bm[t] != t)
                        THX_CHECK(hipMemcpyAsync(m + ((size_t)hf * K + t) * mapN, m + ((size_t)hf * K + bm[t]) * mapN, mapN * sizeof(float),
                                                 hipMemcpyDeviceToDevice, st));
            }
            return 0;
        };
        THX_RC(exchange(h->maps, 0));
        THX_RC(balance(h->maps));
        if (h->cap.mapsFsc) THX_CHECK(hipMemcpyAsync(h->cap.mapsFsc, h->maps, 2 * (size_t)K * mapN * sizeof(float), hipMemcpyDeviceToDevice, st));
        // compareTwoHemispheres(true, false, ...), src/Optimiser.cpp:7547: FSC over _rU shells, core-mask corrected on request
        const float coreR = c.coreFSC ? (float)(int)rint((double)(c.maskRadiusPx)) : 0.f;   // AROUND(maskRadius / pixelSize), :188
        for (int k = 0; k < K; k++) {
            THX_RC(thx_fft3d_fw_dev(h->maps + (size_t)k * mapN, h->ftA, h->N, st));
            THX_RC(thx_fft3d_fw_dev(h->maps + ((size_t)K + k) * mapN, h->ftB, h->N, st));
            // random phases: Philox calls fscCall (half A) and fscCall + 1 (half B) -- a function of the iteration count and the
            // class only, so that every rank substitutes the same phases and arrives at the same curve
            const unsigned fscCall = 0x40000000u + 2u * (h->iterCount * 16u + (unsigned)k);
            THX_RC(thx_compare_hemispheres_dev(h->ftA, h->ftB, h->N, h->rU, fsc.data() + (size_t)k * (h->N / 2), nullptr, coreR,
                                               6.0f /* EDGE_WIDTH_RL */, 0, 0, c.seed, fscCall, nullptr, st));
        }
        THX_RC(exchange(h->mapsX, 1));
        THX_RC(balance(h->mapsX));   // :7727-7733
        if (c.goldenAverage) {
            // compareTwoHemispheres(false, true, AVERAGE_TWO_HEMISPHERE_THRES), :7747.  One class under the gold standard: A = B =
            // (A + B) / 2 inside r = Model::resolutionP(thres = 0.95, false) of the FSC just computed (MODEL_RESOLUTION_BASE_AVERAGE,
            // src/Model.cpp:616-674); several classes: everywhere (:688-696)
            int avgR = -1;
            if (K == 1 && c.goldenAverage == 1) {
                int res = 1;   // resP(_FSC.col(0), 0.95, 1, 1, false), src/Functions/Spectrum.cpp:339-363
                for (; res < h->rU; res++)
                    if (fsc[res] < 0.95f) break;
                avgR = res - 1;
            }
            for (int k = 0; k < K; k++) {
                THX_RC(thx_fft3d_fw_dev(h->mapsX + (size_t)k * mapN, h->ftA, h->N, st));
                THX_RC(thx_fft3d_fw_dev(h->mapsX + ((size_t)K + k) * mapN, h->ftB, h->N, st));
                THX_RC(thx_compare_hemispheres_dev(h->ftA, h->ftB, h->N, h->rU, nullptr, nullptr, 0.f, 6.0f, 1, avgR, c.seed, 0, nullptr, st));
                THX_RC(thx_fft3d_bw_dev(h->ftA, h->mapsX + (size_t)k * mapN, h->N, st));
                THX_RC(thx_fft3d_bw_dev(h->ftB, h->mapsX + ((size_t)K + k) * mapN, h->N, st));
            }
        }
        for (int vi = 0; vi < h->nV; vi++)
            for (int k = 0; k < K; k++) {
                // a class without images that nothing was handed to keeps its reference (the reference's own maps would be NaN:
                // sf = 1 / T(0,0,0) = inf; OPTIMISER_BALANCE_CLASS is what keeps it from getting there)
                if (!(t0[vi][k] > 0.f) && !(bm[k] >= 0 && bm[k] != k && t0[vi][bm[k]] > 0.f)) continue;
                float* m = h->mapsX + ((size_t)h->halves[vi] * K + k) * mapN;
                if (c.solventFlatten)   // softMask(ref, ref, maskRadius / pixelSize, EDGE_WIDTH_RL, 0), :7958-7975
                    THX_RC(thx_soft_mask_volume_dev(m, h->N, c.maskRadiusPx, 6.0f, 0.f, st));
                THX_RC(refresh_projector(h, vi, k, m, st));
            }
        // the round counts of the queued reconstructions (statistics only).  (The synchronisation also comes before fscReco
        // changes: the queued MAP reconstructions copy it from host memory.)
        THX_CHECK(hipStreamSynchronize(st));
        h->fscReco = fsc;             // Model::resetReco: _reco[l]->setFSC(_FSC.col(l)), src/Model.cpp:1122: this iteration's curve, _rU entries
        h->fscRecoN = h->rU;
        for (int map = 0; map < 2; map++)
            for (int vi = 0; vi < h->nV; vi++)
                for (int k = 0; k < K; k++) {
                    const int it = res_of(map, vi, k)[1];
                    h->recoRounds += it;
                    h->lastRoundsK[(map * 2 + vi) * 16 + k] = it;
                    if (k == 0) h->lastRounds[map * 2 + vi] = it;
                }
    }
    // ---- re-centre and re-mask the particle images with the top shift of the last phase; not after a global search
    // (OPTIMISER_RECENTRE_IMAGE_EACH_ITERATION: `if (_searchType != SEARCH_TYPE_GLOBAL)`, src/Optimiser.cpp:3790-3810) ----
    if (!global) {
        Scope s(h, st, EV_STAGE0 + ST_RECENTRE);
        for (int vi = 0; vi < h->nV; vi++) {
            const int lo = h->lo[vi], n = h->hi[vi] - lo;
            if (n <= 0) continue;
            THX_CHECK(hipMemcpyAsync(h->tranTop, h->topT + (size_t)lo * 2, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(k_recentre_state, dim3(n), dim3(64), 0, st, h->offset + (size_t)lo * 2, h->t + (size_t)lo * c.mLT * 2,
                               h->topT + (size_t)lo * 2, h->tranTop, n, c.mLT);
            THX_LAUNCH_CHECK();
            THX_RC(thx_translate_image_dev(h->img + (size_t)lo * imgSize, h->imgOri + (size_t)lo * imgSize, h->offset + (size_t)lo * 2, n,
                                           h->N, -1.0f, st));
            THX_RC(thx_remask_dev(h->img + (size_t)lo * imgSize, n, h->N, c.maskRadiusPx, 6.0f, st));
        }
    }
    h->iterations++;
    h->iterCount++;
    if (fscHost) memcpy(fscHost, fsc.data(), fsc.size() * sizeof(float));
    return 0;
}

int thx_refine_set_capture(thx_refine* h, const thx_refine_capture* capture)
{
    THX_REQUIRE(h, "NULL handle");
    if (capture) h->cap = *capture;
    else h->cap = thx_refine_capture{};
    return 0;
}

int thx_refine_get_map_k(thx_refine* h, int half, int k, float* dstRL, void* stream)
{
    THX_REQUIRE(h && dstRL && (half == 0 || half == 1) && k >= 0 && k < h->nK, "bad arguments");
    const size_t mapN = (size_t)h->N * h->N * h->N;
    THX_CHECK(hipMemcpyAsync(dstRL, h->mapsX + ((size_t)half * h->nK + k) * mapN, mapN * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}

int thx_refine_get_map(thx_refine* h, int half, float* dstRL, void* stream) { return thx_refine_get_map_k(h, half, 0, dstRL, stream); }

int thx_refine_get_state(thx_refine* h, double* offset, double* topR, double* topT, float* sig, void* stream)
{
    THX_REQUIRE(h, "NULL handle");
    hipStream_t st = as_stream(stream);
    const size_t n = h->nImg;
    if (offset) THX_CHECK(hipMemcpyAsync(offset, h->offset, n * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (topR) THX_CHECK(hipMemcpyAsync(topR, h->topR, n * 4 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (topT) THX_CHECK(hipMemcpyAsync(topT, h->topT, n * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (sig) THX_CHECK(hipMemcpyAsync(sig, h->sig, (size_t)h->nV * h->cfg.nGroup * h->rSig * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

int thx_refine_get_view(thx_refine* h, thx_refine_view* v)
{
    THX_REQUIRE(h && v, "NULL argument");
    memset(v, 0, sizeof(*v));
    v->nImg = h->nImg; v->nPxl = h->nPxl; v->nPxlM = h->nPxlM; v->nVol = h->nV * h->nK; v->vdim = h->P; v->rSig = h->rSig;
    v->fdim = h->PF;
    v->iCol = h->iCol; v->iRow = h->iRow; v->iPxl = h->iPxl; v->iSig = h->iSig; v->iColM = h->iColM; v->iRowM = h->iRowM;
    v->img = h->img; v->datP = h->datP; v->ctfP = h->ctfP; v->sigRcpP = h->sigRcpP; v->datM = h->datM; v->ctfM = h->ctfM;
    v->r = h->r; v->t = h->t; v->wR = h->wR; v->wT = h->wT; v->offset = h->offset;
    v->vols = h->vols; v->cells = h->cells; v->F = h->F; v->T = h->T; v->sig = h->sig;
    v->recoRot = h->recoRot; v->recoTran = h->recoTran;
    v->nP = h->nP;
    v->norm = h->norm;
    v->cls = h->cls; v->topR = h->topR; v->topT = h->topT; v->k123 = h->k123; v->s01 = h->s01;
    v->d = h->dD; v->wD = h->wDD;
    v->maps = h->maps; v->mapsMAP = h->mapsX;
    v->nK = h->nK; v->nPxlS = h->nPxlS;
    return 0;
}

int thx_refine_get_stats(thx_refine* h, thx_refine_stats* out, int reset)
{
    THX_REQUIRE(h && out, "NULL argument");
    THX_RC(resolve_events(h));
    memset(out, 0, sizeof(*out));
    out->expectMs = h->accExpect.ms; out->expectLaunches = h->accExpect.launches; out->expectImages = h->accExpect.images;
    out->insertMs = h->accInsert.ms; out->insertLaunches = h->accInsert.launches; out->insertImages = h->accInsert.images;
    out->scanMs = h->accScan.ms; out->scanLaunches = h->accScan.launches; out->scanImages = h->accScan.images;
    for (int i = 0; i < ST_COUNT; i++) out->stageMs[i] = h->stageMs[i];
    out->balancingRounds = h->recoRounds;
    out->iterations = h->iterations;
    out->imagePhases = h->imagePhases;
    out->nPxl = h->nPxl; out->nPxlM = h->nPxlM; out->batch = h->batch; out->nPxlS = h->nPxlS; out->nK = h->nK;
    for (int i = 0; i < 4; i++) out->lastRounds[i] = h->lastRounds[i];
    memcpy(out->lastRoundsK, h->lastRoundsK, sizeof(out->lastRoundsK));
    memcpy(out->classCount, h->clsCountHost, sizeof(out->classCount));
    memcpy(out->balanced, h->balanced, sizeof(out->balanced));
    out->normMedian = h->lastNormMedian; out->normRadius = h->lastNormRadius;
    {
        unsigned long long g = 0;
        THX_RC(thx_insert_groups_total(&g, reset, nullptr));   // cumulative on the device since the last reset
        out->insertGroups = g;
    }
    if (reset) {
        h->accExpect = thx_refine_stats_acc(); h->accInsert = thx_refine_stats_acc(); h->accScan = thx_refine_stats_acc();
        for (int i = 0; i < 8; i++) h->stageMs[i] = 0;
        h->recoRounds = 0; h->iterations = 0; h->imagePhases = 0;
    }
    return 0;
}

}  // extern "C"
