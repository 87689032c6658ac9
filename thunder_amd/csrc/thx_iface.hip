// thx_iface.hip -- the per-image, staged local-search entry points of gpu/interface/Interface.h:18-164
// (ExpectPreidx / Prefre / LocalIn / LocalV3D / LocalP / LocalHostA / LocalRTD / LocalPreI3D / LocalM / LocalHostF /
// LocalFin / FreeIdx) with the two helper classes they pass around (ManagedArrayTexture, ManagedCalPoint) as opaque
// handles.  Optimiser::expectationG (src/Optimiser.cpp:2180-3393) drives them image by image from OpenMP threads; every
// entry keeps the argument meaning of its twin, so a replacement Interface.cpp forwards one to one.
//
// The work itself is the fused kernel of thx_estep.hip on ONE image -- the slices are never materialised.  The reference's caller
// holds a per-GPU lock over ExpectLocalRTD .. ExpectLocalM (omp_set_lock(&mtx[gpuIdx]), src/Optimiser.cpp:2960-3080) and reads the
// weights right after ExpectLocalM returns: one image-phase is in flight per GPU and its LATENCY is what an unchanged Optimiser.cpp
// pays.  So the chain is cut to the minimum (round 6): ExpectLocalV3D also builds the cell-packed copy of the volume (one 64-byte
// request per sample); ExpectLocalRTD only stages -- priors, shifts, quaternions and the rotation matrices it forms on the host
// (rotate3d_colmajor: the device kernel's arithmetic, same bits) -- into a page-locked block the handle owns; ExpectLocalPreI3D sends
// that block with ONE copy (and, in a defocus search, runs the CTF-row kernel); ExpectLocalM launches the one-image form of the
// kernel (a workgroup per 32 pixels instead of 16 per image, class prior by value: expect_local_single), the fixed-order reduce of
// the partial sums and the finalise kernel, which writes wC / wR / wT / wD STRAIGHT into a page-locked host block and, behind a
// system-scope fence, a completion word the calling thread polls.  Per image-phase: 1 H2D, 3 kernels, 1 wait on a host word -- no copy
// back, no call into the runtime while waiting (before: 5 - 7 H2D, 3 kernels, 4 D2H, 2 - 3 synchronisations and 16 workgroups).  The
// batched thx_expect_local_dev stays the throughput form.
#include <chrono>
#include <map>
#include <mutex>

#include "thx_common.h"

using namespace thx;

struct thx_texture {      // ManagedArrayTexture (gpu/include/ManagedArrayTexture.h): a device-resident padded FT
    int mode, vdim, gpu;
    float* vol;           // [vdim][vdim][vdim/2+1] complex64
    float* cells;         // its cell-packed copy (thx_projector_pack_dev), built by ExpectLocalV3D; NULL if it did not fit
};

struct thx_calpoint {     // ManagedCalPoint (gpu/include/ManagedCalPoint.h): per-stream search buffers
    int mode, cSearch, gpu, nR, nT, mD, npxl;
    double *devR, *devT, *devD, *devC;      // priors oldR [nR], oldT [nT], oldD [mD or 1], oldC [1]
    double *devnR, *devnT, *devdP, *devRotm;  // quaternions [nR][4], shifts [nT][2], defocus factors [mD], matrices [nR][9]
    float *devwC, *devwR, *devwT, *devwD, *devBaseL;
    float* devctfD;                          // [mD][npxl] CTF rows of the defocus search
    thx_ctf_attr* attr;                      // one device CTFAttr (amplitude contrast, phase shift of the current image)
    float* k12;                              // device k1, k2 of the current image
    void* ws;
    hipStream_t stream;
    // page-locked host mirrors: hIn = the double block devR .. devRotm as ExpectLocalRTD stages it (ONE copy in ExpectLocalPreI3D),
    // hOut = wC | wR | wT | wD | baseLine | completion word: written by ExpectLocalM's finalise kernel itself (THX_IFACE_ZEROCOPY=0: ONE copy back
    // of devwC .. devBaseL), hSmall = CTFAttr + k1, k2 of the current image (defocus search)
    double* hIn;
    float* hOut;
    char* hSmall;
    size_t nIn, nOut;
    unsigned seq;                            // image-phases run on this ManagedCalPoint: the value the finalise kernel leaves in hOut[nOut]
    // recorded by ExpectLocalPreI3D for ExpectLocalM
    const float* vol;
    const float* cells;
    const int *iCol, *iRow;
    int pf, idim, vdim;
};

// The wait at the end of an image-phase.  hipStreamSynchronize may put the thread to sleep until an interrupt arrives (tens of
// microseconds to wake up -- as long as the kernels it waits for); the image-phase is ~100 us of device work and the caller holds a
// per-GPU lock over it, so the thread polls the stream for up to a millisecond first (THX_IFACE_SPIN=0: never).
static int wait_stream(hipStream_t st)
{
    static const bool spin = []() { const char* e = getenv("THX_IFACE_SPIN"); return !(e && e[0] == '0'); }();
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return 0;
            if (q != hipErrorNotReady) { set_error("stream error: %s", hipGetErrorString(q)); return (int)q; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) break;
        }
    }
    THX_CHECK(hipStreamSynchronize(st));
    return 0;
}

// The same wait on a word the last kernel of the image-phase writes into page-locked host memory (after a system-scope fence behind
// its outputs): the thread sees it without a call into the runtime and without waiting for the stream's completion signal.  Falls back
// to the stream after a millisecond -- or at once when a launch has failed (the word would never arrive).
static int wait_flag(const unsigned* flag, unsigned val, hipStream_t st)
{
    static const bool spin = []() { const char* e = getenv("THX_IFACE_SPIN"); return !(e && e[0] == '0'); }();
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; it++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == val) return 0;
            if ((it & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) break;
        }
    }
    THX_RC(wait_stream(st));
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != val) { set_error("the image-phase finished without its completion word"); return -1; }
    return 0;
}

// copy streams of ExpectLocalP's device slots, one per (device, slot), created on first use and kept for the life of the process
static int slot_stream(int gpu, int slot, hipStream_t* out)
{
    static std::mutex mtx;
    static std::map<std::pair<int, int>, hipStream_t> streams;
    std::lock_guard<std::mutex> g(mtx);
    auto it = streams.find({gpu, slot});
    if (it == streams.end()) {
        hipStream_t st = nullptr;
        THX_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        it = streams.emplace(std::make_pair(gpu, slot), st).first;
    }
    *out = it->second;
    return 0;
}

extern "C" {

/* ManagedArrayTexture::Init(mode, vdim, gpuIdx) */
int thx_texture_create(thx_texture** out, int mode, int vdim, int gpuIdx)
{
    THX_REQUIRE(out && vdim > 0, "bad arguments");
    THX_REQUIRE(mode == 1, "only MODE_3D (1) is implemented");
    THX_CHECK(hipSetDevice(gpuIdx));
    thx_texture* t = new thx_texture();
    t->mode = mode; t->vdim = vdim; t->gpu = gpuIdx; t->vol = nullptr; t->cells = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->vol), (size_t)vdim * vdim * (vdim / 2 + 1) * 2 * sizeof(float));
    if (e != hipSuccess) { delete t; set_error("hipMalloc failed: %s", hipGetErrorString(e)); return (int)e; }
    // the cell-packed copy is 8 x the volume (4.3 GB at 512^3 voxels of padded FT): taken when the device has it to spare (THX_IFACE_PACK=0: never)
    {
        size_t freeB = 0, totalB = 0;
        const size_t need = thx_projector_packed_bytes(vdim);
        const char* env = getenv("THX_IFACE_PACK");
        if (!(env && env[0] == '0') && hipMemGetInfo(&freeB, &totalB) == hipSuccess && need < freeB / 2) {
            if (hipMalloc(reinterpret_cast<void**>(&t->cells), need) != hipSuccess) { t->cells = nullptr; (void)hipGetLastError(); }
        }
    }
    *out = t;
    return 0;
}

int thx_texture_destroy(thx_texture* t)
{
    if (!t) return 0;
    (void)hipSetDevice(t->gpu);
    (void)hipFree(t->vol);
    (void)hipFree(t->cells);
    delete t;
    return 0;
}

int thx_texture_device(const thx_texture* t) { return t ? t->gpu : -1; }

/* ManagedCalPoint::Init(mode, cSearch, gpuIdx, nR, nT, mD, npxl) */
int thx_calpoint_create(thx_calpoint** out, int mode, int cSearch, int gpuIdx, int nR, int nT, int mD, int npxl)
{
    THX_REQUIRE(out && nR > 0 && nT > 0 && mD > 0 && npxl > 0, "bad arguments");
    THX_REQUIRE(mode == 1, "only MODE_3D (1) is implemented");
    THX_CHECK(hipSetDevice(gpuIdx));
    thx_calpoint* c = new thx_calpoint();
    memset(c, 0, sizeof(*c));
    c->mode = mode; c->cSearch = cSearch; c->gpu = gpuIdx; c->nR = nR; c->nT = nT; c->mD = mD; c->npxl = npxl;
    const int nD = cSearch == 2 ? mD : 1;
    const size_t nDbl = (size_t)nR + nT + mD + 1 + 4 * nR + 2 * nT + mD + 9 * nR;
    const size_t nFlt = 1 + (size_t)nR + nT + mD + 1 + (cSearch == 2 ? (size_t)mD * npxl : 0) + 2;
    double* d = nullptr;
    float* f = nullptr;
    c->nIn = nDbl; c->nOut = 1 + (size_t)nR + nT + mD + 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), nDbl * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f), nFlt * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->attr), sizeof(thx_ctf_attr));
    if (e == hipSuccess) e = hipMalloc(&c->ws, expect_local_single_workspace(npxl, nR, nT, nD));
    // (non-blocking: a copy another thread issues on the legacy default stream must not serialise the image-phases of every ManagedCalPoint)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->hIn), nDbl * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->hOut), (c->nOut + 4) * sizeof(float), hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->hSmall), sizeof(thx_ctf_attr) + 2 * sizeof(float), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipFree(d); (void)hipFree(f); (void)hipFree(c->attr); (void)hipFree(c->ws);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        if (c->hIn) (void)hipHostFree(c->hIn);
        if (c->hOut) (void)hipHostFree(c->hOut);
        if (c->hSmall) (void)hipHostFree(c->hSmall);
        delete c;
        set_error("ManagedCalPoint allocation failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    memset(c->hOut, 0, (c->nOut + 4) * sizeof(float));
    c->devR = d; d += nR;
    c->devT = d; d += nT;
    c->devD = d; d += mD;
    c->devC = d; d += 1;
    c->devnR = d; d += 4 * nR;
    c->devnT = d; d += 2 * nT;
    c->devdP = d; d += mD;
    c->devRotm = d;
    c->devwC = f; f += 1;
    c->devwR = f; f += nR;
    c->devwT = f; f += nT;
    c->devwD = f; f += mD;
    c->devBaseL = f; f += 1;
    c->k12 = f; f += 2;
    c->devctfD = cSearch == 2 ? f : nullptr;
    *out = c;
    return 0;
}

int thx_calpoint_destroy(thx_calpoint* c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->gpu);
    (void)hipStreamDestroy(c->stream);
    (void)hipFree(c->devR);    // base of the double block
    (void)hipFree(c->devwC);   // base of the float block
    (void)hipFree(c->attr);
    (void)hipFree(c->ws);
    (void)hipHostFree(c->hIn); (void)hipHostFree(c->hOut); (void)hipHostFree(c->hSmall);
    delete c;
    return 0;
}

/* void ExpectPreidx(int gpuIdx, int** deviCol, int** deviRow, int* iCol, int* iRow, int npxl)      Interface.h:18-23 */
int thx_ExpectPreidx_host(int gpuIdx, int** deviCol, int** deviRow, const int* iCol, const int* iRow, int npxl)
{
    THX_REQUIRE(deviCol && deviRow && iCol && iRow && npxl > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(deviCol), npxl * sizeof(int)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(deviRow), npxl * sizeof(int)));
    THX_CHECK(hipMemcpy(*deviCol, iCol, npxl * sizeof(int), hipMemcpyHostToDevice));
    THX_CHECK(hipMemcpy(*deviRow, iRow, npxl * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

/* void ExpectPrefre(int gpuIdx, RFLOAT** devfreQ, RFLOAT* freQ, int npxl)                           Interface.h:26-29 */
int thx_ExpectPrefre_host(int gpuIdx, float** devfreQ, const float* freQ, int npxl)
{
    THX_REQUIRE(devfreQ && freQ && npxl > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(devfreQ), npxl * sizeof(float)));
    THX_CHECK(hipMemcpy(*devfreQ, freQ, npxl * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

/* void ExpectLocalIn(int gpuIdx, Complex** devdatP, RFLOAT** devctfP, RFLOAT** devdefO, RFLOAT** devsigP, int nPxl,
 *                    int cpyNumL, int searchType)                                                    Interface.h:31-38
 * (gpu/src/cuthunder.cu:2460-2487: ctfP slots unless the defocus search is on, then defO slots) */
int thx_ExpectLocalIn_host(int gpuIdx, float** devdatP, float** devctfP, float** devdefO, float** devsigP, int nPxl,
                           int cpyNumL, int searchType)
{
    THX_REQUIRE(devdatP && devctfP && devdefO && devsigP && nPxl > 0 && cpyNumL > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    const size_t n = (size_t)cpyNumL * nPxl;
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(devdatP), n * 2 * sizeof(float)));
    if (searchType != 2) THX_CHECK(hipMalloc(reinterpret_cast<void**>(devctfP), n * sizeof(float)));
    else THX_CHECK(hipMalloc(reinterpret_cast<void**>(devdefO), n * sizeof(float)));
    THX_CHECK(hipMalloc(reinterpret_cast<void**>(devsigP), n * sizeof(float)));
    return 0;
}

/* void ExpectLocalV3D(int gpuIdx, ManagedArrayTexture* mgr, Complex* volume, int vdim)               Interface.h:45-48 */
int thx_ExpectLocalV3D_host(int gpuIdx, thx_texture* mgr, const float* volume, int vdim)
{
    (void)gpuIdx;  // the reference also uses the texture's own device (cuthunder.cu:2526)
    THX_REQUIRE(mgr && volume && vdim == mgr->vdim, "bad arguments (vdim must match ManagedArrayTexture::Init)");
    THX_CHECK(hipSetDevice(mgr->gpu));
    THX_CHECK(hipMemcpy(mgr->vol, volume, (size_t)vdim * vdim * (vdim / 2 + 1) * 2 * sizeof(float), hipMemcpyHostToDevice));
    if (mgr->cells) {   // the 8 corners of every trilinear cell side by side: the local search's gather becomes one 64-byte request per sample
        THX_RC(thx_projector_pack_dev(mgr->cells, mgr->vol, vdim, 1, nullptr));
        THX_CHECK(hipStreamSynchronize(nullptr));   // (the search runs on the ManagedCalPoints' own streams)
    }
    return 0;
}

/* void ExpectLocalP(int gpuIdx, Complex* devdatP, RFLOAT* devctfP, RFLOAT* devdefO, RFLOAT* devsigP, Complex* datP,
 *                   RFLOAT* ctfP, RFLOAT* defO, RFLOAT* sigP, int threadId, int imgId, int npxl, int cSearch)
 *                                                                                                    Interface.h:50-62 */
int thx_ExpectLocalP_host(int gpuIdx, float* devdatP, float* devctfP, float* devdefO, float* devsigP, const float* datP,
                          const float* ctfP, const float* defO, const float* sigP, int threadId, int imgId, int npxl,
                          int cSearch)
{
    THX_REQUIRE(devdatP && devsigP && datP && sigP && threadId >= 0 && imgId >= 0 && npxl > 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    const size_t shift = (size_t)imgId * npxl, slot = (size_t)threadId * npxl;
    // the slot's own (non-blocking) copy stream: a blocking hipMemcpy goes through the legacy default stream, which waits for -- and
    // holds up -- every other host thread's work; the copies are complete when the call returns, as the reference's are
    hipStream_t st = nullptr;
    THX_RC(slot_stream(gpuIdx, threadId, &st));
    THX_CHECK(hipMemcpyAsync(devdatP + 2 * slot, datP + 2 * shift, (size_t)npxl * 2 * sizeof(float), hipMemcpyHostToDevice, st));
    if (cSearch != 2) {
        THX_REQUIRE(devctfP && ctfP, "ctfP is NULL");
        THX_CHECK(hipMemcpyAsync(devctfP + slot, ctfP + shift, (size_t)npxl * sizeof(float), hipMemcpyHostToDevice, st));
    } else {
        THX_REQUIRE(devdefO && defO, "defO is NULL");
        THX_CHECK(hipMemcpyAsync(devdefO + slot, defO + shift, (size_t)npxl * sizeof(float), hipMemcpyHostToDevice, st));
    }
    THX_CHECK(hipMemcpyAsync(devsigP + slot, sigP + shift, (size_t)npxl * sizeof(float), hipMemcpyHostToDevice, st));
    THX_RC(wait_stream(st));
    return 0;
}

/* void ExpectLocalHostA(int gpuIdx, RFLOAT** wC, RFLOAT** wR, RFLOAT** wT, RFLOAT** wD, double** oldR, double** oldT,
 *                       double** oldD, double** trans, double** rot, double** dpara, int mR, int mT, int mD, int cSearch)
 *                                                                                                    Interface.h:64-78
 * page-locked host staging arrays (cuthunder.cu:2558-2613) */
int thx_ExpectLocalHostA_host(int gpuIdx, float** wC, float** wR, float** wT, float** wD, double** oldR, double** oldT,
                              double** oldD, double** trans, double** rot, double** dpara, int mR, int mT, int mD,
                              int cSearch)
{
    THX_REQUIRE(wC && wR && wT && wD && oldR && oldT && oldD && trans && rot && dpara, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(wC), sizeof(float)));
    THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(wR), mR * sizeof(float)));
    THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(wT), mT * sizeof(float)));
    THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(wD), mD * sizeof(float)));
    THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(oldR), mR * sizeof(double)));
    THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(oldT), mT * sizeof(double)));
    THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(trans), mT * 2 * sizeof(double)));
    THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(rot), mR * 4 * sizeof(double)));
    if (cSearch == 2) {
        THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(dpara), mD * sizeof(double)));
        THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(oldD), mD * sizeof(double)));
    } else {
        *dpara = nullptr;
        THX_CHECK(hipHostMalloc(reinterpret_cast<void**>(oldD), sizeof(double)));
    }
    return 0;
}

/* void ExpectLocalHostF(...)                                                                         Interface.h:141-152 */
int thx_ExpectLocalHostF_host(int gpuIdx, float** wC, float** wR, float** wT, float** wD, double** oldR, double** oldT,
                              double** oldD, double** trans, double** rot, double** dpara, int cSearch)
{
    THX_REQUIRE(wC && wR && wT && wD && oldR && oldT && oldD && trans && rot && dpara, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    void** all[] = {(void**)wC, (void**)wR, (void**)wT, (void**)wD, (void**)oldR, (void**)oldT, (void**)oldD, (void**)trans,
                    (void**)rot};
    for (void** p : all) { THX_CHECK(hipHostFree(*p)); *p = nullptr; }
    if (cSearch == 2 && *dpara) { THX_CHECK(hipHostFree(*dpara)); *dpara = nullptr; }
    return 0;
}

/* void ExpectLocalRTD(int gpuIdx, ManagedCalPoint* mcp, double* oldR, double* oldT, double* oldD, double* trans,
 *                     double* rot, double* dpara)                                                    Interface.h:80-87 */
int thx_ExpectLocalRTD_host(int gpuIdx, thx_calpoint* mcp, const double* oldR, const double* oldT, const double* oldD,
                            const double* trans, const double* rot, const double* dpara)
{
    THX_REQUIRE(mcp && oldR && oldT && oldD && trans && rot, "bad arguments");
    (void)gpuIdx;
    // staged on the host, in the layout of the device block (devR | devT | devD | devC | devnR | devnT | devdP | devRotm): nothing
    // touches the device here.  The caller's arrays may be rewritten as soon as this returns (they need not be page-locked).
    double* h = mcp->hIn;
    const int nR = mcp->nR, nT = mcp->nT, mD = mcp->mD;
    memcpy(h, oldR, nR * sizeof(double)); h += nR;
    memcpy(h, oldT, nT * sizeof(double)); h += nT;
    if (mcp->cSearch == 2) memcpy(h, oldD, mD * sizeof(double)); else h[0] = oldD[0];
    h += mD;
    h[0] = 1.0; h += 1;                                       // (devC: the class prior travels by value in ExpectLocalM)
    memcpy(h, rot, (size_t)nR * 4 * sizeof(double));
    const double* q = h; h += 4 * (size_t)nR;
    memcpy(h, trans, (size_t)nT * 2 * sizeof(double)); h += 2 * (size_t)nT;
    if (mcp->cSearch == 2) {
        THX_REQUIRE(dpara, "dpara is NULL");
        memcpy(h, dpara, mD * sizeof(double));
    }
    h += mD;
    for (int i = 0; i < nR; i++) rotate3d_colmajor(q + 4 * (size_t)i, h + 9 * (size_t)i);   // kernel_getRotMatL, gpu/src/cuthunder.cu:2851
    return 0;
}

/* void ExpectLocalPreI3D(int gpuIdx, int datShift, ManagedArrayTexture* mgr, ManagedCalPoint* mcp, RFLOAT* devdefO,
 *                        RFLOAT* devfreQ, int* deviCol, int* deviRow, RFLOAT phaseShift, RFLOAT conT, RFLOAT k1, RFLOAT k2,
 *                        int pf, int idim, int vdim, int npxl, int interp)                           Interface.h:107-123
 * kernel_TranslateL / kernel_CalCTFL / kernel_getRotMatL / kernel_Project3DL of cuthunder.cu:2834-2907: here the
 * rotation matrices and (defocus search) the CTF rows; ramps and slices are formed inside ExpectLocalM's kernel. */
int thx_ExpectLocalPreI3D_host(int gpuIdx, int datShift, const thx_texture* mgr, thx_calpoint* mcp, const float* devdefO,
                               const float* devfreQ, const int* deviCol, const int* deviRow, float phaseShift, float conT,
                               float k1, float k2, int pf, int idim, int vdim, int npxl, int interp)
{
    THX_REQUIRE(mgr && mcp && deviCol && deviRow && npxl == mcp->npxl && vdim == mgr->vdim && datShift >= 0, "bad arguments");
    THX_REQUIRE(interp == 1, "only LINEAR_INTERP (1) is implemented, as used by the 3D refinement path");
    THX_CHECK(hipSetDevice(gpuIdx));
    hipStream_t st = mcp->stream;
    mcp->vol = mgr->vol; mcp->cells = mgr->cells; mcp->iCol = deviCol; mcp->iRow = deviRow; mcp->pf = pf; mcp->idim = idim; mcp->vdim = vdim;
    // the whole staged block -- priors, quaternions, shifts, defocus factors, rotation matrices -- in ONE copy from page-locked memory
    THX_CHECK(hipMemcpyAsync(mcp->devR, mcp->hIn, mcp->nIn * sizeof(double), hipMemcpyHostToDevice, st));
    if (mcp->cSearch == 2) {
        THX_REQUIRE(devdefO && devfreQ, "defocus search needs devdefO and devfreQ");
        // (page-locked and owned by this ManagedCalPoint: rewritten by its next ExpectLocalPreI3D only, i.e. after ExpectLocalM has waited)
        thx_ctf_attr* hA = reinterpret_cast<thx_ctf_attr*>(mcp->hSmall);
        float* hK = reinterpret_cast<float*>(mcp->hSmall + sizeof(thx_ctf_attr));
        memset(hA, 0, sizeof(*hA));
        hA->amplitudeContrast = conT;
        hA->phaseShift = phaseShift;
        hK[0] = k1; hK[1] = k2;
        THX_CHECK(hipMemcpyAsync(mcp->attr, hA, sizeof(thx_ctf_attr), hipMemcpyHostToDevice, st));
        THX_CHECK(hipMemcpyAsync(mcp->k12, hK, 2 * sizeof(float), hipMemcpyHostToDevice, st));
        THX_RC(thx_ctf_dsearch_dev(mcp->devctfD, devfreQ, devdefO + (size_t)datShift * npxl, mcp->k12, mcp->k12 + 1, mcp->attr,
                                   mcp->devdP, mcp->mD, npxl, 1, st));
    }
    return 0;
}

/* void ExpectLocalM(int gpuIdx, int datShift, ManagedCalPoint* mcp, Complex* devdatP, RFLOAT* devctfP, RFLOAT* devsigP,
 *                   RFLOAT* wC, RFLOAT* wR, RFLOAT* wT, RFLOAT* wD, double oldC, int npxl)           Interface.h:125-139
 * kernel_logDataVSL(C) / kernel_getMaxBaseL / kernel_UpdateWL(C) of cuthunder.cu:2915-3172 = the body of HOT LOOP B
 * (src/Optimiser.cpp:1225-1406) for one image. */
int thx_ExpectLocalM_host(int gpuIdx, int datShift, thx_calpoint* mcp, const float* devdatP, const float* devctfP,
                          const float* devsigP, float* wC, float* wR, float* wT, float* wD, double oldC, int npxl)
{
    THX_REQUIRE(mcp && devdatP && devsigP && wC && wR && wT && wD && npxl == mcp->npxl && datShift >= 0, "bad arguments");
    THX_REQUIRE(mcp->vol, "ExpectLocalPreI3D has not been called on this ManagedCalPoint");
    THX_CHECK(hipSetDevice(gpuIdx));
    hipStream_t st = mcp->stream;
    const size_t slot = (size_t)datShift * npxl;
    const int nD = mcp->cSearch == 2 ? mcp->mD : 1;
    const float* ctf = mcp->cSearch == 2 ? mcp->devctfD : devctfP + slot;
    THX_REQUIRE(ctf, "devctfP is NULL");
    // three launches (fused gather + likelihood over ranges of 32 pixels, the reduce of the partial sums, finalise), one wait
    static const bool zeroCopy = []() { const char* e = getenv("THX_IFACE_ZEROCOPY"); return !(e && e[0] == '0'); }();
    if (zeroCopy) {
        // the finalise kernel writes wC | wR | wT | wD | baseLine straight into the page-locked (device-visible, coherent) host block:
        // one dependent operation less in the chain of an image-phase than a copy back (THX_IFACE_ZEROCOPY=0: the copy)
        float* o = mcp->hOut;
        unsigned* flag = reinterpret_cast<unsigned*>(mcp->hOut + mcp->nOut);
        const unsigned val = ++mcp->seq ? mcp->seq : ++mcp->seq;   // (never 0, the word's initial value)
        THX_RC(expect_local_single(mcp->cells ? mcp->cells : mcp->vol, mcp->cells != nullptr, mcp->vdim, mcp->pf, mcp->idim, mcp->iCol, mcp->iRow, npxl,
                                   devdatP + 2 * slot, ctf, devsigP + slot, mcp->devRotm, mcp->nR, mcp->devnT, mcp->nT, nD, oldC, mcp->devR,
                                   mcp->devT, mcp->devD, o, o + 1, o + 1 + mcp->nR, o + 1 + mcp->nR + mcp->nT, o + 1 + mcp->nR + mcp->nT + mcp->mD, mcp->ws, st, flag, val));
        THX_RC(wait_flag(flag, val, st));
    } else {
        THX_RC(expect_local_single(mcp->cells ? mcp->cells : mcp->vol, mcp->cells != nullptr, mcp->vdim, mcp->pf, mcp->idim, mcp->iCol, mcp->iRow, npxl,
                                   devdatP + 2 * slot, ctf, devsigP + slot, mcp->devRotm, mcp->nR, mcp->devnT, mcp->nT, nD, oldC, mcp->devR,
                                   mcp->devT, mcp->devD, mcp->devwC, mcp->devwR, mcp->devwT, mcp->devwD, mcp->devBaseL, mcp->ws, st));
        THX_CHECK(hipMemcpyAsync(mcp->hOut, mcp->devwC, mcp->nOut * sizeof(float), hipMemcpyDeviceToHost, st));
        THX_RC(wait_stream(st));
    }
    const float* o = mcp->hOut;
    wC[0] = o[0]; o += 1;
    memcpy(wR, o, mcp->nR * sizeof(float)); o += mcp->nR;
    memcpy(wT, o, mcp->nT * sizeof(float)); o += mcp->nT;
    memcpy(wD, o, nD * sizeof(float));
    return 0;
}

/* void ExpectLocalFin(int gpuIdx, Complex** devdatP, RFLOAT** devctfP, RFLOAT** devdefO, RFLOAT** devfreQ,
 *                     RFLOAT** devsigP, int cSearch)                                                 Interface.h:154-160 */
int thx_ExpectLocalFin_host(int gpuIdx, float** devdatP, float** devctfP, float** devdefO, float** devfreQ, float** devsigP,
                            int cSearch)
{
    THX_REQUIRE(devdatP && devsigP, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    THX_CHECK(hipFree(*devdatP)); *devdatP = nullptr;
    THX_CHECK(hipFree(*devsigP)); *devsigP = nullptr;
    if (cSearch != 2) {
        if (devctfP && *devctfP) { THX_CHECK(hipFree(*devctfP)); *devctfP = nullptr; }
    } else {
        if (devdefO && *devdefO) { THX_CHECK(hipFree(*devdefO)); *devdefO = nullptr; }
        if (devfreQ && *devfreQ) { THX_CHECK(hipFree(*devfreQ)); *devfreQ = nullptr; }
    }
    return 0;
}

/* void ExpectFreeIdx(int gpuIdx, int** deviCol, int** deviRow)                                       Interface.h:162-164 */
int thx_ExpectFreeIdx_host(int gpuIdx, int** deviCol, int** deviRow)
{
    THX_REQUIRE(deviCol && deviRow, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    THX_CHECK(hipFree(*deviCol)); *deviCol = nullptr;
    THX_CHECK(hipFree(*deviRow)); *deviRow = nullptr;
    return 0;
}

}  // extern "C"
