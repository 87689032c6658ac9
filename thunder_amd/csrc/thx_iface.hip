// thx_iface.hip -- the per-image, staged local-search entry points of gpu/interface/Interface.h:18-164
// (ExpectPreidx / Prefre / LocalIn / LocalV3D / LocalP / LocalHostA / LocalRTD / LocalPreI3D / LocalM / LocalHostF /
// LocalFin / FreeIdx) with the two helper classes they pass around (ManagedArrayTexture, ManagedCalPoint) as opaque
// handles.  Optimiser::expectationG (src/Optimiser.cpp:2180-3393) drives them image by image from OpenMP threads; every
// entry keeps the argument meaning of its twin, so a replacement Interface.cpp forwards one to one.
//
// The work itself is the batched kernel of thx_estep.hip run on a batch of one image: ExpectLocalPreI3D stages the
// rotation matrices (and, in the defocus search, the CTF rows) and ExpectLocalM runs the fused gather + likelihood +
// weight kernel -- the slices are never materialised.  This is the compatibility granularity; the batched
// thx_expect_local_dev is the one to use for throughput.
#include "thx_common.h"

using namespace thx;

struct thx_texture {      // ManagedArrayTexture (gpu/include/ManagedArrayTexture.h): a device-resident padded FT
    int mode, vdim, gpu;
    float* vol;           // [vdim][vdim][vdim/2+1] complex64
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
    // recorded by ExpectLocalPreI3D for ExpectLocalM
    const float* vol;
    const int *iCol, *iRow;
    int pf, idim, vdim;
    thx_ctf_attr hAttr;
    float hK12[2];
    double hC;
};

extern "C" {

/* ManagedArrayTexture::Init(mode, vdim, gpuIdx) */
int thx_texture_create(thx_texture** out, int mode, int vdim, int gpuIdx)
{
    THX_REQUIRE(out && vdim > 0, "bad arguments");
    THX_REQUIRE(mode == 1, "only MODE_3D (1) is implemented");
    THX_CHECK(hipSetDevice(gpuIdx));
    thx_texture* t = new thx_texture();
    t->mode = mode; t->vdim = vdim; t->gpu = gpuIdx; t->vol = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->vol), (size_t)vdim * vdim * (vdim / 2 + 1) * 2 * sizeof(float));
    if (e != hipSuccess) { delete t; set_error("hipMalloc failed: %s", hipGetErrorString(e)); return (int)e; }
    *out = t;
    return 0;
}

int thx_texture_destroy(thx_texture* t)
{
    if (!t) return 0;
    (void)hipSetDevice(t->gpu);
    (void)hipFree(t->vol);
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
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), nDbl * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f), nFlt * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->attr), sizeof(thx_ctf_attr));
    if (e == hipSuccess) e = hipMalloc(&c->ws, thx_expect_local_workspace(1, nR, nT, nD));
    if (e == hipSuccess) e = hipStreamCreate(&c->stream);
    if (e != hipSuccess) {
        (void)hipFree(d); (void)hipFree(f); (void)hipFree(c->attr); (void)hipFree(c->ws);
        delete c;
        set_error("ManagedCalPoint allocation failed: %s", hipGetErrorString(e));
        return (int)e;
    }
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
    THX_CHECK(hipMemcpy(devdatP + 2 * slot, datP + 2 * shift, (size_t)npxl * 2 * sizeof(float), hipMemcpyHostToDevice));
    if (cSearch != 2) {
        THX_REQUIRE(devctfP && ctfP, "ctfP is NULL");
        THX_CHECK(hipMemcpy(devctfP + slot, ctfP + shift, (size_t)npxl * sizeof(float), hipMemcpyHostToDevice));
    } else {
        THX_REQUIRE(devdefO && defO, "defO is NULL");
        THX_CHECK(hipMemcpy(devdefO + slot, defO + shift, (size_t)npxl * sizeof(float), hipMemcpyHostToDevice));
    }
    THX_CHECK(hipMemcpy(devsigP + slot, sigP + shift, (size_t)npxl * sizeof(float), hipMemcpyHostToDevice));
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
    THX_CHECK(hipSetDevice(gpuIdx));
    hipStream_t st = mcp->stream;
    THX_CHECK(hipMemcpyAsync(mcp->devR, oldR, mcp->nR * sizeof(double), hipMemcpyHostToDevice, st));
    THX_CHECK(hipMemcpyAsync(mcp->devT, oldT, mcp->nT * sizeof(double), hipMemcpyHostToDevice, st));
    THX_CHECK(hipMemcpyAsync(mcp->devnR, rot, (size_t)mcp->nR * 4 * sizeof(double), hipMemcpyHostToDevice, st));
    THX_CHECK(hipMemcpyAsync(mcp->devnT, trans, (size_t)mcp->nT * 2 * sizeof(double), hipMemcpyHostToDevice, st));
    if (mcp->cSearch == 2) {
        THX_REQUIRE(dpara, "dpara is NULL");
        THX_CHECK(hipMemcpyAsync(mcp->devdP, dpara, mcp->mD * sizeof(double), hipMemcpyHostToDevice, st));
        THX_CHECK(hipMemcpyAsync(mcp->devD, oldD, mcp->mD * sizeof(double), hipMemcpyHostToDevice, st));
    } else {
        THX_CHECK(hipMemcpyAsync(mcp->devD, oldD, sizeof(double), hipMemcpyHostToDevice, st));
    }
    // the caller reuses its staging arrays for the next image as soon as the phase returns; they need not be page-locked
    THX_CHECK(hipStreamSynchronize(st));
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
    mcp->vol = mgr->vol; mcp->iCol = deviCol; mcp->iRow = deviRow; mcp->pf = pf; mcp->idim = idim; mcp->vdim = vdim;
    THX_RC(thx_rotmat_dev(mcp->devnR, mcp->devRotm, mcp->nR, st));
    if (mcp->cSearch == 2) {
        THX_REQUIRE(devdefO && devfreQ, "defocus search needs devdefO and devfreQ");
        memset(&mcp->hAttr, 0, sizeof(mcp->hAttr));
        mcp->hAttr.amplitudeContrast = conT;
        mcp->hAttr.phaseShift = phaseShift;
        mcp->hK12[0] = k1; mcp->hK12[1] = k2;
        THX_CHECK(hipMemcpyAsync(mcp->attr, &mcp->hAttr, sizeof(thx_ctf_attr), hipMemcpyHostToDevice, st));
        THX_CHECK(hipMemcpyAsync(mcp->k12, mcp->hK12, 2 * sizeof(float), hipMemcpyHostToDevice, st));
        THX_RC(thx_ctf_dsearch_dev(mcp->devctfD, devfreQ, devdefO + (size_t)datShift * npxl, mcp->k12, mcp->k12 + 1, mcp->attr,
                                   mcp->devdP, mcp->mD, npxl, 1, st));
        THX_CHECK(hipStreamSynchronize(st));  // hAttr / hK12 may be rewritten by the next call
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
    mcp->hC = oldC;
    THX_CHECK(hipMemcpyAsync(mcp->devC, &mcp->hC, sizeof(double), hipMemcpyHostToDevice, st));
    THX_RC(thx_expect_local_dev(mcp->vol, nullptr, mcp->vdim, mcp->pf, mcp->idim, mcp->iCol, mcp->iRow, npxl, 1,
                                devdatP + 2 * slot, ctf, devsigP + slot, mcp->devRotm, mcp->nR, mcp->devnT, mcp->nT, nD,
                                mcp->devC, mcp->devR, mcp->devT, mcp->devD, mcp->devwC, mcp->devwR, mcp->devwT, mcp->devwD,
                                mcp->devBaseL, nullptr, mcp->ws, -1, nullptr, st));
    THX_CHECK(hipMemcpyAsync(wC, mcp->devwC, sizeof(float), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipMemcpyAsync(wR, mcp->devwR, mcp->nR * sizeof(float), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipMemcpyAsync(wT, mcp->devwT, mcp->nT * sizeof(float), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipMemcpyAsync(wD, mcp->devwD, nD * sizeof(float), hipMemcpyDeviceToHost, st));
    THX_CHECK(hipStreamSynchronize(st));
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
