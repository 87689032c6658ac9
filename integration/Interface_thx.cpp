// integration/Interface_thx.cpp -- drop-in replacement of thuem/THUNDER's gpu/interface/Interface.cpp (v1.4.14).
//
// REFERENCE-SIDE FILE: it is compiled in the reference's tree (it needs the reference's Volume, Image, vec, CTFAttr,
// TabFunction, ManagedArrayTexture / ManagedCalPoint types through the reference's own, UNCHANGED gpu/interface/Interface.h)
// with -DGPU_VERSION and -I<this repo>/include, and linked with -lthunder_amd instead of the CUDA objects.  Nothing in this
// repository includes it.  Every function keeps the exact prototype of Interface.h (tools/iface_check.py diffs the
// parameter lists against gpu/interface/Interface.h:16-528 and tests/test_abi_cpu.py runs that check); the body unpacks the C++
// containers and forwards to the C ABI of include/thunder_amd.h.  CK() restores the reference's error convention
// (cudaCheckErrors prints and exits, gpu/config/Device.cuh.in:27-38).
//
// ManagedArrayTexture / ManagedCalPoint: the reference's classes (gpu/include/ManagedArrayTexture.h, ManagedCalPoint.h) wrap
// CUDA objects; the replacement keeps their names and the methods Optimiser.cpp calls (Init, getDeviceId) and holds one opaque
// handle `_h` (thx_texture* / thx_calpoint*) with an accessor handle() -- those two small headers are replaced with it.
//
// Not provided (2-D mode, outside the path of BASELINE.json): ExpectGlobal2D, ExpectLocalV2D, ExpectLocalPreI2D, InsertI2D,
// ExposePT2D, ExposeWT2D, ExposePF2D, ExposeCorrF2D.
#include "Interface.h"          // the reference's own header, unchanged
#include "thunder_amd.h"

#define CK(call) do { if ((call) != 0) { fprintf(stderr, "thunder_amd: %s\n", thx_last_error()); abort(); } } while (0)

void getAviDevice(std::vector<int>& gpus)                              // Interface.h:16
{
    int n = 0; CK(thx_device_count(&n));
    gpus.clear(); for (int i = 0; i < n; i++) gpus.push_back(i);
}

void ExpectRotran(Complex* traP, double* trans, double* rot, double* rotMat,
                  const int* iCol, const int* iRow, int nR, int nT, int idim, int npxl)   // Interface.h:199
{ CK(thx_ExpectRotran_host((float*)traP, trans, rot, rotMat, iCol, iRow, nR, nT, idim, npxl)); }

void ExpectProject(Complex* volume, Complex* rotP, double* rotMat, const int* iCol, const int* iRow,
                   int nR, int pf, int interp, int vdim, int npxl)                         // Interface.h:210
{ CK(thx_ExpectProject_host((const float*)volume, (float*)rotP, rotMat, iCol, iRow, nR, pf, interp, vdim, npxl)); }

void InsertFT(Volume& F3D, Volume& T3D, double* O3D, int* counter, MPI_Comm& hemi, MPI_Comm& slav,
              Complex* datP, RFLOAT* ctfP, RFLOAT* sigRcpP, CTFAttr* ctfaData, double* offS, RFLOAT* w,
              double* nR, double* nT, double* nD, int* nC, const int* iCol, const int* iRow,
              RFLOAT pixelSize, bool cSearch, int opf, int npxl, int mReco, int idim, int dimSize, int imgNum)
{                                                                                          // Interface.h:267
    // CTFAttr (include/Database.h:302-330) and thx_ctf_attr are the same 7 floats in the same order
    CK(thx_InsertFT_host((float*)&F3D[0], (float*)&T3D[0], O3D, counter, (const float*)datP, ctfP,
                         (const thx_ctf_attr*)ctfaData, offS, w, nR, nT, nD, nC, iCol, iRow,
                         pixelSize, cSearch, opf, npxl, mReco, idim, F3D.nRowRL(), /*nK*/ 1, imgNum));
    // hemisphere reduction: the reference does it with NCCL inside InsertFT (gpu/src/cuthunder.cu:4972-5067);
    // with this library either thx_InsertFT_hemi_host (the same call with a thx_comm* hemisphere communicator: RCCL
    // inside, section 4) or, as written here, where the CPU build has it -- MPI_Allreduce_Large in
    // Reconstructor::allReduceF/T (src/Reconstructor.cpp:2383,2436).
    MPI_Allreduce_Large(&F3D[0], 2 * F3D.sizeFT(), TS_MPI_DOUBLE, MPI_SUM, hemi);
    MPI_Allreduce_Large(&T3D[0], 2 * T3D.sizeFT(), TS_MPI_DOUBLE, MPI_SUM, hemi);
    MPI_Allreduce(MPI_IN_PLACE, O3D, 3, MPI_DOUBLE, MPI_SUM, slav);
    MPI_Allreduce(MPI_IN_PLACE, counter, 1, MPI_INT, MPI_SUM, slav);
}

void InsertFT(Volume& F3D, Volume& T3D, double* O3D, int* counter, MPI_Comm& hemi, MPI_Comm& slav,
              Complex* datP, RFLOAT* ctfP, RFLOAT* sigRcpP, CTFAttr* ctfaData, double* offS, RFLOAT* w,
              double* nR, double* nT, double* nD, const int* iCol, const int* iRow,
              RFLOAT pixelSize, bool cSearch, int opf, int npxl, int mReco, int idim, int dimSize, int imgNum)
{                                                                                          // Interface.h:293 (one reference: no nC)
    InsertFT(F3D, T3D, O3D, counter, hemi, slav, datP, ctfP, sigRcpP, ctfaData, offS, w, nR, nT, nD, (int*)NULL, iCol, iRow,
             pixelSize, cSearch, opf, npxl, mReco, idim, dimSize, imgNum);
}

void PrepareTF(int gpuIdx, Volume& F3D, Volume& T3D, double* symMat, int nSymmetryElement,
               int maxRadius, int pf)                                                      // Interface.h:320
{ CK(thx_PrepareTF_host(gpuIdx, (float*)&F3D[0], (float*)&T3D[0], F3D.nRowRL(), symMat, nSymmetryElement, maxRadius, pf)); }

void ExpectGlobal3D(Complex* rotP, Complex* traP, Complex* datP, RFLOAT* ctfP, RFLOAT* sigRcpP, RFLOAT* wC, RFLOAT* wR,
                    RFLOAT* wT, double* pR, double* pT, RFLOAT* baseL, int kIdx, int nK, int nR, int nT,
                    int npxl, int imgNum)                                                  // Interface.h:221
{ CK(thx_ExpectGlobal3D_host((const float*)rotP, (const float*)traP, (const float*)datP, ctfP, sigRcpP, wC, wR, wT,
                             pR, pT, baseL, kIdx, nK, nR, nT, npxl, imgNum)); }

void ExpectPrecal(vector<CTFAttr>& ctfAttr, RFLOAT* def, RFLOAT* k1, RFLOAT* k2, const int* iCol, const int* iRow,
                  int idim, int npxl, int imgNum)                                          // Interface.h:166
{ CK(thx_ExpectPrecal_host((const thx_ctf_attr*)&ctfAttr[0], def, k1, k2, iCol, iRow, idim, npxl, imgNum)); }

void GCTFinit(vector<Image>& img, vector<CTFAttr>& ctfAttr, RFLOAT pixelSize, int idim, int imgNum)  // Interface.h:524
{
    std::vector<float*> p(imgNum);
    for (int l = 0; l < imgNum; l++) p[l] = (float*)&img[l][0];
    CK(thx_GCTFinit_host(p.data(), (const thx_ctf_attr*)&ctfAttr[0], pixelSize, idim, imgNum));
}

void ReMask(vector<Image>& img, RFLOAT maskRadius, RFLOAT pixelSize, RFLOAT ew, int idim, int imgNum)  // Interface.h:517
{
    std::vector<float*> p(imgNum);
    for (int l = 0; l < imgNum; l++) p[l] = (float*)&img[l][0];
    CK(thx_ReMask_host(p.data(), maskRadius, pixelSize, ew, idim, imgNum));
}

void TranslateI2D(int gpuIdx, Image& img, double ox, double oy, int r)                     // Interface.h:504
{ CK(thx_TranslateI2D_host(gpuIdx, (float*)&img[0], ox, oy, r, img.nRowRL())); }

void TranslateI(int gpuIdx, Volume& ref, double ox, double oy, double oz, int r)           // Interface.h:510
{ CK(thx_TranslateI_host(gpuIdx, (float*)&ref[0], ox, oy, oz, r, ref.nRowRL())); }

// ---- the staged entry points: Optimiser.cpp and Reconstructor.cpp stay untouched ----
// ManagedArrayTexture / ManagedCalPoint (gpu/include/*.h) keep their interface; each holds the opaque handle.
void ManagedArrayTexture::Init(int mode, int vdim, int gpuIdx) { CK(thx_texture_create(&_h, mode, vdim, gpuIdx)); }
ManagedArrayTexture::~ManagedArrayTexture()                    { thx_texture_destroy(_h); }
int  ManagedArrayTexture::getDeviceId()                        { return thx_texture_device(_h); }
void ManagedCalPoint::Init(int mode, int cSearch, int gpuIdx, int nR, int nT, int mD, int npxl)
{ CK(thx_calpoint_create(&_h, mode, cSearch, gpuIdx, nR, nT, mD, npxl)); }
ManagedCalPoint::~ManagedCalPoint()                            { thx_calpoint_destroy(_h); }

void ExpectPreidx(int gpuIdx, int** deviCol, int** deviRow, int* iCol, int* iRow, int npxl)          // Interface.h:18
{ CK(thx_ExpectPreidx_host(gpuIdx, deviCol, deviRow, iCol, iRow, npxl)); }
void ExpectPrefre(int gpuIdx, RFLOAT** devfreQ, RFLOAT* freQ, int npxl)                               // :26
{ CK(thx_ExpectPrefre_host(gpuIdx, devfreQ, freQ, npxl)); }
void ExpectLocalIn(int gpuIdx, Complex** devdatP, RFLOAT** devctfP, RFLOAT** devdefO, RFLOAT** devsigP,
                   int nPxl, int cpyNumL, int searchType)                                            // :31
{ CK(thx_ExpectLocalIn_host(gpuIdx, (float**)devdatP, devctfP, devdefO, devsigP, nPxl, cpyNumL, searchType)); }
void ExpectLocalV3D(int gpuIdx, ManagedArrayTexture* mgr, Complex* volume, int vdim)                  // :45
{ CK(thx_ExpectLocalV3D_host(gpuIdx, mgr->handle(), (const float*)volume, vdim)); }
void ExpectLocalP(int gpuIdx, Complex* devdatP, RFLOAT* devctfP, RFLOAT* devdefO, RFLOAT* devsigP, Complex* datP,
                  RFLOAT* ctfP, RFLOAT* defO, RFLOAT* sigP, int threadId, int imgId, int npxl, int cSearch)  // :50
{ CK(thx_ExpectLocalP_host(gpuIdx, (float*)devdatP, devctfP, devdefO, devsigP, (const float*)datP, ctfP, defO, sigP,
                           threadId, imgId, npxl, cSearch)); }
void ExpectLocalHostA(int gpuIdx, RFLOAT** wC, RFLOAT** wR, RFLOAT** wT, RFLOAT** wD, double** oldR, double** oldT,
                      double** oldD, double** trans, double** rot, double** dpara, int mR, int mT, int mD, int cSearch)
{ CK(thx_ExpectLocalHostA_host(gpuIdx, wC, wR, wT, wD, oldR, oldT, oldD, trans, rot, dpara, mR, mT, mD, cSearch)); } // :64
void ExpectLocalRTD(int gpuIdx, ManagedCalPoint* mcp, double* oldR, double* oldT, double* oldD, double* trans,
                    double* rot, double* dpara)                                                      // :80
{ CK(thx_ExpectLocalRTD_host(gpuIdx, mcp->handle(), oldR, oldT, oldD, trans, rot, dpara)); }
void ExpectLocalPreI3D(int gpuIdx, int datShift, ManagedArrayTexture* mgr, ManagedCalPoint* mcp, RFLOAT* devdefO,
                       RFLOAT* devfreQ, int* deviCol, int* deviRow, RFLOAT phaseShift, RFLOAT conT, RFLOAT k1, RFLOAT k2,
                       int pf, int idim, int vdim, int npxl, int interp)                             // :107
{ CK(thx_ExpectLocalPreI3D_host(gpuIdx, datShift, mgr->handle(), mcp->handle(), devdefO, devfreQ, deviCol, deviRow,
                                phaseShift, conT, k1, k2, pf, idim, vdim, npxl, interp)); }
void ExpectLocalM(int gpuIdx, int datShift, ManagedCalPoint* mcp, Complex* devdatP, RFLOAT* devctfP, RFLOAT* devsigP,
                  RFLOAT* wC, RFLOAT* wR, RFLOAT* wT, RFLOAT* wD, double oldC, int npxl)             // :125
{ CK(thx_ExpectLocalM_host(gpuIdx, datShift, mcp->handle(), (const float*)devdatP, devctfP, devsigP, wC, wR, wT, wD,
                           oldC, npxl)); }
void ExpectLocalHostF(int gpuIdx, RFLOAT** wC, RFLOAT** wR, RFLOAT** wT, RFLOAT** wD, double** oldR, double** oldT,
                      double** oldD, double** trans, double** rot, double** dpara, int cSearch)       // :141
{ CK(thx_ExpectLocalHostF_host(gpuIdx, wC, wR, wT, wD, oldR, oldT, oldD, trans, rot, dpara, cSearch)); }
void ExpectLocalFin(int gpuIdx, Complex** devdatP, RFLOAT** devctfP, RFLOAT** devdefO, RFLOAT** devfreQ,
                    RFLOAT** devsigP, int cSearch)                                                    // :154
{ CK(thx_ExpectLocalFin_host(gpuIdx, (float**)devdatP, devctfP, devdefO, devfreQ, devsigP, cSearch)); }
void ExpectFreeIdx(int gpuIdx, int** deviCol, int** deviRow)                                          // :162
{ CK(thx_ExpectFreeIdx_host(gpuIdx, deviCol, deviRow)); }

void ExposePT(int gpuIdx, RFLOAT* T3D, int maxRadius, int pf, int dim, vec FSC, bool joinHalf, const int wienerF) // :337
{ CK(thx_ExposePT_host(gpuIdx, T3D, maxRadius, pf, dim, FSC.data(), (int)FSC.size(), joinHalf, wienerF)); }
void ExposeWT(int gpuIdx, RFLOAT* T3D, RFLOAT* W3D, TabFunction& kernelRL, RFLOAT nf, int maxRadius, int pf, int dim,
              int maxIter, int minIter, int size)                                                    // :438
{ CK(thx_ExposeWT_host(gpuIdx, T3D, W3D, kernelRL.getData(), 100001, nf, maxRadius, pf, dim, maxIter, minIter, size)); }
void ExposeWT(int gpuIdx, RFLOAT* T3D, RFLOAT* W3D, int maxRadius, int pf, int dim)                   // :457
{ CK(thx_ExposeWT_plain_host(gpuIdx, T3D, W3D, maxRadius, pf, dim)); }
void AllocDevicePoint(int gpuIdx, Complex** dev_C, RFLOAT** dev_W, RFLOAT** dev_T, RFLOAT** dev_tab, RFLOAT** devDiff,
                      RFLOAT** devMax, int** devCount, void** stream, int streamNum, int tabSize, int dim)  // :358
{ CK(thx_AllocDevicePoint_host(gpuIdx, (float**)dev_C, dev_W, dev_T, dev_tab, devDiff, devMax, devCount, stream,
                               streamNum, tabSize, dim)); }
void HostDeviceInit(int gpuIdx, Volume& C3D, RFLOAT* W3D, RFLOAT* T3D, RFLOAT* tab, RFLOAT* dev_W, RFLOAT* dev_T,
                    RFLOAT* dev_tab, void** stream, int streamNum, int tabSize, int maxRadius, int pf, int dim) // :371
{ CK(thx_HostDeviceInit_host(gpuIdx, T3D, tab, dev_W, dev_T, dev_tab, stream, streamNum, tabSize, maxRadius, pf, dim)); }
void ExposeC(int gpuIdx, Volume& C3D, Complex* dev_C, RFLOAT* dev_T, RFLOAT* dev_W, void** stream, int streamNum,
             int dim)                                                                                // :386
{ CK(thx_ExposeC_host(gpuIdx, (float*)&C3D[0], (float*)dev_C, dev_T, dev_W, stream, streamNum, dim)); }
void ExposeForConvC(int gpuIdx, Volume& C3D, Complex* dev_C, RFLOAT* dev_tab, void** stream, TabFunction& kernelRL,
                    RFLOAT nf, int streamNum, int tabSize, int pf, int size)                         // :395
{ CK(thx_ExposeForConvC_host(gpuIdx, &C3D(0), (float*)dev_C, dev_tab, stream, kernelRL.getStep(), nf, streamNum,
                             tabSize, pf, size, C3D.nSlcRL())); }
void ExposeWC(int gpuIdx, Volume& C3D, Complex* dev_C, RFLOAT* diff, RFLOAT* cmax, RFLOAT* dev_W, RFLOAT* devDiff,
              RFLOAT* devMax, int* devCount, int* counter, void** stream, RFLOAT& diffC, int streamNum, int maxRadius,
              int pf)                                                                                // :407
{ CK(thx_ExposeWC_host(gpuIdx, (const float*)&C3D[0], (float*)dev_C, cmax, dev_W, devMax, stream, &diffC, streamNum,
                       maxRadius, pf, C3D.nSlcFT())); }
void FreeDevHostPoint(int gpuIdx, Complex** dev_C, RFLOAT** dev_W, RFLOAT** dev_T, RFLOAT** dev_tab, RFLOAT** devDiff,
                      RFLOAT** devMax, int** devCount, void** stream, Volume& C3D, RFLOAT* volumeW, RFLOAT* volumeT,
                      int streamNum, int dim)                                                        // :423
{ CK(thx_FreeDevHostPoint_host(gpuIdx, (float**)dev_C, dev_W, dev_T, dev_tab, devDiff, devMax, devCount, stream,
                               volumeW, streamNum, dim)); }
void ExposePFW(int gpuIdx, Volume& padDst, Volume& F3D, RFLOAT* W3D, int maxRadius, int pf)           // :472
{ CK(thx_ExposePFW_host(gpuIdx, (float*)&padDst[0], (const float*)&F3D[0], W3D, maxRadius, pf, padDst.nSlcFT(),
                        F3D.nSlcFT())); }
void ExposePF(int gpuIdx, Volume& padDst, Volume& padDstR, Volume& F3D, RFLOAT* W3D, int maxRadius, int pf)  // :479
{ CK(thx_ExposePF_host(gpuIdx, (float*)&padDst[0], &padDstR(0), (const float*)&F3D[0], W3D, maxRadius, pf,
                       padDst.nSlcFT(), F3D.nSlcFT())); }
void ExposeCorrF(int gpuIdx, Volume& dst, RFLOAT* mkbRL, RFLOAT nf)                                   // :493
{ CK(thx_ExposeCorrF_host(gpuIdx, &dst(0), mkbRL, nf, dst.nSlcRL())); }
void ExposeCorrF(int gpuIdx, Volume& dstN, Volume& dst, RFLOAT* mkbRL, RFLOAT nf)                     // :498
{ CK(thx_ExposeCorrF_fft_host(gpuIdx, &dstN(0), (float*)&dst[0], mkbRL, nf, dstN.nSlcRL())); }
