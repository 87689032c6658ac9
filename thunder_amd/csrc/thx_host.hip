// thx_host.hip -- Interface.h-shaped entry points on caller-owned HOST buffers
// (gpu/interface/Interface.h:199-219,267-326; Reconstructor::reconstructG src/Reconstructor.cpp:1835-2315).
// Each one uploads, runs the *_dev path, and writes results back into the caller's arrays before returning,
// which is the contract the reference's -DGPU_VERSION call sites rely on.
#include <mutex>
#include <vector>

#include "thx_common.h"

using namespace thx;

namespace {
// The reference's T volume is a complex Volume whose imaginary part is never written (src/Reconstructor.cpp:782-863): the device works on
// the real part.  The (de)interleaving runs ON THE DEVICE -- the host loops over 67 M voxels it replaces cost 50 ms each way at 512^3.
__global__ void k_take_real(float* __restrict__ dst, const float2* __restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i].x;
}
__global__ void k_put_real(float2* __restrict__ dst, const float* __restrict__ src, size_t n, int zeroImag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { dst[i].x = src[i]; if (zeroImag) dst[i].y = 0.f; }
}
// host complex T -> device real T (dT) through a device copy of the complex volume (dTc, kept for the way back)
int upload_T(DevBuf& dTc, DevBuf& dT, const float* T3D_complex, size_t n)
{
    THX_RC(dTc.upload(T3D_complex, n * 2 * sizeof(float)));
    THX_RC(dT.alloc(n * sizeof(float)));
    hipLaunchKernelGGL(k_take_real, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, dT.as<float>(), dTc.as<float2>(), n);
    THX_LAUNCH_CHECK();
    return 0;
}
int download_T(float* T3D_complex, DevBuf& dTc, const float* dT, size_t n, int zeroImag)
{
    hipLaunchKernelGGL(k_put_real, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, dTc.as<float2>(), dT, n, zeroImag);
    THX_LAUNCH_CHECK();
    THX_CHECK(hipMemcpy(T3D_complex, dTc.p, n * 2 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
}  // namespace

extern "C" {

int thx_malloc_dev(void** ptr, size_t bytes)
{
    THX_REQUIRE(ptr, "ptr is NULL");
    THX_CHECK(hipMalloc(ptr, bytes ? bytes : 4));
    return 0;
}
int thx_free_dev(void* ptr)
{
    if (ptr) THX_CHECK(hipFree(ptr));
    return 0;
}
int thx_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes)
{
    THX_CHECK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return 0;
}
int thx_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes)
{
    THX_CHECK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int thx_memset_dev(void* dst_dev, int value, size_t bytes)
{
    THX_CHECK(hipMemset(dst_dev, value, bytes));
    return 0;
}
int thx_device_sync(void)
{
    THX_CHECK(hipDeviceSynchronize());
    return 0;
}

int thx_ExpectProject_host(const float* volume, float* rotP, const double* rotMat, const int* iCol, const int* iRow,
                           int nR, int pf, int interp, int vdim, int npxl)
{
    THX_REQUIRE(interp == 1, "only LINEAR_INTERP (1) is implemented, as used by the 3D refinement path");
    const size_t nvol = (size_t)vdim * vdim * (vdim / 2 + 1);
    DevBuf dVol, dOut, dMat, dCol, dRow;
    THX_RC(dVol.upload(volume, nvol * 2 * sizeof(float)));
    THX_RC(dOut.alloc((size_t)nR * npxl * 2 * sizeof(float)));
    THX_RC(dMat.upload(rotMat, (size_t)nR * 9 * sizeof(double)));
    THX_RC(dCol.upload(iCol, npxl * sizeof(int)));
    THX_RC(dRow.upload(iRow, npxl * sizeof(int)));
    THX_RC(thx_project_dev(dVol.as<float>(), dOut.as<float>(), dMat.as<double>(), dCol.as<int>(), dRow.as<int>(), nR, pf,
                           vdim, npxl, nullptr));
    THX_CHECK(hipMemcpy(rotP, dOut.p, (size_t)nR * npxl * 2 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int thx_ExpectRotran_host(float* traP, const double* trans, const double* rot, double* rotMat, const int* iCol,
                          const int* iRow, int nR, int nT, int idim, int npxl)
{
    DevBuf dTra, dTrans, dQuat, dMat, dCol, dRow;
    THX_RC(dTra.alloc((size_t)nT * npxl * 2 * sizeof(float)));
    THX_RC(dTrans.upload(trans, (size_t)nT * 2 * sizeof(double)));
    THX_RC(dQuat.upload(rot, (size_t)nR * 4 * sizeof(double)));
    THX_RC(dMat.alloc((size_t)nR * 9 * sizeof(double)));
    THX_RC(dCol.upload(iCol, npxl * sizeof(int)));
    THX_RC(dRow.upload(iRow, npxl * sizeof(int)));
    THX_RC(thx_translate_dev(dTra.as<float>(), dTrans.as<double>(), nT, dCol.as<int>(), dRow.as<int>(), npxl, idim, nullptr));
    THX_RC(thx_rotmat_dev(dQuat.as<double>(), dMat.as<double>(), nR, nullptr));
    THX_CHECK(hipMemcpy(traP, dTra.p, (size_t)nT * npxl * 2 * sizeof(float), hipMemcpyDeviceToHost));
    THX_CHECK(hipMemcpy(rotMat, dMat.p, (size_t)nR * 9 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

static int insert_ft_host(thx_comm* hemi, int maxRadius, float* F3D, float* T3D_complex, double* O3D, int* counter, const float* datP,
                          const float* ctfP, const thx_ctf_attr* ctfaData, const double* offS, const float* w, const double* nR,
                          const double* nT, const double* nD, const int* nC, const int* iCol, const int* iRow,
                          float pixelSize, int cSearch, int opf, int npxl, int mReco, int idim, int vdim, int nK,
                          int imgNum)
{
    THX_REQUIRE(F3D && T3D_complex && datP && ctfP && w && nR && nT && iCol && iRow, "NULL pointer");
    const size_t nvol = (size_t)vdim * vdim * (vdim / 2 + 1);
    const size_t nd = (size_t)imgNum * mReco;
    DevBuf dF, dT, dTc, dO, dCnt, dDat, dCtf, dW, dQuat, dMat, dTran, dOff, dCls, dAttr, dDf, dCol, dRow;
    THX_RC(dF.upload(F3D, nvol * nK * 2 * sizeof(float)));
    THX_RC(upload_T(dTc, dT, T3D_complex, nvol * nK));   // T: complex on the host (imaginary part unused) -> real on the device
    double O0[3] = {0, 0, 0};
    int c0 = 0;
    THX_RC(dO.upload(O0, sizeof(O0)));
    THX_RC(dCnt.upload(&c0, sizeof(int)));
    THX_RC(dDat.upload(datP, (size_t)imgNum * npxl * 2 * sizeof(float)));
    THX_RC(dCtf.upload(ctfP, (size_t)imgNum * npxl * sizeof(float)));
    THX_RC(dW.upload(w, imgNum * sizeof(float)));
    THX_RC(dQuat.upload(nR, nd * 4 * sizeof(double)));
    THX_RC(dMat.alloc(nd * 9 * sizeof(double)));
    THX_RC(dTran.upload(nT, nd * 2 * sizeof(double)));
    if (offS) THX_RC(dOff.upload(offS, (size_t)imgNum * 2 * sizeof(double)));
    if (nC) THX_RC(dCls.upload(nC, nd * sizeof(int)));
    if (cSearch) {
        THX_REQUIRE(ctfaData && nD, "cSearch needs ctfaData and nD");
        THX_RC(dAttr.upload(ctfaData, imgNum * sizeof(thx_ctf_attr)));
        THX_RC(dDf.upload(nD, nd * sizeof(double)));
    }
    THX_RC(dCol.upload(iCol, npxl * sizeof(int)));
    THX_RC(dRow.upload(iRow, npxl * sizeof(int)));
    THX_RC(thx_rotmat_dev(dQuat.as<double>(), dMat.as<double>(), (int)nd, nullptr));
    THX_RC(thx_insert_dev(dF.as<float>(), dT.as<float>(), dO.as<double>(), dCnt.as<int>(), vdim, nK, dDat.as<float>(),
                          dCtf.as<float>(), dW.as<float>(), dMat.as<double>(), dTran.as<double>(),
                          offS ? dOff.as<double>() : nullptr, nC ? dCls.as<int>() : nullptr,
                          cSearch ? dAttr.as<thx_ctf_attr>() : nullptr, cSearch ? dDf.as<double>() : nullptr, cSearch,
                          pixelSize, dCol.as<int>(), dRow.as<int>(), opf, npxl, mReco, idim, imgNum, nullptr));
    if (hemi && thx_comm_size(hemi) > 1) {   // the reference's ncclAllReduce of F, T, O, counter (gpu/src/cuthunder.cu:4972-5067)
        DevBuf ws;
        THX_RC(ws.alloc(thx_reco_allreduce_workspace(vdim, maxRadius, opf)));
        for (int k = 0; k < nK; k++)
            THX_RC(thx_reco_allreduce(hemi, dF.as<float>() + (size_t)k * nvol * 2, dT.as<float>() + (size_t)k * nvol,
                                      k == 0 ? dO.as<double>() : nullptr, k == 0 ? dCnt.as<int>() : nullptr, vdim, maxRadius, opf,
                                      ws.p, nullptr));
        THX_CHECK(hipDeviceSynchronize());
    }
    THX_CHECK(hipMemcpy(F3D, dF.p, nvol * nK * 2 * sizeof(float), hipMemcpyDeviceToHost));
    THX_RC(download_T(T3D_complex, dTc, dT.as<float>(), nvol * nK, 0));   // (imaginary parts as the caller had them)
    if (O3D) {
        double o[3];
        THX_CHECK(hipMemcpy(o, dO.p, sizeof(o), hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; i++) O3D[i] += o[i];
    }
    if (counter) {
        int c;
        THX_CHECK(hipMemcpy(&c, dCnt.p, sizeof(int), hipMemcpyDeviceToHost));
        counter[0] += c;
    }
    return 0;
}

int thx_InsertFT_host(float* F3D, float* T3D_complex, double* O3D, int* counter, const float* datP, const float* ctfP,
                      const thx_ctf_attr* ctfaData, const double* offS, const float* w, const double* nR,
                      const double* nT, const double* nD, const int* nC, const int* iCol, const int* iRow,
                      float pixelSize, int cSearch, int opf, int npxl, int mReco, int idim, int vdim, int nK,
                      int imgNum)
{
    return insert_ft_host(nullptr, 0, F3D, T3D_complex, O3D, counter, datP, ctfP, ctfaData, offS, w, nR, nT, nD, nC, iCol, iRow,
                          pixelSize, cSearch, opf, npxl, mReco, idim, vdim, nK, imgNum);
}

int thx_InsertFT_hemi_host(thx_comm* hemi, int maxRadius, float* F3D, float* T3D_complex, double* O3D, int* counter,
                           const float* datP, const float* ctfP, const thx_ctf_attr* ctfaData, const double* offS, const float* w,
                           const double* nR, const double* nT, const double* nD, const int* nC, const int* iCol,
                           const int* iRow, float pixelSize, int cSearch, int opf, int npxl, int mReco, int idim, int vdim,
                           int nK, int imgNum)
{
    THX_REQUIRE(!hemi || maxRadius > 0, "maxRadius must be given with a communicator");
    return insert_ft_host(hemi, maxRadius, F3D, T3D_complex, O3D, counter, datP, ctfP, ctfaData, offS, w, nR, nT, nD, nC, iCol, iRow,
                          pixelSize, cSearch, opf, npxl, mReco, idim, vdim, nK, imgNum);
}

int thx_PrepareTF_host(int gpuIdx, float* F3D, float* T3D_complex, int vdim, const double* symMat,
                       int nSymmetryElement, int maxRadius, int pf)
{
    THX_CHECK(hipSetDevice(gpuIdx));
    const size_t nvol = (size_t)vdim * vdim * (vdim / 2 + 1);
    DevBuf dF, dT, dTc, dF2, dT2;
    THX_RC(dF.upload(F3D, nvol * 2 * sizeof(float)));
    THX_RC(upload_T(dTc, dT, T3D_complex, nvol));
    THX_RC(dF2.alloc(nvol * 2 * sizeof(float)));
    THX_RC(dT2.alloc(nvol * sizeof(float)));
    THX_RC(thx_normalise_tf_dev(dF.as<float>(), dT.as<float>(), vdim, nullptr));
    const double r = (double)(maxRadius * pf + 1);
    THX_RC(thx_symmetrize_dev(dT2.as<float>(), dT.as<float>(), vdim, 0, symMat, nSymmetryElement, r, nullptr));
    THX_RC(thx_symmetrize_dev(dF2.as<float>(), dF.as<float>(), vdim, 1, symMat, nSymmetryElement, r, nullptr));
    THX_CHECK(hipMemcpy(F3D, dF2.p, nvol * 2 * sizeof(float), hipMemcpyDeviceToHost));
    THX_RC(download_T(T3D_complex, dTc, dT2.as<float>(), nvol, 1));
    return 0;
}

int thx_ReconstructG_host(int gpuIdx, const float* F3D, const float* T3D_complex, int size, int N, int pf,
                          int maxRadius, float a, float alpha, const float* FSC, int nFSC, int joinHalf, int MAP,
                          int gridCorr, float* dstRL)
{
    THX_CHECK(hipSetDevice(gpuIdx));
    const int PF = pf * size;
    const size_t nvol = (size_t)PF * PF * (PF / 2 + 1);
    DevBuf dF, dT, dTc, dDst;
    THX_RC(dF.upload(F3D, nvol * 2 * sizeof(float)));
    THX_RC(upload_T(dTc, dT, T3D_complex, nvol));
    THX_RC(dDst.alloc((size_t)N * N * N * sizeof(float)));
    // the plan (W, C, scratch, the FFT plans: tens of milliseconds to build at 512^3) is kept between calls -- the reference calls
    // reconstructG four times per iteration with the same geometry (src/Optimiser.cpp:7366-7371,7600-7620); one cached plan per process,
    // rebuilt when the geometry or the device changes, calls serialised on it
    static std::mutex mtx;
    static thx_reco* cached = nullptr;
    static int key[4] = {0, 0, 0, -1};
    static float keyf[2] = {0.f, 0.f};
    std::lock_guard<std::mutex> g(mtx);
    if (!cached || key[0] != size || key[1] != N || key[2] != pf || key[3] != gpuIdx || keyf[0] != a || keyf[1] != alpha) {
        if (cached) { (void)thx_reco_destroy(cached); cached = nullptr; }
        THX_RC(thx_reco_create(&cached, size, N, pf, a, alpha));
        key[0] = size; key[1] = N; key[2] = pf; key[3] = gpuIdx; keyf[0] = a; keyf[1] = alpha;
    }
    int rc = thx_reco_reconstruct_dev(cached, dF.as<float>(), dT.as<float>(), maxRadius, FSC, nFSC, joinHalf, MAP, gridCorr,
                                      dDst.as<float>(), nullptr, nullptr, nullptr);
    if (!rc) {
        hipError_t e = hipMemcpy(dstRL, dDst.p, (size_t)N * N * N * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("hipMemcpy D2H failed: %s", hipGetErrorString(e)); rc = (int)e; }
    }
    return rc;
}

int thx_ReMask_host(float* const* imgFT, float maskRadius, float pixelSize, float ew, int idim, int imgNum)
{
    THX_REQUIRE(imgFT && idim > 0 && imgNum >= 0, "bad arguments");
    const size_t imgBytes = (size_t)idim * (idim / 2 + 1) * 2 * sizeof(float);
    const int kChunk = 1024;  // images staged per round trip (the reference streams BUFF_SIZE images per stream)
    DevBuf d;
    THX_RC(d.alloc((size_t)(imgNum < kChunk ? imgNum : kChunk) * imgBytes));
    for (int b = 0; b < imgNum; b += kChunk) {
        const int nb = imgNum - b < kChunk ? imgNum - b : kChunk;
        for (int l = 0; l < nb; l++)
            THX_CHECK(hipMemcpyAsync(d.as<char>() + (size_t)l * imgBytes, imgFT[b + l], imgBytes, hipMemcpyHostToDevice,
                                     nullptr));
        THX_RC(thx_remask_dev(d.as<float>(), nb, idim, maskRadius / pixelSize, ew, nullptr));
        for (int l = 0; l < nb; l++)
            THX_CHECK(hipMemcpyAsync(imgFT[b + l], d.as<char>() + (size_t)l * imgBytes, imgBytes, hipMemcpyDeviceToHost,
                                     nullptr));
        THX_CHECK(hipStreamSynchronize(nullptr));
    }
    return 0;
}

int thx_TranslateI2D_host(int gpuIdx, float* imgFT, double ox, double oy, int r, int idim)
{
    THX_REQUIRE(imgFT && idim > 0 && r >= 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    const size_t bytes = (size_t)idim * (idim / 2 + 1) * 2 * sizeof(float);
    const double t[2] = {ox, oy};
    DevBuf d, dt;
    THX_RC(d.upload(imgFT, bytes));
    THX_RC(dt.upload(t, sizeof(t)));
    THX_RC(thx_translate_image_dev(d.as<float>(), d.as<float>(), dt.as<double>(), 1, idim, (float)r, nullptr));
    THX_CHECK(hipMemcpy(imgFT, d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int thx_TranslateI_host(int gpuIdx, float* volFT, double ox, double oy, double oz, int r, int dim)
{
    THX_REQUIRE(volFT && dim > 0 && r >= 0, "bad arguments");
    THX_CHECK(hipSetDevice(gpuIdx));
    const size_t bytes = (size_t)dim * dim * (dim / 2 + 1) * 2 * sizeof(float);
    DevBuf d;
    THX_RC(d.upload(volFT, bytes));
    THX_RC(thx_translate_volume_dev(d.as<float>(), d.as<float>(), dim, (float)r, ox, oy, oz, nullptr));
    THX_CHECK(hipMemcpy(volFT, d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int thx_ExpectPrecal_host(const thx_ctf_attr* ctfAttr, float* def, float* k1, float* k2, const int* iCol,
                          const int* iRow, int idim, int npxl, int imgNum)
{
    THX_REQUIRE(ctfAttr && def && k1 && k2 && iCol && iRow, "NULL pointer");
    if (imgNum <= 0 || npxl <= 0) return 0;
    DevBuf dA, dDef, dK1, dK2, dC, dR;
    THX_RC(dA.upload(ctfAttr, (size_t)imgNum * sizeof(thx_ctf_attr)));
    THX_RC(dC.upload(iCol, (size_t)npxl * sizeof(int)));
    THX_RC(dR.upload(iRow, (size_t)npxl * sizeof(int)));
    THX_RC(dDef.alloc((size_t)imgNum * npxl * sizeof(float)));
    THX_RC(dK1.alloc((size_t)imgNum * sizeof(float)));
    THX_RC(dK2.alloc((size_t)imgNum * sizeof(float)));
    THX_RC(thx_expect_precal_dev(nullptr, dDef.as<float>(), dK1.as<float>(), dK2.as<float>(), dA.as<thx_ctf_attr>(), idim,
                                 1.0f, dC.as<int>(), dR.as<int>(), npxl, imgNum, nullptr));
    THX_CHECK(hipMemcpy(def, dDef.p, (size_t)imgNum * npxl * sizeof(float), hipMemcpyDeviceToHost));
    THX_CHECK(hipMemcpy(k1, dK1.p, (size_t)imgNum * sizeof(float), hipMemcpyDeviceToHost));
    THX_CHECK(hipMemcpy(k2, dK2.p, (size_t)imgNum * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int thx_ExpectGlobal3D_host(const float* rotP, const float* traP, const float* datP, const float* ctfP,
                            const float* sigRcpP, float* wC, float* wR, float* wT, const double* pR, const double* pT,
                            float* baseL, int kIdx, int nK, int nR, int nT, int npxl, int imgNum)
{
    THX_REQUIRE(rotP && traP && datP && ctfP && sigRcpP && wC && wR && wT && pR && pT && baseL, "NULL pointer");
    if (imgNum <= 0) return 0;
    const size_t nI = (size_t)imgNum;
    DevBuf dRot, dTra, dDat, dCtf, dSig, dWC, dWR, dWT, dPR, dPT, dBase, dWs;
    THX_RC(dRot.upload(rotP, (size_t)nR * npxl * 2 * sizeof(float)));
    THX_RC(dTra.upload(traP, (size_t)nT * npxl * 2 * sizeof(float)));
    THX_RC(dDat.upload(datP, nI * npxl * 2 * sizeof(float)));
    THX_RC(dCtf.upload(ctfP, nI * npxl * sizeof(float)));
    THX_RC(dSig.upload(sigRcpP, nI * npxl * sizeof(float)));
    THX_RC(dWC.upload(wC, nI * nK * sizeof(float)));
    THX_RC(dWR.upload(wR, (size_t)nK * nI * nR * sizeof(float)));
    THX_RC(dWT.upload(wT, (size_t)nK * nI * nT * sizeof(float)));
    THX_RC(dPR.upload(pR, nI * nR * sizeof(double)));
    THX_RC(dPT.upload(pT, nI * nT * sizeof(double)));
    THX_RC(dBase.upload(baseL, nI * sizeof(float)));
    // images in slabs of <= 65535 (grid.y limit of the scan kernel)
    for (int l0 = 0; l0 < imgNum; l0 += 65535) {
        const int nl = imgNum - l0 < 65535 ? imgNum - l0 : 65535;
        DevBuf ws;
        THX_RC(ws.alloc(thx_expect_global_workspace(nl, nR, nT)));
        // wR / wT are [nK][imgNum][.]: a slab of images is not contiguous across classes, so only kIdx's plane is offset
        THX_REQUIRE(l0 == 0 || nK == 1, "more than 65535 images per call needs nK == 1");
        THX_RC(thx_expect_global_dev(dRot.as<float>(), dTra.as<float>(), dDat.as<float>() + (size_t)l0 * npxl * 2,
                                     dCtf.as<float>() + (size_t)l0 * npxl, dSig.as<float>() + (size_t)l0 * npxl,
                                     dPR.as<double>() + (size_t)l0 * nR, dPT.as<double>() + (size_t)l0 * nT,
                                     dWC.as<float>() + (size_t)l0 * nK, dWR.as<float>() + (size_t)l0 * nR,
                                     dWT.as<float>() + (size_t)l0 * nT, dBase.as<float>() + l0, kIdx, nK, nR, nT, npxl, nl,
                                     ws.p, nullptr));
        THX_CHECK(hipStreamSynchronize(nullptr));
    }
    THX_CHECK(hipMemcpy(wC, dWC.p, nI * nK * sizeof(float), hipMemcpyDeviceToHost));
    THX_CHECK(hipMemcpy(wR, dWR.p, (size_t)nK * nI * nR * sizeof(float), hipMemcpyDeviceToHost));
    THX_CHECK(hipMemcpy(wT, dWT.p, (size_t)nK * nI * nT * sizeof(float), hipMemcpyDeviceToHost));
    THX_CHECK(hipMemcpy(baseL, dBase.p, nI * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int thx_GCTFinit_host(float* const* ctfFT, const thx_ctf_attr* ctfAttr, float pixelSize, int idim, int imgNum)
{
    THX_REQUIRE(ctfFT && ctfAttr && idim > 0, "bad arguments");
    if (imgNum <= 0) return 0;
    const size_t imgBytes = (size_t)idim * (idim / 2 + 1) * 2 * sizeof(float);
    const int kChunk = 1024;
    DevBuf d, dA;
    THX_RC(dA.upload(ctfAttr, (size_t)imgNum * sizeof(thx_ctf_attr)));
    THX_RC(d.alloc((size_t)(imgNum < kChunk ? imgNum : kChunk) * imgBytes));
    for (int b = 0; b < imgNum; b += kChunk) {
        const int nb = imgNum - b < kChunk ? imgNum - b : kChunk;
        THX_RC(thx_ctf_image_dev(d.as<float>(), dA.as<thx_ctf_attr>() + b, pixelSize, idim, nb, nullptr));
        for (int l = 0; l < nb; l++)
            THX_CHECK(hipMemcpyAsync(ctfFT[b + l], d.as<char>() + (size_t)l * imgBytes, imgBytes, hipMemcpyDeviceToHost,
                                     nullptr));
        THX_CHECK(hipStreamSynchronize(nullptr));
    }
    return 0;
}

}  // extern "C"
