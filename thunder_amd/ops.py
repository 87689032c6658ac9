"""Torch-tensor front-end of the C ABI: thin argument marshalling only (no arithmetic happens here).

Every function takes CUDA(HIP) tensors already resident in HBM and enqueues on torch's current stream.
Names follow the reference (thuem/THUNDER): project, translate, CTF, logDataVSPrior, insert, prepareTF ...
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import CtfAttr, ptr, stream_ptr

_F32, _F64, _I32, _C64 = torch.float32, torch.float64, torch.int32, torch.complex64


def _chk(t, dtype, name):
    assert t.is_cuda, name + " must be a device tensor"
    assert t.dtype == dtype, "%s must be %s, got %s" % (name, dtype, t.dtype)
    assert t.is_contiguous(), name + " must be contiguous"
    return t


def device_count():
    n = C.c_int(0)
    capi.call("thx_device_count", C.byref(n))
    return n.value


def ctf_attr_tensor(attrs, device):
    """list of 7-tuples (voltage, defocusU, defocusV, theta, Cs, ampContrast, phaseShift) -> [n][7] float32"""
    a = np.asarray(attrs, dtype=np.float32).reshape(-1, 7)
    return torch.from_numpy(a).to(device)


def rotmat(quat):
    """rotate3D(dmat33&, dvec4) src/Geometry/Euler.cpp:181-189: [n][4] f64 -> [n][9] f64 column-major"""
    _chk(quat, _F64, "quat")
    n = quat.numel() // 4
    out = torch.empty((n, 9), dtype=_F64, device=quat.device)
    capi.call("thx_rotmat_dev", ptr(quat), ptr(out), n, stream_ptr())
    return out


def translate(trans, iCol, iRow, idim):
    """translate() src/Image/ImageFunctions.cpp:233-252: trans [nT][2] f64 -> [nT][nPxl] complex64"""
    _chk(trans, _F64, "trans"); _chk(iCol, _I32, "iCol"); _chk(iRow, _I32, "iRow")
    nT, nPxl = trans.numel() // 2, iCol.numel()
    out = torch.empty((nT, nPxl), dtype=_C64, device=trans.device)
    capi.call("thx_translate_dev", ptr(out), ptr(trans), nT, ptr(iCol), ptr(iRow), nPxl, idim, stream_ptr())
    return out


def ctf(attr, pixelSize, iCol, iRow, idim, dfac=None):
    """CTF(RFLOAT* dst, ...) src/CTF.cpp:113-151 for nImg images: attr [nImg][7] f32 -> [nImg][nPxl] f32"""
    _chk(attr, _F32, "attr"); _chk(iCol, _I32, "iCol"); _chk(iRow, _I32, "iRow")
    nImg, nPxl = attr.shape[0], iCol.numel()
    if dfac is not None:
        _chk(dfac, _F64, "dfac")
    out = torch.empty((nImg, nPxl), dtype=_F32, device=attr.device)
    capi.call("thx_ctf_dev", ptr(out), ptr(attr), ptr(dfac), float(pixelSize), ptr(iCol), ptr(iRow), nPxl, idim, nImg,
              stream_ptr())
    return out


def expect_precal(attr, pixelSize, iCol, iRow, idim):
    """allocPreCal ctf=true branch (src/Optimiser.cpp:8124-8169 / ExpectPrecal) -> freq, def, k1, k2"""
    _chk(attr, _F32, "attr"); _chk(iCol, _I32, "iCol"); _chk(iRow, _I32, "iRow")
    nImg, nPxl, dev = attr.shape[0], iCol.numel(), attr.device
    freq = torch.empty(nPxl, dtype=_F32, device=dev)
    de = torch.empty((nImg, nPxl), dtype=_F32, device=dev)
    k1 = torch.empty(nImg, dtype=_F32, device=dev)
    k2 = torch.empty(nImg, dtype=_F32, device=dev)
    capi.call("thx_expect_precal_dev", ptr(freq), ptr(de), ptr(k1), ptr(k2), ptr(attr), idim, float(pixelSize),
              ptr(iCol), ptr(iRow), nPxl, nImg, stream_ptr())
    return freq, de, k1, k2


def ctf_dsearch(freq, de, k1, k2, attr, dpara):
    """defocus-search CTF rows (src/Optimiser.cpp:1246-1272): dpara [nImg][nD] f64 -> [nImg][nD][nPxl] f32"""
    _chk(dpara, _F64, "dpara")
    nImg, nD, nPxl = dpara.shape[0], dpara.shape[1], freq.numel()
    out = torch.empty((nImg, nD, nPxl), dtype=_F32, device=freq.device)
    capi.call("thx_ctf_dsearch_dev", ptr(out), ptr(freq), ptr(de), ptr(k1), ptr(k2), ptr(attr), ptr(dpara), nD, nPxl,
              nImg, stream_ptr())
    return out


def ctf_image(attr, pixelSize, idim):
    """CTF(Image&, ...) src/CTF.cpp:31-66 for nImg images -> [nImg][idim][idim/2+1] complex64"""
    _chk(attr, _F32, "attr")
    out = torch.empty((attr.shape[0], idim, idim // 2 + 1), dtype=_C64, device=attr.device)
    capi.call("thx_ctf_image_dev", ptr(out), ptr(attr), float(pixelSize), idim, attr.shape[0], stream_ptr())
    return out


def gather_pixels(img, iPxl, idim, out=None):
    """allocPreCal gather src/Optimiser.cpp:8055-8075: img [nImg][idim][idim/2+1] c64 -> [nImg][nPxl] c64"""
    _chk(img, _C64, "img"); _chk(iPxl, _I32, "iPxl")
    nImg, nPxl = img.shape[0], iPxl.numel()
    if out is None:
        out = torch.empty((nImg, nPxl), dtype=_C64, device=img.device)
    else:
        _chk(out, _C64, "out")
    capi.call("thx_gather_pixels_dev", ptr(out), ptr(img), ptr(iPxl), nPxl, idim, nImg, stream_ptr())
    return out


def project(volume, rotMat, iCol, iRow, pf, out=None):
    """Projector::project src/Projector.cpp:356-374 for [nR][9] matrices -> [nR][nPxl] complex64"""
    _chk(volume, _C64, "volume"); _chk(rotMat, _F64, "rotMat"); _chk(iCol, _I32, "iCol"); _chk(iRow, _I32, "iRow")
    vdim = volume.shape[0]
    nR, nPxl = rotMat.numel() // 9, iCol.numel()
    if out is None:
        out = torch.empty((nR, nPxl), dtype=_C64, device=volume.device)
    capi.call("thx_project_dev", ptr(volume), ptr(out), ptr(rotMat), ptr(iCol), ptr(iRow), nR, pf, vdim, nPxl,
              stream_ptr())
    return out


def logDataVSPrior(dat, pri, ctf_, sigRcp):
    """logDataVSPrior_m_huabin src/Optimiser.cpp:9187-9213 for pri rows [n][nPxl] against one image row"""
    _chk(dat, _C64, "dat"); _chk(pri, _C64, "pri"); _chk(ctf_, _F32, "ctf"); _chk(sigRcp, _F32, "sigRcp")
    nPxl = dat.numel()
    n = pri.numel() // nPxl
    out = torch.empty(n, dtype=_F32, device=dat.device)
    capi.call("thx_logdatavsprior_dev", ptr(out), ptr(dat), ptr(pri), ptr(ctf_), ptr(sigRcp), n, nPxl, stream_ptr())
    return out


class ExpectLocalResult:
    __slots__ = ("wC", "wR", "wT", "wD", "baseLine", "logW")


def pack_projector(volumes, vdim):
    """cell-packed copies of [nVol] projector volumes (thx_projector_pack_dev): float32 [nVol][P][P][P/2+1][16]"""
    _chk(volumes, _C64, "volumes")
    nVol = volumes.numel() // (vdim * vdim * (vdim // 2 + 1))
    cells = torch.empty((nVol, vdim, vdim, vdim // 2 + 1, 16), dtype=_F32, device=volumes.device)
    capi.call("thx_projector_pack_dev", ptr(cells), ptr(volumes), vdim, nVol, stream_ptr())
    return cells


def expect_local(volumes, vdim, pf, idim, iCol, iRow, datP, ctfP, sigRcpP, rotMat, trans, nD=1, volIdx=None, pC=None,
                 pR=None, pT=None, pD=None, want_logW=False, workspace=None, packed=False, wg_per_cu=-1, active=None):
    """One particle-filter phase for a batch of images (src/Optimiser.cpp:1225-1406); see thunder_amd.h.
    packed=True: `volumes` is the output of pack_projector.  wg_per_cu: occupancy argument of the C ABI (0 = unlimited,
    negative = library default)."""
    dev = datP.device
    _chk(volumes, _F32 if packed else _C64, "volumes"); _chk(datP, _C64, "datP"); _chk(ctfP, _F32, "ctfP"); _chk(sigRcpP, _F32, "sigRcpP")
    _chk(rotMat, _F64, "rotMat"); _chk(trans, _F64, "trans")
    nImg, nPxl = datP.shape[0], datP.shape[1]
    nR = rotMat.numel() // (9 * nImg)
    nT = trans.numel() // (2 * nImg)
    ones = lambda *s: torch.ones(s, dtype=_F64, device=dev)
    pC = ones(nImg) if pC is None else _chk(pC, _F64, "pC")
    pR = ones(nImg, nR) if pR is None else _chk(pR, _F64, "pR")
    pT = ones(nImg, nT) if pT is None else _chk(pT, _F64, "pT")
    pD = ones(nImg, nD) if pD is None else _chk(pD, _F64, "pD")
    if volIdx is not None:
        _chk(volIdx, _I32, "volIdx")
    res = ExpectLocalResult()
    res.wC = torch.empty(nImg, dtype=_F32, device=dev)
    res.wR = torch.empty((nImg, nR), dtype=_F32, device=dev)
    res.wT = torch.empty((nImg, nT), dtype=_F32, device=dev)
    res.wD = torch.empty((nImg, nD), dtype=_F32, device=dev)
    res.baseLine = torch.empty(nImg, dtype=_F32, device=dev)
    res.logW = torch.empty((nImg, nD, nT, nR), dtype=_F32, device=dev) if want_logW else None
    need = capi.load().thx_expect_local_workspace(nImg, nR, nT, nD)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    capi.call("thx_expect_local_packed_dev" if packed else "thx_expect_local_dev", ptr(volumes), ptr(volIdx), vdim, pf, idim, ptr(iCol), ptr(iRow), nPxl, nImg,
              ptr(datP), ptr(ctfP), ptr(sigRcpP), ptr(rotMat), nR, ptr(trans), nT, nD, ptr(pC), ptr(pR), ptr(pT),
              ptr(pD), ptr(res.wC), ptr(res.wR), ptr(res.wT), ptr(res.wD), ptr(res.baseLine), ptr(res.logW),
              ptr(workspace), int(wg_per_cu), ptr(active), stream_ptr())
    return res


def expect_global(rotP, traP, datP, ctfP, sigRcpP, pR, pT, wC, wR, wT, baseL, kIdx, nK, workspace=None):
    """Scanning phase for class kIdx (src/Optimiser.cpp:756-894); wC/wR/wT/baseL updated in place."""
    nR, nPxl = rotP.shape
    nT = traP.shape[0]
    nImg = datP.shape[0]
    need = capi.load().thx_expect_global_workspace(nImg, nR, nT)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=datP.device)
    capi.call("thx_expect_global_dev", ptr(rotP), ptr(traP), ptr(datP), ptr(ctfP), ptr(sigRcpP), ptr(pR), ptr(pT),
              ptr(wC), ptr(wR), ptr(wT), ptr(baseL), kIdx, nK, nR, nT, nPxl, nImg, ptr(workspace), stream_ptr())


def insert(F, T, dim, datP, ctfP, w, rotMat, trans, iCol, iRow, opf, idim, O=None, counter=None, offS=None, cls=None,
           attr=None, dfac=None, cSearch=False, pixelSize=1.0, nK=1):
    """HOT LOOP C / Reconstructor::insertP for a batch (src/Optimiser.cpp:7038-7241); F, T accumulated in place."""
    _chk(F, _C64, "F"); _chk(T, _F32, "T"); _chk(datP, _C64, "datP"); _chk(ctfP, _F32, "ctfP"); _chk(w, _F32, "w")
    _chk(rotMat, _F64, "rotMat"); _chk(trans, _F64, "trans")
    nImg, nPxl = datP.shape[0], datP.shape[1]
    mReco = rotMat.numel() // (9 * nImg)
    capi.call("thx_insert_dev", ptr(F), ptr(T), ptr(O), ptr(counter), dim, nK, ptr(datP), ptr(ctfP), ptr(w),
              ptr(rotMat), ptr(trans), ptr(offS), ptr(cls), ptr(attr), ptr(dfac), 1 if cSearch else 0,
              float(pixelSize), ptr(iCol), ptr(iRow), opf, nPxl, mReco, idim, nImg, stream_ptr())


def normalise_TF(F, T, dim):
    capi.call("thx_normalise_tf_dev", ptr(F), ptr(T), dim, stream_ptr())


def symmetrize(src, dim, symMat, r):
    """SYMMETRIZE_FT; symMat: host numpy [nSym][9] f64 column-major; returns a new tensor"""
    symMat = np.ascontiguousarray(np.asarray(symMat, dtype=np.float64).reshape(-1, 9))
    dst = torch.empty_like(src)
    capi.call("thx_symmetrize_dev", ptr(dst), ptr(src), dim, 1 if src.dtype == _C64 else 0,
              symMat.ctypes.data if len(symMat) else None, len(symMat), float(r), stream_ptr())
    return dst


def fsc(A, B, dim, nShell):
    out = torch.empty(nShell, dtype=_F32, device=A.device)
    capi.call("thx_fsc_dev", ptr(out), nShell, ptr(A), ptr(B), dim, stream_ptr())
    return out


def fft3d_fw(rl):
    n = rl.shape[0]
    ft = torch.empty((n, n, n // 2 + 1), dtype=_C64, device=rl.device)
    capi.call("thx_fft3d_fw_dev", ptr(rl), ptr(ft), n, stream_ptr())
    return ft


# ---------------------------------------------------------------------------------------------
# callers either side of the E/M loop
def fft3d_bw(ft, n):
    """FFT::bw (c2r incl. 1/size) of a half-complex [n][n][n/2+1] FT; the input is destroyed"""
    rl = torch.empty((n, n, n), dtype=torch.float32, device=ft.device)
    capi.call("thx_fft3d_bw_dev", ptr(ft), ptr(rl), n, stream_ptr())
    return rl


def compare_hemispheres(A, B, N, rU, fsc=True, coreR=0.0, ew=6.0, avg_r=None, seed=0, call_id=0):
    """Model::compareTwoHemispheres on two half-map FTs (in place for the averaging) -> FSC [rU] (numpy) or None"""
    import numpy as np
    out = np.zeros(rU, np.float32) if fsc else None
    capi.call("thx_compare_hemispheres_dev", ptr(A), ptr(B), N, rU, out.ctypes.data if fsc else None, None, float(coreR), float(ew),
         0 if avg_r is None else 1, 0 if avg_r is None else int(avg_r), int(seed), int(call_id), None, stream_ptr())
    return out


def soft_mask_volume(vol, r, ew=6.0, bg=0.0):
    capi.call("thx_soft_mask_volume_dev", ptr(vol), vol.shape[0], float(r), float(ew), float(bg), stream_ptr())
    return vol


def remask(imgFT, maskRadiusPx, ew=6.0):
    """Optimiser::reMaskImg src/Optimiser.cpp:6093-6149 (ReMask Interface.h:517) IN PLACE on [nImg][N][N/2+1] c64"""
    _chk(imgFT, _C64, "imgFT")
    nImg, idim = imgFT.shape[0], imgFT.shape[1]
    capi.call("thx_remask_dev", ptr(imgFT), nImg, idim, float(maskRadiusPx), float(ew), stream_ptr())
    return imgFT


def translate_image(src, trans, r=-1.0, out=None):
    """translate(Image&, const Image&, [r,] tx, ty) src/Image/ImageFunctions.cpp:269-284 / :322-339; trans [nImg][2] f64"""
    _chk(src, _C64, "src"); _chk(trans, _F64, "trans")
    nImg, idim = src.shape[0], src.shape[1]
    if out is None:
        out = src.clone() if r >= 0 else torch.empty_like(src)
    capi.call("thx_translate_image_dev", ptr(out), ptr(src), ptr(trans), nImg, idim, float(r), stream_ptr())
    return out


def translate_volume(vol, r, ox, oy, oz):
    """translate(Volume&, const Volume&, r, tx, ty, tz) src/Image/ImageFunctions.cpp:363-384, in place (TranslateI)"""
    _chk(vol, _C64, "vol")
    capi.call("thx_translate_volume_dev", ptr(vol), ptr(vol), vol.shape[0], float(r), float(ox), float(oy), float(oz),
              stream_ptr())
    return vol


def sigma_spectra(volumes, vdim, pf, projR, rSig, img, imgOri, attr, pixelSize, rotMat, trans, offset=None,
                  dfac=None, volIdx=None, packed=False):
    """Per-image shell spectra of Optimiser::allReduceSigma (src/Optimiser.cpp:6443-6565) -> [nImg][4][rSig] f32.
    packed=True: `volumes` is the output of pack_projector (thx_sigma_spectra_packed_dev)."""
    _chk(volumes, _F32 if packed else _C64, "volumes"); _chk(img, _C64, "img"); _chk(imgOri, _C64, "imgOri"); _chk(attr, _F32, "attr")
    _chk(rotMat, _F64, "rotMat"); _chk(trans, _F64, "trans")
    nImg, idim = img.shape[0], img.shape[1]
    spec = torch.empty((nImg, 4, rSig), dtype=_F32, device=img.device)
    capi.call("thx_sigma_spectra_packed_dev" if packed else "thx_sigma_spectra_dev", ptr(spec), ptr(volumes), ptr(volIdx), vdim, pf, idim, projR, rSig, ptr(img),
              ptr(imgOri), ptr(attr), ptr(dfac), float(pixelSize), ptr(rotMat), ptr(trans), ptr(offset), nImg,
              stream_ptr())
    return spec


def norm_residual(volumes, vdim, pf, projR, rL, rNorm, img, attr, pixelSize, rotMat, trans, volIdx=None, dfac=None, packed=False):
    """per-image part of Optimiser::normCorrection (thx_norm_residual_dev; packed=True: on pack_projector's cells): float32 [nImg]"""
    _chk(volumes, _F32 if packed else _C64, "volumes"); _chk(img, _C64, "img"); _chk(rotMat, _F64, "rotMat"); _chk(trans, _F64, "trans")
    nImg, idim = img.shape[0], img.shape[1]
    norm = torch.empty((nImg,), dtype=_F32, device=img.device)
    capi.call("thx_norm_residual_packed_dev" if packed else "thx_norm_residual_dev", ptr(norm), ptr(volumes), ptr(volIdx), vdim, pf, idim, int(projR), float(rL), float(rNorm), ptr(img),
              ptr(attr), ptr(dfac), float(pixelSize), ptr(rotMat), ptr(trans), nImg, stream_ptr())
    return norm


def median_f32(values):
    """median(vec, n) of the reference (gsl quantile 0.5 of the sorted data) on the device: a 1-element float32 tensor"""
    _chk(values, _F32, "values")
    out = torch.empty((1,), dtype=_F32, device=values.device)
    capi.call("thx_median_f32_dev", ptr(out), ptr(values), values.numel(), stream_ptr())
    return out


def norm_scale(img, imgOri, norm, median):
    """img[l] *= sqrt(median / norm[l]), imgOri[l] likewise, in place (thx_norm_scale_dev)"""
    _chk(img, _C64, "img"); _chk(imgOri, _C64, "imgOri"); _chk(norm, _F32, "norm"); _chk(median, _F32, "median")
    capi.call("thx_norm_scale_dev", ptr(img), ptr(imgOri), ptr(norm), ptr(median), img.shape[1], img.shape[0], stream_ptr())


def sigma_accum(acc, spec, groupID, group=True):
    """acc = (sigM, sigN, svd) device [nGroup][rSig+1] f32, updated in place; groupID host int32 (1-based)"""
    sigM, sigN, svd = acc
    nGroup, rSig = sigM.shape[0], sigM.shape[1] - 1
    g = None if groupID is None else np.ascontiguousarray(np.asarray(groupID, dtype=np.int32))
    capi.call("thx_sigma_accum_dev", ptr(sigM), ptr(sigN), ptr(svd), ptr(spec), ptr(g), spec.shape[0], nGroup, rSig,
              1 if group else 0, stream_ptr())


def sigma_final(acc, maskRadius, size, pixelSize, group=True):
    """closing arithmetic of allReduceSigma (src/Optimiser.cpp:6654-6707) -> sig, sigRcp [nGroup][rSig]"""
    sigM, sigN, svd = acc
    nGroup, rSig = sigM.shape[0], sigM.shape[1] - 1
    sig = torch.empty((nGroup, rSig), dtype=_F32, device=sigM.device)
    rcp = torch.empty_like(sig)
    capi.call("thx_sigma_final_dev", ptr(sig), ptr(rcp), ptr(sigM), ptr(sigN), ptr(svd), nGroup, rSig,
              1 if group else 0, float(maskRadius), int(size), float(pixelSize), stream_ptr())
    return sig, rcp


# ---------------------------------------------------------------------------------------------
# image ingestion
def init_images(imgRL, maskRadiusPx, ew=6.0, reduce_stats=None):
    """Optimiser::initImg after reading (src/Optimiser.cpp:4700-4800): substractBgImg -> statImg -> maskImg (zeroMask)
    -> normaliseImg -> fwImg for a DEVICE stack imgRL float32 [n][N][N] (in-memory layout; modified in place).
    reduce_stats(sums[4] float64 tensor, n) -> (sums, N) lets the caller all-reduce over the hemisphere.
    Returns imgFT, imgOriFT (complex64 [n][N][N/2+1]) and the statistics dict."""
    _chk(imgRL, _F32, "imgRL")
    n, N, dev = imgRL.shape[0], imgRL.shape[1], imgRL.device
    capi.call("thx_img_subtract_bg_dev", ptr(imgRL), n, N, float(maskRadiusPx), stream_ptr())
    stat = torch.empty((n, 4), dtype=_F64, device=dev)
    capi.call("thx_img_stats_dev", ptr(stat), ptr(imgRL), n, N, float(maskRadiusPx), stream_ptr())
    sums, cnt = stat.sum(0), n
    if reduce_stats is not None:
        sums, cnt = reduce_stats(sums, n)
    mean, stdN, stdD, q = [np.float32(x / cnt) for x in sums.cpu().numpy()]
    stdS = np.float32(stdD - stdN)
    stdStdN = np.float32(np.sqrt(max(0.0, float(q) - float(np.float32(float(stdN) ** 2)))))
    scale = np.float32(1.0 / np.float64(stdN))
    imgFT = torch.empty((n, N, N // 2 + 1), dtype=_C64, device=dev)
    oriFT = torch.empty_like(imgFT)
    scratch = torch.empty((min(n, 1024), N, N), dtype=_F32, device=dev)
    capi.call("thx_img_mask_normalise_fft_dev", ptr(imgFT), ptr(oriFT), ptr(imgRL), ptr(scratch), n, N,
              float(maskRadiusPx), float(ew), float(scale), stream_ptr())
    return imgFT, oriFT, dict(mean=float(mean), stdN=float(stdN), stdD=float(stdD), stdS=float(stdS),
                              stdStdN=float(stdStdN))


# ---------------------------------------------------------------------------------------------
# particle filter
def pf_ctx(img0=0, symQuat=None):
    """thx_pf_ctx: img0 = the launch's first image in the Philox numbering, symQuat = DEVICE [nSym][4] (None: C1)"""
    import ctypes as C
    c = capi.PfCtx()
    c.symQuat = ptr(symQuat) if symQuat is not None and symQuat.numel() else None
    c.nSym = 0 if symQuat is None else int(symQuat.shape[0])
    c.img0 = int(img0)
    return C.byref(c)


def pf_perturb(r, t, wR, wT, k123, s01, pfR, pfT, transS, transQ, seed, call, active=None, img0=0, symQuat=None):
    """Particle::perturb(pf, PAR_R) + perturb(pf, PAR_T) (src/Particle.cpp:1149-1272) for [n][nR][4] / [n][nT][2] f64"""
    for x, nm in ((r, "r"), (t, "t"), (wR, "wR"), (wT, "wT"), (k123, "k123"), (s01, "s01")):
        _chk(x, _F64, nm)
    capi.call("thx_pf_perturb_ex_dev", ptr(r), ptr(t), ptr(wR), ptr(wT), ptr(k123), ptr(s01), r.shape[0], r.shape[1],
              t.shape[1], float(pfR), float(pfT), float(transS), float(transQ), int(seed), int(call), ptr(active), pf_ctx(img0, symQuat),
              stream_ptr())


def pf_update(r, t, wR, wT, uR, uT, k123, s01, topR, topT, peakFactorR, seed, call, active=None, img0=0, symQuat=None):
    """setUR/UT, keepHalfHeightPeak, calRank1st, calVari, resample (src/Optimiser.cpp:1410-1475)"""
    _chk(uR, _F32, "uR"); _chk(uT, _F32, "uT")
    capi.call("thx_pf_update_ex_dev", ptr(r), ptr(t), ptr(wR), ptr(wT), ptr(uR), ptr(uT), ptr(k123), ptr(s01), ptr(topR),
              ptr(topT), r.shape[0], r.shape[1], t.shape[1], float(peakFactorR), int(seed), int(call), ptr(active), pf_ctx(img0, symQuat),
              stream_ptr())


def pf_cal_vari(r, t, seed, call, img0=0, symQuat=None):
    """Particle::calVari(PAR_R) + calVari(PAR_T) on their own (thx_pf_cal_vari_dev): returns k123 [n][3], s01 [n][2]; with a
    point group r is symmetrised in place"""
    n = r.shape[0]
    k = torch.zeros((n, 3), dtype=_F64, device=r.device)
    s = torch.zeros((n, 2), dtype=_F64, device=r.device)
    capi.call("thx_pf_cal_vari_dev", ptr(r), ptr(t), ptr(k), ptr(s), n, r.shape[1], t.shape[1], int(seed), int(call),
              pf_ctx(img0, symQuat), stream_ptr())
    return k, s


def draw_reco(r, t, mReco, seed, call, img0=0):
    """the mReco draws of the insertion from a resampled filter (thx_draw_reco_dev): r [n][nR][4], t [n][nT][2] f64 ->
    rotation matrices [n][mReco][9], shifts [n][mReco][2]"""
    _chk(r, _F64, "r"); _chk(t, _F64, "t")
    n, nR, nT = r.shape[0], r.shape[1], t.shape[1]
    rot = torch.empty((n, mReco, 9), dtype=_F64, device=r.device)
    tran = torch.empty((n, mReco, 2), dtype=_F64, device=r.device)
    capi.call("thx_draw_reco_dev", ptr(rot), ptr(tran), ptr(r), ptr(t), n, nR, nT, mReco, int(seed), int(call), int(img0),
              stream_ptr())
    return rot, tran


def pf_stop_init(n, transS, device, ctfRefineS=0.01):
    """state of the per-image stop rule (src/Optimiser.cpp:1168-1183): (active int32 [n], nP int32 [n], state f64 [n][8])"""
    active = torch.empty(n, dtype=_I32, device=device)
    nP = torch.empty(n, dtype=_I32, device=device)
    state = torch.empty((n, 8), dtype=_F64, device=device)
    capi.call("thx_pf_stop_init_dev", ptr(active), ptr(nP), ptr(state), float(transS), float(ctfRefineS), n, stream_ptr())
    return active, nP, state


def pf_stop_rule(active, nP, state, k123, s01, phase, sD=None):
    """src/Optimiser.cpp:1510-1615 after phase `phase`; returns the number of images still active"""
    cnt = torch.zeros(1, dtype=_I32, device=active.device)
    capi.call("thx_pf_stop_rule_dev", ptr(active), ptr(nP), ptr(state), ptr(k123), ptr(s01), ptr(sD), int(phase), active.numel(),
              ptr(cnt), stream_ptr())
    return int(cnt.item())


def pf_class_select(uC, seed, call_id, wC=None, peakFactorC=1.0 - 1e-2):
    """class of every image after the global scan (src/Optimiser.cpp:925-952): keepHalfHeightPeak(PAR_C), resample(k, PAR_C),
    Particle::rand(cls); uC [nImg][nK] f32 scan weights -> int32 [nImg]"""
    _chk(uC, _F32, "uC")
    nImg, nK = uC.shape
    cls = torch.empty((nImg,), dtype=_I32, device=uC.device)
    capi.call("thx_pf_class_select_dev", ptr(cls), ptr(uC), ptr(wC), nImg, nK, float(peakFactorC), int(seed), int(call_id),
              stream_ptr())
    return cls


def pf_scan_support(gridR, gridT, uR, uT, cls, mLR, mLT, peakFactorR, seed, call, minK=0.0, minS=0.0):
    """support points of the local search after a global scan (src/Optimiser.cpp:953-1008, thx_pf_scan_support_dev): gridR
    [nRin][4], gridT [nTin][2] f64; uR [nK][nImg][nRin], uT [nK][nImg][nTin] f32 scan weights; cls [nImg] int32 or None ->
    dict(r, t, wR, wT, k, s, topR, topT) = the filter state thx_pf_perturb_dev continues from"""
    _chk(gridR, _F64, "gridR"); _chk(gridT, _F64, "gridT"); _chk(uR, _F32, "uR"); _chk(uT, _F32, "uT")
    nImg, nRin, nTin, dev = uR.shape[1], uR.shape[2], uT.shape[2], uR.device
    if cls is not None:
        _chk(cls, _I32, "cls")
    e = lambda *sh: torch.empty(sh, dtype=_F64, device=dev)
    st = dict(r=e(nImg, mLR, 4), t=e(nImg, mLT, 2), wR=e(nImg, mLR), wT=e(nImg, mLT), k=e(nImg, 3), s=e(nImg, 2), topR=e(nImg, 4),
              topT=e(nImg, 2))
    capi.call("thx_pf_scan_support_dev", ptr(st["r"]), ptr(st["t"]), ptr(st["wR"]), ptr(st["wT"]), ptr(st["k"]), ptr(st["s"]),
              ptr(st["topR"]), ptr(st["topT"]), ptr(gridR), ptr(gridT), ptr(uR), ptr(uT), ptr(cls), nImg, nRin, nTin, int(mLR), int(mLT),
              float(peakFactorR), float(minK), float(minS), int(seed), int(call), stream_ptr())
    return st


def pf_acg_stats(quat):
    """inferACG / mean / k1..k3 / balance weights of [n][m][4] f64 clouds -> (A [n][16], mean [n][4], k [n][3], w [n][m])"""
    _chk(quat, _F64, "quat")
    n, m, dev = quat.shape[0], quat.shape[1], quat.device
    A = torch.empty((n, 16), dtype=_F64, device=dev)
    mean = torch.empty((n, 4), dtype=_F64, device=dev)
    k = torch.empty((n, 3), dtype=_F64, device=dev)
    w = torch.empty((n, m), dtype=_F64, device=dev)
    capi.call("thx_pf_acg_stats_dev", ptr(A), ptr(mean), ptr(k), ptr(w), None, ptr(quat), n, m, stream_ptr())
    return A, mean, k, w


class RecoPlan:
    """thx_reco handle: Reconstructor::allocSpace state (FFT plans, W, C, kernel table)."""

    def __init__(self, size, N, pf=2, a=1.9, alpha=15.0):
        self.size, self.N, self.pf = size, N, pf
        h = C.c_void_p()
        capi.call("thx_reco_create", C.byref(h), size, N, pf, a, alpha)
        self._h = h

    def close(self):
        if self._h is not None:
            capi.call("thx_reco_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_balance_rounds(self, max_iter=30, min_iter=10):
        """bounds of the gridding loop (MAX_N_ITER_BALANCE / MIN_N_ITER_BALANCE) of the following reconstructions"""
        capi.call("thx_reco_set_balance_rounds", self._h, int(max_iter), int(min_iter))

    def reconstruct(self, F, T, maxRadius, FSC=None, joinHalf=False, MAP=True, gridCorr=True):
        """Reconstructor::reconstruct (src/Reconstructor.cpp:1129-1831). T is modified in place (as the reference)."""
        _chk(F, _C64, "F"); _chk(T, _F32, "T")
        dst = torch.empty((self.N,) * 3, dtype=_F32, device=F.device)
        fsc_h = None
        nF = 0
        if MAP:
            fsc_h = np.ascontiguousarray(np.asarray(FSC, dtype=np.float32))
            nF = len(fsc_h)
        it, dc = C.c_int(0), C.c_float(0)
        capi.call("thx_reco_reconstruct_dev", self._h, ptr(F), ptr(T), maxRadius,
                  fsc_h.ctypes.data if fsc_h is not None else None, nF, 1 if joinHalf else 0, 1 if MAP else 0,
                  1 if gridCorr else 0, ptr(dst), C.byref(it), C.byref(dc), stream_ptr())
        self.last_iters, self.last_diffC = it.value, dc.value
        return dst

    def set_projectee(self, refRL):
        """Projector::setProjectee (src/Projector.cpp:123-148) from the real-space map -> padded FT"""
        _chk(refRL, _F32, "refRL")
        P = self.N * self.pf
        vol = torch.empty((P, P, P // 2 + 1), dtype=_C64, device=refRL.device)
        capi.call("thx_reco_set_projectee_dev", self._h, ptr(refRL), ptr(vol), stream_ptr())
        return vol
