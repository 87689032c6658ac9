"""The per-image / per-stage Interface.h entry points (gpu/interface/Interface.h:18-164, 337-502) driven the way
src/Optimiser.cpp:2255-2600 and src/Reconstructor.cpp:1835-2330 drive them, against the batched device path and the
oracle.  The test plays the reference's host side: it owns the host arrays and performs the host FFTs."""
import ctypes as C

import numpy as np
import pytest
import scipy.fft as sfft

from _util import make_case, make_images

pytestmark = pytest.mark.gpu


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _vp():
    return C.c_void_p()


def _host_array(ptr, n, dtype):
    ct = {np.float32: C.c_float, np.float64: C.c_double}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))


@pytest.mark.parametrize("cSearch", [0, 2])
def test_staged_local_search(oracle, dev, cSearch):
    """ExpectPreidx/Prefre/LocalIn/LocalV3D/LocalP/LocalHostA/LocalRTD/LocalPreI3D/LocalM/LocalHostF/LocalFin/FreeIdx
    image by image == thx_expect_local_dev on the batch (same kernel: bit-equal) and the oracle (tolerances of
    test_parity_gpu.test_expect_local)"""
    from thunder_amd import capi, ops, synth
    O = oracle
    rng = np.random.default_rng(77 + cSearch)
    N, nR, nT, mD, nImg, cpyNumL = 32, 20, 9, 3, 3, 2
    P = 2 * N
    ref, vol, pl = make_case(O, N, rL=1)
    nPxl = pl["nPxl"]
    im = make_images(O, vol, pl, N, nImg, rng, snr_sigma=2.0)
    im["attr"][:, 6] = rng.uniform(0, 0.3, nImg)
    quat = np.ascontiguousarray(synth.perturb_quats(im["quat"], nR, 0.04, rng))      # [nImg][nR][4]
    rot = np.stack([[O.rotate3D(q) for q in qs] for qs in quat])
    tran = np.ascontiguousarray(im["shift"][:, None, :] + rng.normal(0, 0.5, size=(nImg, nT, 2)))
    pR = rng.uniform(0.5, 1.5, size=(nImg, nR))
    pT = rng.uniform(0.5, 1.5, size=(nImg, nT))
    pC = rng.uniform(0.5, 1.5, size=nImg)
    nD = mD if cSearch == 2 else 1
    pD = rng.uniform(0.5, 1.5, size=(nImg, nD))
    dpara = 1.0 + rng.normal(0, 0.02, size=(nImg, mD))
    fq, de, k1, k2 = O.expect_precal(im["attr"], N, 1.32, pl["iCol"], pl["iRow"])

    # ---- the reference's call sequence ----
    deviCol, deviRow, devfreQ = _vp(), _vp(), _vp()
    capi.call("thx_ExpectPreidx_host", 0, C.byref(deviCol), C.byref(deviRow), pl["iCol"].ctypes.data, pl["iRow"].ctypes.data,
              nPxl)
    if cSearch == 2:
        capi.call("thx_ExpectPrefre_host", 0, C.byref(devfreQ), fq.ctypes.data, nPxl)
    devdatP, devctfP, devdefO, devsigP = _vp(), _vp(), _vp(), _vp()
    capi.call("thx_ExpectLocalIn_host", 0, C.byref(devdatP), C.byref(devctfP), C.byref(devdefO), C.byref(devsigP), nPxl,
              cpyNumL, cSearch)
    mgr, mcp = _vp(), _vp()
    capi.call("thx_texture_create", C.byref(mgr), 1, P, 0)
    capi.call("thx_ExpectLocalV3D_host", 0, mgr, vol.ctypes.data, P)
    capi.call("thx_calpoint_create", C.byref(mcp), 1, cSearch, 0, nR, nT, mD, nPxl)
    hp = [_vp() for _ in range(10)]   # wC wR wT wD oldR oldT oldD trans rot dpara
    capi.call("thx_ExpectLocalHostA_host", 0, *[C.byref(p) for p in hp], nR, nT, mD, cSearch)
    wC, wR, wT = _host_array(hp[0], 1, np.float32), _host_array(hp[1], nR, np.float32), _host_array(hp[2], nT, np.float32)
    wD = _host_array(hp[3], mD, np.float32)
    oldR, oldT = _host_array(hp[4], nR, np.float64), _host_array(hp[5], nT, np.float64)
    oldD = _host_array(hp[6], nD, np.float64)
    h_tr, h_rot = _host_array(hp[7], 2 * nT, np.float64), _host_array(hp[8], 4 * nR, np.float64)
    h_dp = _host_array(hp[9], mD, np.float64) if cSearch == 2 else None
    got = []
    for l in range(nImg):
        slot = l % cpyNumL
        capi.call("thx_ExpectLocalP_host", 0, devdatP, devctfP, devdefO, devsigP, im["dat"].ctypes.data,
                  im["ctf"].ctypes.data, de.ctypes.data, im["sigRcp"].ctypes.data, slot, l, nPxl, cSearch)
        oldR[:] = pR[l]; oldT[:] = pT[l]; oldD[:] = pD[l]
        h_tr[:] = tran[l].reshape(-1); h_rot[:] = quat[l].reshape(-1)
        if cSearch == 2:
            h_dp[:] = dpara[l]
        capi.call("thx_ExpectLocalRTD_host", 0, mcp, hp[4], hp[5], hp[6], hp[7], hp[8], hp[9])
        a = im["attr"][l]
        capi.call("thx_ExpectLocalPreI3D_host", 0, slot, mgr, mcp, devdefO, devfreQ, deviCol, deviRow, float(a[6]),
                  float(a[5]), float(k1[l]) if cSearch == 2 else 0.0, float(k2[l]) if cSearch == 2 else 0.0, 2, N, P, nPxl,
                  1)
        capi.call("thx_ExpectLocalM_host", 0, slot, mcp, devdatP, devctfP, devsigP, hp[0], hp[1], hp[2], hp[3], float(pC[l]),
                  nPxl)
        got.append(dict(wC=wC.copy(), wR=wR.copy(), wT=wT.copy(), wD=wD[:nD].copy()))
    capi.call("thx_ExpectLocalHostF_host", 0, *[C.byref(p) for p in hp], cSearch)
    capi.call("thx_ExpectLocalFin_host", 0, C.byref(devdatP), C.byref(devctfP), C.byref(devdefO), C.byref(devfreQ),
              C.byref(devsigP), cSearch)
    capi.call("thx_ExpectFreeIdx_host", 0, C.byref(deviCol), C.byref(deviRow))
    capi.call("thx_calpoint_destroy", mcp)
    capi.call("thx_texture_destroy", mgr)
    assert devdatP.value is None and deviCol.value is None

    # ---- the batched device path on the same inputs ----
    d_attr = ops.ctf_attr_tensor(im["attr"], dev)
    if cSearch == 2:
        gfq, gde, gk1, gk2 = ops.expect_precal(d_attr, 1.32, T(pl["iCol"], dev), T(pl["iRow"], dev), N)
        # the staged path was fed the oracle's def rows: use the same here
        ctfD = ops.ctf_dsearch(T(fq, dev), T(de, dev), T(k1, dev), T(k2, dev), d_attr, T(dpara, dev))
    else:
        ctfD = T(im["ctf"], dev)
    res = ops.expect_local(T(vol, dev), P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(im["dat"], dev), ctfD,
                           T(im["sigRcp"], dev), T(rot, dev), T(tran, dev), nD=nD, pC=T(pC, dev), pR=T(pR, dev),
                           pT=T(pT, dev), pD=T(pD, dev))
    for l in range(nImg):
        for name in ("wC", "wR", "wT", "wD"):
            b = getattr(res, name)[l].cpu().numpy().reshape(-1)
            # rotation matrices come from the device rotate3D in the staged path and from the oracle's here: <= 4e-16
            # apart, which moves a log-likelihood by far less than its float rounding
            np.testing.assert_allclose(got[l][name].reshape(-1), b, rtol=1e-4, atol=1e-30, err_msg="%s[%d]" % (name, l))
    # ---- and the oracle ----
    for l in range(nImg):
        if cSearch == 2:
            ctf_l = np.stack([O.ctf_dsearch(fq, de[l], k1[l], k2[l], im["attr"][l, 6], im["attr"][l, 5], dpara[l])])[0]
            want = O.expect_local(vol, P, 2, N, pl["iCol"], pl["iRow"], im["dat"][l], ctf_l, im["sigRcp"][l], rot[l],
                                  tran[l], nD=nD, pC=pC[l], pR=pR[l], pT=pT[l], pD=pD[l], cSearch=True)
        else:
            want = O.expect_local(vol, P, 2, N, pl["iCol"], pl["iRow"], im["dat"][l], im["ctf"][l], im["sigRcp"][l],
                                  rot[l], tran[l], nD=1, pC=pC[l], pR=pR[l], pT=pT[l], pD=pD[l])
        for name in ("wC", "wR", "wT", "wD"):
            np.testing.assert_allclose(got[l][name].reshape(-1), want[name].reshape(-1), rtol=2e-3, atol=1e-30,
                                       err_msg="oracle %s[%d]" % (name, l))


def _inserted_volume(O, N, rng, n=400):
    from thunder_amd import synth
    P = 2 * N
    ref, vol, pl = make_case(O, N)
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    Tt = np.zeros((P, P, P // 2 + 1), np.float32)
    one = np.ones(pl["nPxl"], np.float32)
    for q in synth.random_quats(n, rng):
        R = O.rotate3D(q)
        O.insertP(F, Tt, P, O.project(vol, P, 2, R, pl["iCol"], pl["iRow"]), one, R, 1.0, pl["iColPad"], pl["iRowPad"])
    O.normalise_TF(F, Tt, P)
    return F, Tt


def _tik_table(O, N, pf):
    """mkbRL of src/Reconstructor.cpp:2284-2305 under RECONSTRUCTOR_TRILINEAR_KERNEL: TIK_RL(|x| / (pf N))"""
    h = N // 2 + 1
    k, j, i = np.meshgrid(np.arange(h), np.arange(h), np.arange(h), indexing="ij")
    r = np.sqrt((i * i + j * j + k * k).astype(np.float64)) / (pf * N)
    return O.tik_rl(r.astype(np.float32).reshape(-1)).reshape(h, h, h)


@pytest.mark.parametrize("N", [32, 24])
def test_staged_reconstructG(oracle, dev, N):
    """ExposePT -> AllocDevicePoint/HostDeviceInit/{ExposeC, host FFT, ExposeForConvC, host FFT, ExposeWC}*/FreeDevHostPoint
    -> ExposePFW -> host FFT -> ExposeCorrF, chained exactly as Reconstructor::reconstructG does, == the oracle's
    reconstruct and the fused thx_ReconstructG_host; ExposeWT (device-resident loop) gives the same weights."""
    from thunder_amd import capi
    O = oracle
    rng = np.random.default_rng(5)
    P, pf, maxRadius = 2 * N, 2, N // 2 - 2
    F, Tt = _inserted_volume(O, N, rng)
    fscv = np.clip(np.linspace(1.0, 0.05, N // 2), 0, 1).astype(np.float32)
    want, it_w, diffs, W_want = O.reconstruct(F, Tt, P, N, pf, maxRadius, FSC=fscv, joinHalf=True, MAP=True,
                                              gridCorr=True, return_iters=True)
    tab = O.kernelRL_table(1.9, 15.0)
    nf = float(O.mkb_rl(0.0, 1.9, 15.0))
    nHalf = P * P * (P // 2 + 1)

    volumeT = Tt.copy()
    capi.call("thx_ExposePT_host", 0, volumeT.ctypes.data, maxRadius, pf, P, fscv.ctypes.data, len(fscv), 1, 5)
    volumeW = np.zeros(nHalf, np.float32)
    ptr = [_vp() for _ in range(7)]     # dev_C dev_W dev_T dev_tab devDiff devMax devCount
    streamNum, tabSize = 3, 100000
    stream = (C.c_void_p * streamNum)()
    capi.call("thx_AllocDevicePoint_host", 0, *[C.byref(p) for p in ptr], stream, streamNum, tabSize, P)
    dev_C, dev_W, dev_T, dev_tab, _, devMax, _ = ptr
    capi.call("thx_HostDeviceInit_host", 0, volumeT.ctypes.data, tab.ctypes.data, dev_W, dev_T, dev_tab, stream, streamNum,
              tabSize, maxRadius, pf, P)
    C3D = np.zeros((P, P, P // 2 + 1), np.complex64)
    diffC = C.c_float(np.finfo(np.float32).max)
    diffCPrev, nNoDec, m = diffC.value, 0, 0
    for m in range(30):
        capi.call("thx_ExposeC_host", 0, C3D.ctypes.data, dev_C, dev_T, dev_W, stream, streamNum, P)
        crl = np.ascontiguousarray(sfft.irfftn(C3D, s=(P, P, P)).astype(np.float32))       # _fft.bwExecutePlan
        capi.call("thx_ExposeForConvC_host", 0, crl.ctypes.data, dev_C, dev_tab, stream, np.float32(1.0) / np.float32(1e5),
                  nf, streamNum, tabSize, pf, N, P)
        C3D = np.ascontiguousarray(sfft.rfftn(crl).astype(np.complex64))                   # _fft.fwExecutePlan
        diffCPrev = diffC.value
        capi.call("thx_ExposeWC_host", 0, C3D.ctypes.data, dev_C, None, dev_W, devMax, stream, C.byref(diffC), streamNum,
                  maxRadius, pf, P)
        nNoDec = nNoDec + 1 if diffC.value > diffCPrev * 0.95 else 0
        if diffC.value < 1e-2 or (m >= 10 and nNoDec == 2):
            break
    capi.call("thx_FreeDevHostPoint_host", 0, *[C.byref(p) for p in ptr], stream, volumeW.ctypes.data, streamNum, P)
    assert m + 1 == it_w and abs(diffC.value - diffs[-1]) <= 1e-3 * max(1.0, diffs[-1])
    assert np.abs(volumeW - W_want).max() <= 1e-4 * np.abs(W_want).max()

    padDst = np.zeros((P, P, P // 2 + 1), np.complex64)
    capi.call("thx_ExposePFW_host", 0, padDst.ctypes.data, F.ctypes.data, volumeW.ctypes.data, maxRadius, pf, P, P)
    prl = sfft.irfftn(padDst, s=(P, P, P)).astype(np.float32)                               # fft.bw(padDst)
    idx = np.r_[0:N // 2, P - N // 2:P]
    dst = np.ascontiguousarray(prl[np.ix_(idx, idx, idx)])                                  # dst.setRL(padDst.getRL(i,j,k))
    tik = _tik_table(O, N, pf)
    capi.call("thx_ExposeCorrF_host", 0, dst.ctypes.data, tik.ctypes.data, 0.0, N)
    assert np.abs(dst - want).max() <= 1e-4 * np.abs(want).max()

    # the fused entry on the same host inputs
    Tin = np.zeros((P, P, P // 2 + 1), np.complex64)
    Tin.real = Tt
    fused = np.zeros((N, N, N), np.float32)
    capi.call("thx_ReconstructG_host", 0, F.ctypes.data, Tin.ctypes.data, N, N, pf, maxRadius, 1.9, 15.0, fscv.ctypes.data,
              len(fscv), 1, 1, 1, fused.ctypes.data)
    assert np.abs(dst - fused).max() <= 1e-4 * np.abs(want).max()

    # ExposeWT: the same iteration, device-resident
    W2 = np.zeros(nHalf, np.float32)
    capi.call("thx_ExposeWT_host", 0, volumeT.ctypes.data, W2.ctypes.data, tab.ctypes.data, tabSize, nf, maxRadius, pf, P,
              30, 10, N)
    assert np.abs(W2 - W_want).max() <= 1e-4 * np.abs(W_want).max()

    # ExposePF = ExposePFW + inverse FFT; ExposeCorrF (dstN, dst) = correction + forward FFT
    padR = np.zeros((P, P, P), np.float32)
    capi.call("thx_ExposePF_host", 0, None, padR.ctypes.data, F.ctypes.data, volumeW.ctypes.data, maxRadius, pf, P, P)
    assert np.abs(padR - prl).max() <= 1e-5 * np.abs(prl).max()
    dstN = np.ascontiguousarray(prl[np.ix_(idx, idx, idx)])
    dstFT = np.zeros((N, N, N // 2 + 1), np.complex64)
    capi.call("thx_ExposeCorrF_fft_host", 0, dstN.ctypes.data, dstFT.ctypes.data, tik.ctypes.data, 0.0, N)
    wantFT = sfft.rfftn(dst)
    assert np.abs(dstFT - wantFT).max() <= 1e-5 * np.abs(wantFT).max()


def test_staged_no_gridcorr_weights(oracle, dev):
    """ExposeWT(gpuIdx, T3D, W3D, maxRadius, pf, dim): W = 1 / max(|T|, 1e-6) in the sphere, 0 outside"""
    from thunder_amd import capi
    O = oracle
    rng = np.random.default_rng(6)
    N = 32
    P, pf, maxRadius = 2 * N, 2, N // 2 - 2
    F, Tt = _inserted_volume(O, N, rng, n=100)
    want, _, _, W_want = O.reconstruct(F, Tt, P, N, pf, maxRadius, MAP=False, gridCorr=False, return_iters=True)
    W = np.zeros(P * P * (P // 2 + 1), np.float32)
    capi.call("thx_ExposeWT_plain_host", 0, Tt.ctypes.data, W.ctypes.data, maxRadius, pf, P)
    assert np.array_equal(W, W_want.reshape(-1))


class _HArgs(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("gpu", "N", "pf", "nPxl", "nImg", "mLR", "mLT", "phases", "threads", "lock")] + \
               [(k, C.c_void_p) for k in ("volume", "iCol", "iRow", "datP", "ctfP", "sigP", "quat", "tran", "attr", "wR", "wT")] + \
               [("seconds", C.c_double), ("callSeconds", C.c_double * 4)]


@pytest.mark.parametrize("N,nImg,nR", [(32, 24, 20), (64, 12, 125)])
def test_caller_loop_harness_matches_batched_and_oracle(oracle, dev, N, nImg, nR):
    """The reference's CALLER loop over the plug-in surface (Optimiser::expectationG, src/Optimiser.cpp:2790-3100) as
    integration/expectationG_harness.cpp restates it in C++ / OpenMP over the C ABI -- with the per-GPU lock the reference holds over
    ExpectLocalRTD .. ExpectLocalM (2 threads) and without it (6 threads on their own ManagedCalPoints / streams) -- against the batched
    kernel on the same inputs (same arithmetic, partial sums grouped differently: 1e-4) and the oracle (test_expect_local's bar).
    nR = 125 at N = 64: two rotation groups per workgroup, one workgroup per 256-pixel chunk (8 chunks), cell-packed volume.  The two
    harness runs are the same computation image by image: bit-identical to each other."""
    import torch
    from thunder_amd import build, ops, synth
    O = oracle
    rng = np.random.default_rng(900 + N)
    nT, phases = 9, 2
    P = 2 * N
    ref, vol, pl = make_case(O, N, rL=1)
    nPxl = pl["nPxl"]
    im = make_images(O, vol, pl, N, nImg, rng, snr_sigma=2.0)
    quat = np.ascontiguousarray(synth.perturb_quats(im["quat"], nR, 0.04, rng))
    rot = np.stack([[O.rotate3D(q) for q in qs] for qs in quat])
    tran = np.ascontiguousarray(im["shift"][:, None, :] + rng.normal(0, 0.5, size=(nImg, nT, 2)))
    H = C.CDLL(build.build_harness())
    H.thx_harness_expectation_local.restype = C.c_int
    iCol, iRow = np.ascontiguousarray(pl["iCol"]), np.ascontiguousarray(pl["iRow"])
    attr = np.ascontiguousarray(im["attr"], np.float32)
    got = {}
    for lock, threads in ((1, 2), (0, 6)):
        wR, wT = np.zeros((nImg, nR), np.float32), np.zeros((nImg, nT), np.float32)
        a = _HArgs(gpu=0, N=N, pf=2, nPxl=nPxl, nImg=nImg, mLR=nR, mLT=nT, phases=phases, threads=threads, lock=lock, volume=vol.ctypes.data,
                   iCol=iCol.ctypes.data, iRow=iRow.ctypes.data, datP=im["dat"].ctypes.data, ctfP=im["ctf"].ctypes.data,
                   sigP=im["sigRcp"].ctypes.data, quat=quat.ctypes.data, tran=tran.ctypes.data, attr=attr.ctypes.data, wR=wR.ctypes.data,
                   wT=wT.ctypes.data)
        assert H.thx_harness_expectation_local(C.byref(a)) == 0
        assert a.seconds > 0
        got[lock] = (wR, wT)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    res = ops.expect_local(T(vol, dev), P, 2, N, T(iCol, dev), T(iRow, dev), T(im["dat"], dev), T(im["ctf"], dev), T(im["sigRcp"], dev),
                           T(rot, dev), T(tran, dev), nD=1, pR=T(np.full((nImg, nR), 1.0 / nR), dev), pT=T(np.full((nImg, nT), 1.0 / nT), dev))
    np.testing.assert_allclose(got[1][0], res.wR.cpu().numpy(), rtol=1e-4, atol=1e-30)
    np.testing.assert_allclose(got[1][1], res.wT.cpu().numpy(), rtol=1e-4, atol=1e-30)
    for l in range(0, nImg, 5):
        want = O.expect_local(vol, P, 2, N, iCol, iRow, im["dat"][l], im["ctf"][l], im["sigRcp"][l], rot[l], tran[l], nD=1, pC=1.0,
                              pR=np.full(nR, 1.0 / nR), pT=np.full(nT, 1.0 / nT))
        np.testing.assert_allclose(got[1][0][l], want["wR"], rtol=2e-3, atol=1e-30)
        np.testing.assert_allclose(got[1][1][l], want["wT"], rtol=2e-3, atol=1e-30)
    torch.cuda.synchronize()
