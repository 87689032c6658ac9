"""thx_classify_iterate (thunder_amd/csrc/thx_classify.hip: one K-class classification iteration in native code) against the
oracle, stage boundary by stage boundary, on seeded inputs built with numpy + the oracle only:

  scan weights of every class (carried baseline)      oracle.expect_global over the K classes            relative weight bar
  class of every image                                oracle.pf_class_select on the device's uC, replayed Philox draws   equal
  support points after the scan                       oracle.pf_scan_support on the device's uR / uT of that class       tie rule of test_pf_gpu.py
  [local phases: the launches of thx_refine_iterate's chain (tests/test_iteration_gpu.py) with volIdx; the order of the calls is
   pinned bit for bit against the Python sequencing in tests/test_next_gpu.py]
  F / T of every class as the insertion left them     oracle.insertP of the replayed draws from the device's final support points  1e-5 max
  MAP-off / MAP-on maps of every class                oracle.reconstruct from the device's raw F / T (one T through both)          the chain test's bars

The Philox numbering follows the driver's launches: call 1 = class selection, 2 = support points, two calls per batch and
phase, one per batch for the insertion draws (image index = global index)."""
import numpy as np
import pytest

import _philox as PH

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def _case(O, N, K, nImg, nR, nT, seed):
    import _classify_util as U
    return U.make_case(O, N, K, nImg, nR, nT, seed)


@pytest.mark.parametrize("N,K,nImg,nR,nT,rScan,mLR,mLT,nPhase,mReco", [(32, 2, 128, 150, 6, 9, 40, 5, 2, 20), (32, 3, 150, 300, 4, 8, 24, 4, 1, 16)])
def test_native_classification_stages_against_oracle(oracle, dev, N, K, nImg, nR, nT, rScan, mLR, mLT, nPhase, mReco):
    from thunder_amd.native import NativeClassify
    O = oracle
    pf, P, rU, rL, seed = 2, 2 * N, N // 2 - 2, 2, 777001
    cs = _case(O, N, K, nImg, nR, nT, 50 + K)
    import _classify_util as U
    st_o = U.stages(O, N, K, cs["vols"], cs["quat"], cs["shifts"], rScan, rL)
    plM, plS = cs["plM"], st_o.plS
    s2m = U.sub_rows(plM, plS)
    w = np.full(nImg, np.float32(1.0) / np.float32(mReco), np.float32)
    d = dict(datM=T(cs["datM"], dev), ctfM=T(cs["ctfM"], dev), sigM=T(cs["sigM"], dev), w=T(w, dev))
    nat = NativeClassify(N, K, nImg, nR, nT, rScan, rL=rL, pf=pf, mLR=mLR, mLT=mLT, nPhase=nPhase, mReco=mReco, batch=nImg, seed=seed)
    nat.set_grid(T(cs["quat"], dev), T(cs["shifts"], dev))
    nat.set_particles(d["datM"], d["ctfM"], d["sigM"], d["w"])
    nat.set_references(T(cs["refs"], dev))
    cap = nat.capture()
    nat.iterate()
    torch.cuda.synchronize()
    v, c = nat.view(), nat.cfg
    assert (v.nPxlS, v.nPxlM) == (plS["nPxl"], plM["nPxl"])
    # ---- the projectors of the K references (Projector::setProjectee) ----
    volN = P * P * (P // 2 + 1)
    vd = nat.fetch(v.vols, np.complex64, (K, P, P, P // 2 + 1))
    for k in range(K):
        assert _rel(vd[k], cs["vols"][k]) <= 2e-6
    # ---- scan: class after class, the baseline and the weights carried along (src/Optimiser.cpp:756-894) ----
    wC, wR, wT, base = st_o.scan(cs["datM"][:, s2m], cs["ctfM"][:, s2m], cs["sigM"][:, s2m])
    uC, uR, uT = nat.fetch(v.uC, np.float32, (nImg, K)), nat.fetch(v.uR, np.float32, (K, nImg, nR)), nat.fetch(v.uT, np.float32, (K, nImg, nT))
    tol = max(6e-5 * float(np.abs(base).max()), 3e-4)      # the bar of test_expect_global
    np.testing.assert_allclose(uC, wC, rtol=tol)
    np.testing.assert_allclose(uR, wR, rtol=tol, atol=1e-30)
    np.testing.assert_allclose(uT, wT, rtol=tol, atol=1e-30)
    # ---- class of every image (:925-952), from the device's class weights with replayed draws: call 1 ----
    cls = nat.fetch(v.cls, np.int32, (nImg,))
    want = st_o.classes(uC, seed, c.peakFactorC)
    assert np.array_equal(cls, want)
    assert (cls == cs["cls_true"]).mean() >= 0.9
    # ---- support points from the scan posterior of that class (:953-1079): call 2 ----
    r0, t0 = cap["r0"].cpu().numpy(), cap["t0"].cpu().numpy()
    near = 0
    for l in range(nImg):
        ws, rankR = st_o.support(uR, uT, cls, l, seed, c.peakFactorR, mLR, mLT, c.scanMinK, c.scanMinS)
        dR = np.abs(r0[l][:, None, :] - cs["quat"][None, ws["srcR"], :]).max(axis=2)
        same = dR.diagonal() <= 1e-13
        for j in np.nonzero(~same)[0]:     # a draw on its threshold: the neighbour in the shuffled order (test_scan_support_points)
            srcD = int(np.argmin(np.abs(cs["quat"] - r0[l][j]).max(axis=1)))
            assert abs(int(rankR[srcD]) - int(rankR[ws["srcR"][j]])) <= 1, (l, j)
            near += 1
        assert np.array_equal(t0[l], ws["t"])
    assert near <= 2
    # ---- insertion (:7038-7241): the draws of every image from its FINAL support points into the F / T of its class ----
    q, t = nat.fetch(v.r, np.float64, (nImg, mLR, 4)), nat.fetch(v.t, np.float64, (nImg, mLT, 2))
    call = 2 + 2 * nPhase + 1
    F = np.zeros((K, P, P, P // 2 + 1), np.complex64)
    Tt = np.zeros((K, P, P, P // 2 + 1), np.float32)
    for l in range(nImg):
        u = PH.draw_u4(seed, l, call, 7, np.arange(mReco))
        iR = np.minimum((u[0] * mLR).astype(np.int64), mLR - 1)
        iT = np.minimum((u[1] * mLT).astype(np.int64), mLT - 1)
        for m in range(mReco):
            tt = t[l, iT[m]]
            src = O.translate(np.float32(-tt[0]), np.float32(-tt[1]), N, plM["iCol"], plM["iRow"], src=cs["datM"][l])
            O.insertP(F[cls[l]], Tt[cls[l]], P, src, cs["ctfM"][l], O.rotate3D(q[l, iR[m]]), w[l], plM["iColPad"], plM["iRowPad"])
    Fd, Td = cap["Fraw"].cpu().numpy(), cap["Traw"].cpu().numpy()
    for k in range(K):
        assert (cls == k).any()
        eF, eT = _rel(Fd[k], F[k]), _rel(Td[k], Tt[k])
        print("class %d: inserted F %.2e T %.2e of max" % (k, eF, eT))
        assert eF <= 1e-5 and eT <= 1e-5
    # ---- prepareTF's normalisation, reconstruct with MAP off, then MAP on with the all-ones FSC, ONE T through both ----
    st = nat.stats()
    m0, m1 = nat.fetch(v.maps, np.float32, (K, N, N, N)), nat.fetch(v.mapsMAP, np.float32, (K, N, N, N))
    ones = np.ones(rU, np.float32)
    for k in range(K):
        Fk, Tk = Fd[k].copy(), Td[k].copy()
        O.normalise_TF(Fk, Tk, P)
        dev_rounds = [int(st.lastRounds[2 * k]), int(st.lastRounds[2 * k + 1])]
        assert all(10 < r_ <= 30 for r_ in dev_rounds)
        a, ita, _, _ = O.reconstruct(Fk, Tk, P, N, pf, rU, MAP=False, joinHalf=False, gridCorr=True, return_iters=True, T_inplace=True)
        if ita != dev_rounds[0]:    # the stop rule is noise-sensitive (tests/test_iteration_gpu.py): compare after the SAME round
            Fk, Tk = Fd[k].copy(), Td[k].copy()
            O.normalise_TF(Fk, Tk, P)
            a, _, _, _ = O.reconstruct(Fk, Tk, P, N, pf, rU, MAP=False, joinHalf=False, gridCorr=True, return_iters=True, T_inplace=True,
                                       force_rounds=dev_rounds[0])
        Tsave = Tk.copy()
        b, itb, _, _ = O.reconstruct(Fk, Tk, P, N, pf, rU, FSC=ones, MAP=True, joinHalf=False, gridCorr=True, return_iters=True, T_inplace=True)
        if itb != dev_rounds[1]:
            Tk = Tsave
            b, _, _, _ = O.reconstruct(Fk, Tk, P, N, pf, rU, FSC=ones, MAP=True, joinHalf=False, gridCorr=True, return_iters=True,
                                       T_inplace=True, force_rounds=dev_rounds[1])
        e0, e1 = _rel(m0[k], a), _rel(m1[k], b)
        f0 = O.fsc(np.fft.rfftn(m0[k]).astype(np.complex64), np.fft.rfftn(a).astype(np.complex64), N, rU)
        f1 = O.fsc(np.fft.rfftn(m1[k]).astype(np.complex64), np.fft.rfftn(b).astype(np.complex64), N, rU)
        print("class %d: rounds device %s oracle %s; maps %.2e / %.2e of max, min FSC %.5f / %.5f" % (k, dev_rounds, [ita, itb], e0, e1, f0.min(), f1.min()))
        # the bars of the refinement chain (tests/test_iteration_gpu.py): with some tens of images per class the gridding loop is
        # far from converged when its stop rule fires and amplifies rounding noise of the FFTs on rim voxels where T is 1e-6 of
        # its largest value; identical-input reconstruction on well-covered volumes is held to 1e-4 in test_parity_gpu.py
        # (measured here with equal round counts: 5e-4 ... 1e-2 of max, FSC >= 0.9989 -- the MAP pass with the all-ones FSC adds no
        # regularisation and stops after ~12 rounds, the least converged of the four)
        for e, f, same in ((e0, f0, ita == dev_rounds[0]), (e1, f1, itb == dev_rounds[1])):
            assert e <= (3e-2 if same else 1e-1) and f.min() >= (0.997 if same else 0.95), (k, e, f.min(), same)   # (3 x the measured worst)
        # the class map resembles its own reference, not the next class's
        own = O.fsc(np.fft.rfftn(m0[k]).astype(np.complex64), np.fft.rfftn(cs["refs"][k]).astype(np.complex64), N, 6)
        other = O.fsc(np.fft.rfftn(m0[k]).astype(np.complex64), np.fft.rfftn(cs["refs"][(k + 1) % K]).astype(np.complex64), N, 6)
        assert own[1:6].mean() > other[1:6].mean() + 0.05, (k, own, other)
    assert st.balancingRounds == sum(int(x) for x in st.lastRounds)
    assert [int(x) for x in st.classCount[:K]] == [int((cls == k).sum()) for k in range(K)]
    nat.close()


def test_native_classification_refresh_and_errors(dev):
    """cfg.refresh: the MAP-on maps become the next iteration's references (Model::refreshProj) -- the second iteration's scan sees
    them; argument errors come back as status + message, not as crashes"""
    from thunder_amd import capi, synth
    from thunder_amd.native import NativeClassify
    N, K, nImg, nR, nT = 32, 2, 16, 64, 4
    rng = np.random.default_rng(3)
    with pytest.raises(capi.ThxError):
        NativeClassify(N, 17, nImg, nR, nT, 8)
    with pytest.raises(capi.ThxError):
        NativeClassify(N, K, nImg, nR, nT, N // 2)          # rScan beyond N / 2 - 2
    nat = NativeClassify(N, K, nImg, nR, nT, 8, mLR=16, mLT=4, nPhase=1, mReco=4, refresh=True)
    with pytest.raises(capi.ThxError):
        nat.iterate()                                       # nothing set yet
    nM = nat.stats().nPxlM
    datM = torch.view_as_complex(torch.randn((nImg, nM, 2), device=dev)).contiguous()
    ctfM = torch.rand((nImg, nM), device=dev) * 2 - 1
    sigM = torch.full((nImg, nM), -0.5, device=dev)
    w = torch.full((nImg,), 0.25, device=dev)
    nat.set_grid(T(synth.random_quats(nR, rng), dev), T(rng.normal(0, 1, (nT, 2)), dev))
    nat.set_particles(datM, ctfM, sigM, w)
    refs = T(np.stack([synth.blob_map(N, seed=5 + k, nblob=6) for k in range(K)]), dev)
    nat.set_references(refs)
    v = nat.view()
    P = 2 * N
    before = nat.fetch(v.vols, np.complex64, (K, P, P, P // 2 + 1))
    nat.iterate()
    after = nat.fetch(v.vols, np.complex64, (K, P, P, P // 2 + 1))
    cnt = [int(x) for x in nat.stats().classCount[:K]]
    assert sum(cnt) == nImg
    for k in range(K):
        assert np.all(np.isfinite(after[k]))
        assert (not np.array_equal(after[k], before[k])) == (cnt[k] > 0)      # an empty class keeps its reference
    nat.iterate()
    assert nat.stats().iterations == 2
    nat.close()


def test_native_classification_forced_one_rank_reduce(dev, knob_env):
    """the per-class integer half-set reduce of the classification driver (thx_reco_allreduce_acc_class: sphere rows of ONE class of
    the [nK][vol][2] F | [nK][vol] T session buffer packed, ncclAllReduce(int64), unpacked) on a forced one-rank communicator:
    the iteration with the communicator must equal the iteration without it bit for bit -- which also shows that class k's
    pack / unpack addresses class k's rows and nothing else; and the reduce called directly leaves the other classes untouched"""
    from thunder_amd import capi, synth
    from thunder_amd.capi import ptr, stream_ptr
    from thunder_amd.native import Comm, NativeClassify
    knob_env("THX_COMM_FORCE", "1")
    comm = Comm(0, 1, lambda uid: uid)
    N, K, nImg, nR, nT = 32, 3, 48, 96, 4
    P = 2 * N
    rng = np.random.default_rng(8)
    quat, shifts = T(synth.random_quats(nR, rng), dev), T(rng.normal(0, 1, (nT, 2)), dev)
    refs = T(np.stack([synth.blob_map(N, seed=15 + k, nblob=6) for k in range(K)]), dev)
    out = []
    for hemi in (None, comm):
        nat = NativeClassify(N, K, nImg, nR, nT, 8, mLR=16, mLT=4, nPhase=1, mReco=5, hemi=hemi, seed=99)
        nM = nat.stats().nPxlM
        g = torch.Generator(device=dev).manual_seed(4)
        datM = torch.view_as_complex(torch.randn((nImg, nM, 2), device=dev, generator=g)).contiguous()
        ctfM = torch.rand((nImg, nM), device=dev, generator=g) * 2 - 1
        sigM = torch.full((nImg, nM), -0.5, device=dev)
        w = torch.full((nImg,), 0.2, device=dev)
        nat.set_grid(quat, shifts); nat.set_particles(datM, ctfM, sigM, w); nat.set_references(refs)
        cap = nat.capture()
        nat.iterate()
        torch.cuda.synchronize()
        v = nat.view()
        out.append((cap["Fraw"].clone(), cap["Traw"].clone(), nat.fetch(v.maps, np.float32, (K, N, N, N)), nat.fetch(v.cls, np.int32, (nImg,))))
        nat.close()
    assert out[0][0].abs().max().item() > 0
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])
    # the reduce of class 1 on its own: a buffer of distinct integers comes back unchanged, classes 0 and 2 included
    volN = P * P * (P // 2 + 1)
    nbytes = capi.load().thx_insert_acc_bytes(P, K)
    assert nbytes == K * volN * 3 * 8
    acc = torch.arange(K * volN * 3, dtype=torch.int64, device=dev) * 7 - 12345
    acc0 = acc.clone()
    ws = torch.empty(capi.load().thx_reco_allreduce_acc_workspace(P, N // 2 - 2, 2), dtype=torch.uint8, device=dev)
    for k in range(K):
        capi.call("thx_reco_allreduce_acc_class", comm.handle, ptr(acc), K, k, None, None, P, N // 2 - 2, 2, ptr(ws), stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(acc, acc0)
    comm.close()


def test_native_classification_against_committed_fixture(dev):
    """fixture-only variant: tests/golden/classify_n16.npz holds the inputs and the oracle's scan weights, classes and support
    points of a K = 2 iteration at N = 16 (generator: tests/golden/make_golden_classify.py); thx_classify_iterate on the same inputs
    must reproduce them -- weights at the scan's bar, classes and resampled grid points exactly (no weight of this case sits
    within rounding of a threshold)"""
    import os
    import sys
    from thunder_amd.native import NativeClassify
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden_classify as G
    g = np.load(os.path.join(here, "golden", "classify_n16.npz"))
    c = G.CFG
    nImg, K, nR, nT, mLR, mLT = c["nImg"], c["K"], c["nR"], c["nT"], c["mLR"], c["mLT"]
    nat = NativeClassify(c["N"], K, nImg, nR, nT, c["rScan"], rL=c["rL"], mLR=mLR, mLT=mLT, nPhase=1, mReco=4, batch=nImg, seed=c["seed"],
                         peakFactorR=c["peakFactorR"], peakFactorC=c["peakFactorC"], scan_min=(c["minK"], c["minS"]))
    d = [T(g[k], dev) for k in ("datM", "ctfM", "sigM")]
    w = torch.full((nImg,), 0.25, device=dev)
    nat.set_grid(T(g["quat"], dev), T(g["shifts"], dev))
    nat.set_particles(d[0], d[1], d[2], w)
    nat.set_references(T(g["refs"], dev))
    cap = nat.capture()
    nat.iterate()
    torch.cuda.synchronize()
    v = nat.view()
    tol = max(6e-5 * float(np.abs(g["base"]).max()), 3e-4)
    np.testing.assert_allclose(nat.fetch(v.uC, np.float32, (nImg, K)), g["wC"], rtol=tol)
    np.testing.assert_allclose(nat.fetch(v.uR, np.float32, (K, nImg, nR)), g["wR"], rtol=tol, atol=1e-30)
    np.testing.assert_allclose(nat.fetch(v.uT, np.float32, (K, nImg, nT)), g["wT"], rtol=tol, atol=1e-30)
    assert np.array_equal(nat.fetch(v.cls, np.int32, (nImg,)), g["cls"])
    r0, t0 = cap["r0"].cpu().numpy(), cap["t0"].cpu().numpy()
    assert np.abs(r0 - g["quat"][g["srcR"]]).max() <= 1e-13 and np.array_equal(t0, g["t0"])
    nat.close()
