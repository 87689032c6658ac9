"""Frequency cut-offs as per-iteration inputs and the resized reconstruction grid (round-5 review #1).

The reference's E-step runs on allocPreCalIdx(_r, _rL) (src/Optimiser.cpp:631,1693), its M-step on allocPreCalIdx(rU, 0)
(:6722-6741), and every iteration below Nyquist ends with Reconstructor::resizeSpace(min(_size, (_rU + ceil(a)) * 2)) + setFSC +
setMaxRadius(_rU) (src/Model.cpp:1100-1125, src/Reconstructor.cpp:184-198): F / T / W / C and the gridding loop then live on a
(pf size)^3 grid and reconstruct() places F W into an (N pf)^3 padDst at the end (:1677-1701).  Here:
  * thx_reco_create(size < N) + thx_reco_reconstruct_dev against oracle.reconstruct on the resized grid, at SURVEY 8c(9)'s bars;
  * the brick-sorted insertion into a resized grid (a dimension that is no multiple of the brick) against the oracle's insertP;
  * thx_refine_set_cutoff: Nyquist set explicitly == the default path bit for bit; a three-iteration chain at N = 64 with
    (r, rU) = (12, 14) -> (20, 22) -> (30, 30) against oracle.Iteration.set_cutoff stage by stage (tests/test_iteration_gpu.py's
    checker); a property run at 256^3 with r = rU = 48.
"""
import numpy as np
import pytest
import scipy.fft as sfft

import _iter_util as U
from _util import make_case

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _fsc_np(O, a, b, N, n):
    return O.fsc(sfft.rfftn(a).astype(np.complex64), sfft.rfftn(b).astype(np.complex64), N, n)


def _inserted(O, N, rU, P, n, seed, vol=None):
    """F / T on a P^3 half grid from n clean slices of a blob map on the rU list (oracle insertP), normalised"""
    from thunder_amd import synth
    rng = np.random.default_rng(seed)
    ref, vol_, pl = make_case(O, N, rU=rU)
    vol = vol_ if vol is None else vol
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    Tt = np.zeros((P, P, P // 2 + 1), np.float32)
    ctf1 = np.ones(pl["nPxl"], np.float32)
    for q in synth.random_quats(n, rng):
        R = O.rotate3D(q)
        O.insertP(F, Tt, P, O.project(vol, 2 * N, 2, R, pl["iCol"], pl["iRow"]), ctf1, R, 1.0, pl["iColPad"], pl["iRowPad"])
    O.normalise_TF(F, Tt, P)
    return ref, F, Tt


@pytest.mark.parametrize("N,size,MAP,joinHalf", [(64, 40, False, False), (64, 40, True, True), (64, 32, True, False), (256, 128, True, True)])
def test_reconstruct_resized(oracle, dev, N, size, MAP, joinHalf):
    """Reconstructor::reconstruct with _size < _N (after resizeSpace): PAD_SIZE = pf size for F / T / W / C and the gridding loop,
    convoluteC's kernel argument still QUAD_3 / (N pf)^2, F W padded into (N pf)^3 for the last c2r.  size 40 -> an 80^3 grid on
    rocFFT, size 32 / 128 -> 64^3 / 256^3 on the hand-written passes.  Bars: SURVEY 8c(9) -- 1e-4 of max, FSC >= 0.9999 on every shell,
    the same number of balancing rounds."""
    from thunder_amd import ops
    O = oracle
    rU = size // 2 - 2                                      # Model::resetReco: size = (rU + ceil(a)) * 2
    P = 2 * size
    # (enough views for the balancing loop to converge instead of jittering: with 300 slices at 256^3 the rim of the sphere holds ~2
    # samples per voxel, W runs away where T sits on its 1e-25 floor and the two sides leave the loop in different rounds --
    # tests/test_iteration_cpu.py::test_stop_rule_is_noise_sensitive)
    ref, F, Tt = _inserted(O, N, rU, P, 400 if N == 64 else 4000, seed=33 + size)
    fscv = np.clip(np.linspace(1.0, 0.05, rU), 0, 1).astype(np.float32)      # setFSC: _rU entries
    want, it_w, diffs, _ = O.reconstruct(F, Tt, P, N, 2, rU, FSC=fscv, joinHalf=joinHalf, MAP=MAP, gridCorr=True, return_iters=True)
    plan = ops.RecoPlan(size, N, 2)
    got = plan.reconstruct(T(F, dev), T(Tt, dev), rU, FSC=fscv, joinHalf=joinHalf, MAP=MAP, gridCorr=True).cpu().numpy()
    e = float(np.abs(got - want).max() / np.abs(want).max())
    f = _fsc_np(O, got, want, N, rU)
    print("N %d size %d (grid %d^3): rounds device %d oracle %d, checkC %.5g / %.5g, map %.2e of max, min FSC %.7f"
          % (N, size, P, plan.last_iters, it_w, plan.last_diffC, diffs[-1], e, f.min()))
    assert plan.last_iters == it_w
    assert abs(plan.last_diffC - diffs[-1]) <= 1e-3 * max(1.0, diffs[-1])
    assert e <= 1e-4 and f.min() >= 0.9999
    # and it is a reconstruction: the map agrees with the generating map inside the cut-off
    # (the inner shells: a blob map has no signal to correlate with further out -- at 256^3 the curve is interpolation error beyond ~30)
    fr = _fsc_np(O, got, ref, N, rU)
    assert fr[: min(12, max(3, rU - 3))].min() >= 0.95, fr
    # without the gridding loop (gridCorr off) the same placement is exercised on its own
    Tt2 = Tt.copy()
    want2 = O.reconstruct(F, Tt2, P, N, 2, rU, MAP=False, gridCorr=False)
    got2 = plan.reconstruct(T(F, dev), T(Tt, dev), rU, MAP=False, gridCorr=False).cpu().numpy()
    assert np.abs(got2 - want2).max() <= 1e-5 * np.abs(want2).max()
    plan.close()


@pytest.mark.parametrize("N,rU", [(64, 18), (32, 9)])
def test_insert_into_resized_grid(oracle, dev, N, rU):
    """Reconstructor::insertP into the (pf size)^3 volumes of a resized reconstructor: grid 80^3 (N = 64, rU = 18) and 44^3
    (N = 32, rU = 9) -- neither a multiple of the 16 x 8 x 8 bricks of the sorted insertion.  Against the oracle's insertP per
    draw: 1e-5 of max (fixed-point sums vs float adds in another order)."""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(5 + N)
    size = min(N, (rU + 2) * 2)
    P = 2 * size
    ref, vol, pl = make_case(O, N, rU=rU)
    nImg, mReco = 40, 6
    quat = synth.random_quats(nImg, rng)
    rots = np.stack([[O.rotate3D(q) for q in synth.perturb_quats(quat[l:l + 1], mReco, 0.05, rng)[0]] for l in range(nImg)])
    tran = rng.normal(0, 1.0, size=(nImg, mReco, 2))
    attr = synth.ctf_params(nImg, rng)
    dat = np.stack([O.project(vol, 2 * N, 2, rots[l, 0], pl["iCol"], pl["iRow"]) for l in range(nImg)]).astype(np.complex64)
    ctf = np.stack([O.ctf(1.32, *attr[l], N, pl["iCol"], pl["iRow"]) for l in range(nImg)])
    w = np.full(nImg, 1.0 / mReco, np.float32)
    Fh = np.zeros((P, P, P // 2 + 1), np.complex64)
    Th = np.zeros((P, P, P // 2 + 1), np.float32)
    for l in range(nImg):
        for m in range(mReco):
            src = O.translate(np.float32(-tran[l, m, 0]), np.float32(-tran[l, m, 1]), N, pl["iCol"], pl["iRow"], src=dat[l])
            O.insertP(Fh, Th, P, src, ctf[l], rots[l, m], w[l], pl["iColPad"], pl["iRowPad"])
    F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    ops.insert(F, Tt, P, T(dat, dev), T(ctf, dev), T(w, dev), T(rots, dev), T(tran, dev), T(pl["iCol"], dev), T(pl["iRow"], dev), 2, N)
    eF = float(np.abs(F.cpu().numpy() - Fh).max() / np.abs(Fh).max())
    eT = float(np.abs(Tt.cpu().numpy() - Th).max() / np.abs(Th).max())
    print("N %d rU %d grid %d^3: inserted F %.2e T %.2e of max" % (N, rU, P, eF, eT))
    assert eF <= 1e-5 and eT <= 1e-5
    # nothing lands outside the sphere the list can reach (|k| <= pf rU + 1)
    k = np.fft.fftfreq(P, 1.0 / P)
    kk = np.sqrt(k[:, None, None] ** 2 + k[None, :, None] ** 2 + np.arange(P // 2 + 1)[None, None, :] ** 2)
    assert not np.any(Tt.cpu().numpy()[kk > 2 * rU + 2])


def _native(inp, dev):
    import test_iteration_gpu as TI
    return TI.native_from_inputs(inp, dev)


def test_set_cutoff_at_nyquist_is_the_default_path(oracle, dev):
    """thx_refine_set_cutoff(N / 2 - 2, N / 2 - 2) states what the handle does by default: two iterations with the explicit call
    before each equal two iterations without it bit for bit (FSC, both half maps, support points, sigma)."""
    O = oracle
    N, n = 32, 160
    inp = U.make_inputs(O, N, n, seed=77, mReco=12, batch=64, snr=1.0)
    res = []
    for explicit in (False, True):
        nat, _ = _native(inp, dev)
        nat.reset()
        fscs = []
        for _ in range(2):
            if explicit:
                nat.set_cutoff(N // 2 - 2, N // 2 - 2)
            fscs.append(nat.iterate().copy())
        torch.cuda.synchronize()
        assert nat.cutoff() == (N // 2 - 2, N // 2 - 2, N, 0)
        v = nat.view()
        res.append((np.stack(fscs), nat.map(0).cpu().numpy(), nat.map(1).cpu().numpy(), nat.fetch(v.r, np.float64, (n, inp["cfg"]["mLR"], 4)),
                    nat.fetch(v.sig, np.float32, (2, inp["cfg"]["nGroup"], N // 2 - 1))))
        nat.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)


def test_cutoff_arguments_are_checked(oracle, dev):
    from thunder_amd import capi
    O = oracle
    inp = U.make_inputs(O, 32, 40, seed=78, mReco=4, batch=64, snr=1.0)
    nat, _ = _native(inp, dev)
    for r, rU in ((2, 10), (16, 10), (10, 16), (10, 0)):     # r <= rL, r > N / 2 - 1, rU > N / 2 - 1, rU <= 0
        with pytest.raises(capi.ThxError):
            nat.set_cutoff(r, rU)
    nat.set_cutoff(15, 15)                                    # Optimiser::maxR() = N / 2 - 1 is allowed: _size = _N
    assert nat.cutoff()[:3] == (15, 15, 32)
    nat.set_cutoff(8, 9)
    assert nat.cutoff()[:3] == (8, 9, 22)
    v = nat.view()
    assert (v.nPxl, v.nPxlM, v.fdim) == (O.pixel_list(32, 8, 2)["nPxl"], O.pixel_list(32, 9, 0)["nPxl"], 44)
    nat.close()


def test_chain_with_growing_cutoffs(oracle, dev):
    """Three local-search iterations at N = 64 with the cut-offs a caller following Model::updateR would hand in:
    (r, rU) = (12, 14) -> (20, 22) -> (30, 30).  Every stage against oracle.Iteration with the same cut-offs (the checker of
    tests/test_iteration_gpu.py): the E-step on the r list, sigma / (no normCorrection here) with the projector cut at r, the M-step
    rows on the rU list, insertion into the (pf size)^3 grids -- 64^3, 96^3, 128^3 --, prepareTF, the four reconstructions on the
    resized grid, the FSC of rU shells, the MAP reconstruction of iteration i with the rU_{i-1} entries of iteration i - 1's
    curve, the refreshed projectors, re-centring / re-masking."""
    import test_iteration_gpu as TI
    O = oracle
    N, n = 64, 200
    inp = U.make_inputs(O, N, n, seed=164, mReco=20, batch=64, snr=0.2)
    cut = [(12, 14), (20, 22), (30, 30)]
    nat, it, outs = TI._run_chain(O, dev, inp, "N=64 cut-offs", 0.05, 0.35, searches=("local", "local", "local"), cutoffs=cut)
    assert [o["fsc"].shape[1] for o in outs] == [14, 22, 30]
    assert it.PF == 128 and nat.cutoff()[:3] == (30, 30, 64)
    # the low-resolution iterations did their job: the half maps agree with the generating map inside the first cut-off
    for h in (0, 1):
        f = U.fsc_curve(O, outs[0]["maps"][h][0], inp["ref"], N, 8)
        assert f[1:6].min() >= 0.8, f
    nat.close()


def test_chain_with_cutoffs_norm_correction_and_symmetry(oracle, dev):
    """the same with what else hangs on the radii: normCorrection ON (rNorm = min(_r, resolutionP(0.75)), slices cut at
    Projector::_maxRadius = _r) and a point group (SYMMETRIZE_FT inside rU pf + 1 on the resized grid), two iterations below Nyquist"""
    import test_iteration_gpu as TI
    O = oracle
    N, n = 32, 200
    inp = U.make_inputs(O, N, n, seed=165, mReco=16, batch=64, snr=2.0, norm_correction=1, amp_spread=0.15, sym="C4")
    cut = [(8, 9), (10, 12)]
    nat, it, outs = TI._run_chain(O, dev, inp, "N=32 C4 norm cut-offs", 0.3, 0.3, searches=("local", "local"), cutoffs=cut)
    assert "norm" in outs[1] and outs[1]["rNorm"] <= 10
    nat.close()


def test_refinement_at_low_cutoff_n256(oracle, dev):
    """256^3 box, r = rU = 48 (Reconstructor::_size = 100: a 200^3 grid on rocFFT): one iteration of 600 particles through the native
    driver as bench.py's `--cutoff 48 48` line runs it.  Size-independent properties: the lists are allocPreCalIdx's, F / T of a half
    equal the oracle's insertion of the device's own draws on the resized grid (1e-5), the FSC has 48 shells and is high where the
    signal is, the two half maps agree with the generating map inside the cut-off."""
    from thunder_amd.native import NativeRefine
    from thunder_amd.refine import RefineShard
    O = oracle
    N, n, r = 256, 600, 48
    shard = RefineShard(N, n, dev, rank=0, world=1, mLR=125, mLT=9, nPhase=3, mReco=20, batch=512, particle_filter=True, allocate=False, snr=0.1)
    ref = shard.ref.cpu().numpy()
    shard.release_generation_state()
    shard.wg_per_cu = 2
    nat = NativeRefine(shard, norm_correction=False)
    nat.set_cutoff(r, r)
    cap = nat.capture(maps=True)
    nat.reset()
    fsc = nat.iterate()
    torch.cuda.synchronize()
    v = nat.view()
    P = 2 * 100
    assert nat.cutoff()[:3] == (r, r, 100) and v.fdim == P
    pl, plM = O.pixel_list(N, r, shard.rL), O.pixel_list(N, r, 0)
    assert (v.nPxl, v.nPxlM) == (pl["nPxl"], plM["nPxl"])
    assert np.all(fsc[r:] == 0) and fsc[1:12].min() > 0.8, fsc[:r]
    # insertion of half 0 on the resized grid against the oracle, from the device's own rows and draws
    lo, hi = shard.ranges[0]
    nh = hi - lo
    volF = P * P * (P // 2 + 1)
    Fd = cap["Fraw"].cpu().numpy().reshape(-1)[:volF].reshape(P, P, P // 2 + 1)
    Td = cap["Traw"].cpu().numpy().reshape(-1)[:volF].reshape(P, P, P // 2 + 1)
    datM = nat.fetch(v.datM, np.complex64, (nh, plM["nPxl"]))
    ctfM = nat.fetch(v.ctfM, np.float32, (nh, plM["nPxl"]))
    recoRot = nat.fetch(v.recoRot, np.float64, (n - nh, shard.mReco, 9))       # (the draws of the LAST inserted half: half 1)
    recoTran = nat.fetch(v.recoTran, np.float64, (n - nh, shard.mReco, 2))
    off = nat.state()[0].cpu().numpy()
    # half 1's volumes follow half 0's in the capture
    Fd1 = cap["Fraw"].cpu().numpy().reshape(-1)[volF:2 * volF].reshape(P, P, P // 2 + 1)
    Td1 = cap["Traw"].cpu().numpy().reshape(-1)[volF:2 * volF].reshape(P, P, P // 2 + 1)
    datM1 = nat.fetch(v.datM, np.complex64, (n - nh, plM["nPxl"]), offset_elems=nh * plM["nPxl"])
    ctfM1 = nat.fetch(v.ctfM, np.float32, (n - nh, plM["nPxl"]), offset_elems=nh * plM["nPxl"])
    Fh = np.zeros((P, P, P // 2 + 1), np.complex64)
    Th = np.zeros((P, P, P // 2 + 1), np.float32)
    w = np.float32(np.float32(1.0) / np.float32(shard.mReco))
    for l in range(n - nh):
        for m in range(shard.mReco):
            # (offsets are zero in the first iteration: the draw's shift is what the image is moved back by)
            src = O.translate(np.float32(-recoTran[l, m, 0]), np.float32(-recoTran[l, m, 1]), N, plM["iCol"], plM["iRow"], src=datM1[l])
            O.insertP(Fh, Th, P, src, ctfM1[l], recoRot[l, m], w, plM["iColPad"], plM["iRowPad"])
    # T(0,0,0) is the sum of (images x draws) EQUAL addends w ctf(0)^2: the oracle -- like the reference -- adds them in float and
    # drifts by up to n ulp / 2 (2e-5 at 6 000 adds) where the device's fixed-point sum is exact: that one voxel is held
    # to 1e-4, every other voxel to 1e-5 of max
    dT = np.abs(Td1 - Th)
    e0 = float(dT[0, 0, 0] / Th[0, 0, 0])
    dT[0, 0, 0] = 0
    eF, eT = float(np.abs(Fd1 - Fh).max() / np.abs(Fh).max()), float(dT.max() / np.abs(Th).max())
    print("256^3, r = rU = 48, grid 200^3: half 1 inserted F %.2e T %.2e of max (T(0,0,0) %.2e); FSC %s" % (eF, eT, e0, np.round(fsc[:r:4], 3)))
    assert eF <= 1e-5 and eT <= 1e-5 and e0 <= 1e-4 and np.any(Fd) and np.any(Td) and datM.shape[0] == nh and ctfM.shape == datM.shape and off.shape == (n, 2)
    for h in (0, 1):
        f = _fsc_np(O, nat.map(h).cpu().numpy(), ref, N, 30)
        assert f[1:14].min() >= 0.8, f
    nat.close()


def test_classification_chain_with_cutoffs(oracle, dev):
    """A K = 2 classification below Nyquist: iteration 1 a GLOBAL search at r = 8 -- the scan runs on allocPreCalIdx(_r, _rL) itself
    (radius min(cfg.rScan = 9, r) = 8: list, CTF rows and the grid shifts' ramps re-cut), the local phases on the same list, as the
    reference has it (src/Optimiser.cpp:631) --, insertion routed per class into 44^3 grids (rU = 9); iteration 2 a local search at
    (11, 12) in the assigned classes.  Every stage against oracle.Iteration with the same cut-offs."""
    import test_iteration_gpu as TI
    O = oracle
    N, n, K = 32, 192, 2
    inp = U.make_inputs(O, N, n, seed=702, mLR=40, mLT=4, nPhase=2, mReco=16, batch=64, snr=2.0, K=K, scan=dict(nR=150, nT=6, rScan=9), balance=1)
    cut = [(8, 9), (11, 12)]
    nat, it, (out1, out2) = TI._run_chain(O, dev, inp, "K=2 cut-offs", 0.3, 0.35, searches=("global", "local"), thin=True, cutoffs=cut)
    assert nat.cutoff() == (11, 12, 28, 9) and it.rS == 9
    assert (out1["cls"] == inp["cls_true"]).mean() >= 0.8 and np.array_equal(out2["cls"], out1["cls"])
    assert out1["fsc"].shape == (K, 9) and out2["fsc"].shape == (K, 12)
    nat.close()


def test_ctf_search_chain_with_cutoffs(oracle, dev):
    """SEARCH_TYPE_CTF below Nyquist: the defocus search's pre-calculated rows (allocPreCal's ctf = true branch: frequency, defocus,
    K1, K2 on the E-step list) are re-cut with the list -- iteration 1 a local search at (9, 10), iteration 2 a CTF search with nD = 9
    at (12, 13); the draw's defocus factor enters the insertion on the resized grid.  Against oracle.Iteration with the same cut-offs."""
    import test_iteration_gpu as TI
    O = oracle
    N, n = 32, 120
    inp = U.make_inputs(O, N, n, seed=901, mReco=16, batch=40, snr=4.0)
    rng = np.random.default_rng(77)
    fac = 1.0 + 0.02 * rng.standard_normal(n)
    inp["attr"] = inp["attr"].copy()
    inp["attr"][:, 1] = (inp["attr"][:, 1] / fac).astype(np.float32)
    inp["attr"][:, 2] = (inp["attr"][:, 2] / fac).astype(np.float32)
    inp["cfg"].update(mLD=9, ctfRefineS=0.01, pfSCTF=0.5)
    nat, it, (out1, out2) = TI._run_chain(O, dev, inp, "CTF cut-offs", 0.3, 0.3, searches=("local", "ctf"), cutoffs=[(9, 10), (12, 13)])
    v = nat.view()
    d = nat.fetch(v.d, np.float64, (n, 9))
    assert np.abs(d - out2["d"]).max() <= 1e-12 and nat.cutoff()[:3] == (12, 13, 30)
    nat.close()
