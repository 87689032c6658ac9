"""Iteration-level parity of the path bench.py times: thx_refine_iterate (native driver, C ABI) against the oracle's chain
of the same iteration (oracle.Iteration: Optimiser::expectation -> allReduceSigma -> reconstructRef -> compareTwoHemispheres
-> reCentreImg / reMaskImg -> solventFlatten -> refreshProj -> resetReco, src/Optimiser.cpp:1141-1660,3405-3530,6395-7766,
3800-4073) on identical particles, with the device's Philox draws replayed (tests/_philox.py).

Compared, value by value: the re-masked stack and projector after reset; every support point's weight in every phase of
every image; the filter's variances; resampled indices and top points (tie rule: tests/_iter_util.py); sigma tables; the
inserted F / T of both halves; the two MAP-off half maps, the FSC, the two MAP-on maps after averaging / flattening, the
number of balancing rounds; the refreshed projectors; offsets, shifted support points and the re-centred, re-masked stack.
Two iterations: the second one runs with non-zero offsets, updated sigma tables and the first iteration's FSC in the Wiener term.
"""
import ctypes as C
import types

import numpy as np
import pytest

import _iter_util as U

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def native_from_inputs(inp, dev):
    """NativeRefine over host inputs (no RefineShard: nothing here is generated on the device)"""
    from thunder_amd.native import NativeRefine
    c = inp["cfg"]
    s = types.SimpleNamespace(**{k: c[k] for k in ("N", "pf", "nImg", "mLR", "mLT", "nPhase", "mReco", "batch", "rL", "nGroup",
                                                   "pixelSize", "maskRadiusPx", "transS", "transQ", "pfL", "pfS", "peakFactorR")})
    s.world, s.ranges, s.dev = 1, {0: (0, c["nHalfA"]), 1: (c["nHalfA"], c["nImg"])}, dev
    s.groupSig, s.wg_per_cu, s.sigma2, s.pf_seed, s.use_pf = bool(c["groupSig"]), 2, c["sigma2Init"], c["seed"], True
    s.coreFSC, s.goldenAverage, s.solventFlatten = c["coreFSC"], c["goldenAverage"], c["solventFlatten"]
    s.gid = inp["gid"]
    s.imgOri, s.attr, s.ref = T(inp["imgOri"], dev), T(inp["attr"], dev), T(inp["ref"], dev)
    s.pf0 = dict(r=T(inp["quat0"], dev), t=T(inp["tran0"], dev))
    return NativeRefine(s), s


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def _check_iteration(O, nat, it, cap, inp, label, max_degenerate, max_adopted):
    c = inp["cfg"]
    N, n, P, rU = c["N"], c["nImg"], 2 * c["N"], c["N"] // 2 - 2
    st0 = nat.stats(reset=True)
    fsc_dev = nat.iterate()
    torch.cuda.synchronize()
    capn = {k: v.cpu().numpy() for k, v in cap.items() if v is not None}
    fol = U.Follower(O, capn, c)
    dev_rounds = list(nat.stats().lastRounds)
    out = it.iterate(fol, force_rounds=dev_rounds)
    v = nat.view()
    # ---- the local search: every weight of every phase was checked inside the follower ----
    assert fol.n_checked == c["nPhase"] * n
    frac = len(fol.adopted) / float(fol.n_checked)
    print("%s: %d image-phases, weights within %.2g (bar %.2g..), %d adopted by the tie rule (%.2f %%): %s"
          % (label, fol.n_checked, fol.max_rel, U.weight_bar(0.0), len(fol.adopted), 100 * frac, fol.adopted[:8]))
    assert frac <= max_adopted
    # images whose resampled cloud had collapsed (degenerate-cloud rule of tests/_iter_util.py)
    dg = fol.degenerate
    ma = np.asarray(fol.mean_angles)
    print("%s: mean frames of Particle::perturb: median %.1e, 90 %% %.1e, max %.1e rad apart; %d collapsed clouds beyond %.0e"
          % (label, np.median(ma[:, 0]), np.percentile(ma[:, 0], 90), ma[:, 0].max(), len(dg), fol.max_mean_angle))
    pe = np.asarray(fol.prior_err)
    print("%s: rotation priors (balanceWeight) device vs oracle: median %.1e, 90 %% %.1e, max %.1e relative" % (label, np.median(pe), np.percentile(pe, 90), pe.max()))
    for lo_, hi_ in ((0, 8), (8, 16), (16, 32), (32, 64), (64, 126)):
        sel = ma[(ma[:, 1] >= lo_) & (ma[:, 1] < hi_)]
        if len(sel):
            print("      %3d-%3d distinct incoming rotations: %4d clouds, angle median %.1e max %.1e, largest multiplicity up to %d"
                  % (lo_, hi_ - 1, len(sel), np.median(sel[:, 0]), sel[:, 0].max(), sel[:, 2].max()))
    assert len(dg) <= max_degenerate * fol.n_checked
    # Particle::calVari of every phase: an ACG fixed point whose rounds invert a matrix of condition ~1e5 by cofactors
    np.testing.assert_allclose(capn["k123"], out["k"], rtol=2e-3)
    assert np.median(np.abs(capn["k123"] / out["k"] - 1)) <= 1e-6
    np.testing.assert_allclose(capn["s01"], out["s"], rtol=1e-10)
    # filter state after the iteration: support points (shifts already re-centred), top rotation, offsets
    off, topR, topT = [x.cpu().numpy() for x in nat.state()]
    assert np.abs(nat.fetch(v.r, np.float64, (n, c["mLR"], 4)) - out["q"]).max() <= 1e-12
    assert np.abs(nat.fetch(v.t, np.float64, (n, c["mLT"], 2)) - out["t"]).max() <= 1e-9
    assert np.abs(topR - out["topR"]).max() <= 1e-12 and np.abs(off - out["offset"]).max() <= 1e-9 and np.abs(topT).max() <= 1e-12
    # ---- allReduceSigma: shell sums in another order + 2-ulp ramps / CTF (the bar of test_sigma_update) ----
    sig = nat.fetch(v.sig, np.float32, (2, c["nGroup"], N // 2 - 1))
    np.testing.assert_allclose(sig, out["sig"], rtol=2e-5)
    # ---- insertion: the accumulators F / T of both halves as the insertion left them, 1e-5 of the largest value ----
    volN = P * P * (P // 2 + 1)
    t0ratio = []
    for h in (0, 1):
        eF, eT = _rel(capn["Fraw"][h], out["F_raw"][h]), _rel(capn["Traw"][h], out["T_raw"][h])
        t0ratio.append(float(capn["Traw"][h][0, 0, 0]) / float(out["T_raw"][h][0, 0, 0]))
        print("%s: half %d inserted F %.2e T %.2e of max; T(0,0,0) device / oracle - 1 = %.2e" % (label, h, eF, eT, t0ratio[h] - 1))
        assert eF <= 1e-5 and eT <= 1e-5
    # after prepareTF (sf = 1 / T(0,0,0), src/Reconstructor.cpp:2455-2476) and the Wiener term: T(0,0,0) is the sum of
    # nImg * mReco EQUAL addends w ctf(0)^2, which the reference (and the oracle) accumulates in RFLOAT -- a sum of equal terms
    # rounds the same way every time, so it drifts by up to n ulp / 2 (~2e-5 at 2 000 adds) where the device's fixed-point sum is
    # exact; sf spreads that ratio over both volumes (the maps are invariant under a common factor of F and T)
    for h in (0, 1):
        Fd = nat.fetch(v.F, np.complex64, (P, P, P // 2 + 1), offset_elems=h * volN)
        Td = nat.fetch(v.T, np.float32, (P, P, P // 2 + 1), offset_elems=h * volN)
        eF, eT = _rel(Fd * np.float32(t0ratio[h]), out["F"][h]), _rel(Td * np.float32(t0ratio[h]), out["T"][h])
        print("%s: half %d F %.2e T %.2e of max after prepareTF (common factor %.2e removed)" % (label, h, eF, eT, t0ratio[h] - 1))
        assert eF <= 1e-5 and eT <= 1e-5 and abs(t0ratio[h] - 1) <= 1e-4
    # ---- reconstructions ----
    # Reconstructor::reconstruct ends its balancing loop on a MAX norm over the sphere (checkC, src/Reconstructor.cpp:2563-2592)
    # compared with 0.95 x its previous value (:1530-1551).  With a few hundred particles that norm sits on rim voxels whose T is
    # 1e-6 of the largest and hovers around 0.93 - 0.99 with spikes: the loop is nowhere near converged when the rule fires,
    # and WHICH round it fires in changes under 1e-7 relative noise on F / T -- the level of the reference's own unordered
    # `omp atomic` float adds (tests/test_iteration_cpu.py::test_stop_rule_is_noise_sensitive measures it on the oracle: 2 of 3
    # trials stop 3 rounds earlier, the map moves by 9 % of its maximum).  So: the round counts are compared and reported; where
    # they differ the oracle is run again for exactly the device's number of rounds (oracle.reconstruct(force_rounds=)) and
    # the maps are compared after the SAME round.  Identical-input reconstruction is held to 1e-4 in test_parity_gpu.py.
    same_rounds = dev_rounds == out["rounds"]
    print("%s: balancing rounds device %s oracle %s%s" % (label, dev_rounds, out["rounds"], "" if same_rounds else "  (oracle re-run at the device's)"))
    assert nat.stats().balancingRounds == sum(dev_rounds)
    assert all(10 < r_ <= 30 for r_ in dev_rounds)     # MIN_N_ITER_BALANCE, MAX_N_ITER_BALANCE
    ref_ = out if same_rounds else out["forced"]
    for h in (0, 1):
        for name, dv, ov in (("MAP off", capn["mapsFsc"][h], ref_["mapsFsc"][h]), ("final", nat.map(h).cpu().numpy(), ref_["maps"][h])):
            e = _rel(dv, ov)
            f = U.fsc_curve(O, dv, ov, N, rU)
            print("%s: half %d %s map %.2e of max, min FSC %.6f" % (label, h, name, e, f.min()))
            # measured with equal round counts: 8e-6 ... 2e-3 of max, FSC >= 0.9997 (the lowest on shells beyond the signal);
            # with the oracle forced to the device's count (its own rule had stopped elsewhere: two trajectories of a loop
            # that is not converging; seen at N = 64 / 200 particles in the second iteration, where the device's own count
            # changes from run to run) up to 3e-2 of max on single voxels and FSC >= 0.978 on the outermost shells
            assert e <= (5e-3 if same_rounds else 1e-1) and f.min() >= (0.999 if same_rounds else 0.95)
    # the FSC of the iteration (core-mask corrected: two more FFT round trips of the maps above)
    assert np.all(fsc_dev[rU:] == 0)
    print("%s: FSC dev %s\n      oracle %s" % (label, np.round(fsc_dev[:rU], 4), np.round(ref_["fsc"], 4)))
    np.testing.assert_allclose(fsc_dev[:rU], ref_["fsc"], atol=5e-3 if same_rounds else 5e-2)
    # compareTwoHemispheres on identical maps: the oracle's curve from the DEVICE's two MAP-off maps (replayed phases)
    own = it.fsc_of_maps(capn["mapsFsc"][0], capn["mapsFsc"][1], it.iterCount - 1)
    np.testing.assert_allclose(fsc_dev[:rU], own, atol=2e-4)
    # ---- Model::refreshProj: the projector of the next iteration, from the device's own final maps ----
    nv = P * P * (P // 2 + 1)
    for h in (0, 1):
        vd = nat.fetch(v.vols, np.complex64, (P, P, P // 2 + 1), offset_elems=h * nv)
        want = O.set_projectee(nat.map(h).cpu().numpy(), 2)
        assert _rel(vd, want) <= 2e-6
        assert _rel(vd, O.set_projectee(ref_["maps"][h], 2)) <= (5e-3 if same_rounds else 1e-1)
        it.vols[h] = want                                   # the chain continues from the device's reference ...
    it.fscReco = fsc_dev[:rU].astype(np.float32).copy()     # ... and the device's FSC (Model::resetReco)
    # ---- reCentreImg + reMaskImg ----
    img = nat.fetch(v.img, np.complex64, (n, N, N // 2 + 1))
    sc = np.abs(out["img"]).reshape(n, -1).max(1)[:, None, None]
    assert (np.abs(img - out["img"]) / sc).max() <= 1e-5
    return out


@pytest.mark.parametrize("N,n,batch,snr,max_degenerate,max_adopted", [(32, 240, 50, 2.0, 0.25, 0.25), (64, 200, 64, 0.2, 0.05, 0.35)])
def test_iteration_matches_oracle_chain(oracle, dev, N, n, batch, snr, max_degenerate, max_adopted):
    """N = 64: every cloud keeps >= 9 distinct rotations through resampling (the regime of the bench: ~40 of 125) and the
    chain is compared without exception.  N = 32 (32 x 32 images, 700 pixels) is the hard case for the filter: a few per cent
    of the clouds collapse onto 1-4 points, where the reference's own arithmetic (a 4 x 4 inverse of a singular matrix) is
    undetermined -- those perturbations are counted and bounded."""
    O = oracle
    inp = U.make_inputs(O, N, n, seed=100 + N, mReco=20, batch=batch, snr=snr)
    c = inp["cfg"]
    it = U.oracle_chain(O, inp)
    nat, shim = native_from_inputs(inp, dev)
    cap = nat.capture()
    nat.reset()
    torch.cuda.synchronize()
    v = nat.view()
    # state before the first iteration: masked stack (Optimiser::initImg), projector (Projector::setProjectee), rows
    P, rU = 2 * N, N // 2 - 2
    img = nat.fetch(v.img, np.complex64, (n, N, N // 2 + 1))
    sc = np.abs(it.img).reshape(n, -1).max(1)[:, None, None]
    assert (np.abs(img - it.img) / sc).max() <= 5e-6
    vd = nat.fetch(v.vols, np.complex64, (P, P, P // 2 + 1))
    assert _rel(vd, it.vols[0]) <= 2e-6
    assert (v.nPxl, v.nPxlM) == (it.pl["nPxl"], it.plM["nPxl"])
    out1 = _check_iteration(O, nat, it, cap, inp, "N=%d iteration 1" % N, max_degenerate, max_adopted)
    # the MAP reconstruction of the first iteration used the all-ones FSC of Model::initProjReco, the second one uses the
    # first iteration's curve (Model::resetReco)
    assert out1["fsc"][rU // 2:].min() < 0.5 and it.fscReco[0] > 0.99
    out2 = _check_iteration(O, nat, it, cap, inp, "N=%d iteration 2" % N, max_degenerate, max_adopted)
    # and the chain does what an EM iteration should: the half maps agree with the generating map at low resolution
    for h in (0, 1):
        f = U.fsc_curve(O, out2["maps"][h], inp["ref"], N, 6)
        assert np.all(f[1:5] > 0.9), f
    nat.close()


def test_iteration_against_committed_fixture(oracle, dev):
    """fixture-only variant: tests/golden/iteration_n32.npz holds the oracle's chain of two iterations run on its own (every
    discrete decision the oracle's).  The device cannot be led through it -- a resampling that lands on the other side of a
    threshold sends that image down another path (a few per cent of the image-phases, see the tie rule) -- so beyond the
    first phase the comparison is at the level the iteration is judged by: sigma tables, FSC curve, half maps."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden_iteration as G
    O = oracle
    gold = np.load(os.path.join(here, "golden", "iteration_n32.npz"))
    p = G.PARAMS
    inp = U.make_inputs(O, p["N"], p["n"], seed=p["seed"], mReco=p["mReco"], batch=p["batch"], snr=p["snr"])
    assert G.input_hash(inp).encode() == bytes(gold["input_sha256"]), "the seeded inputs changed"
    N, n, rU = p["N"], p["n"], p["N"] // 2 - 2
    nat, shim = native_from_inputs(inp, dev)
    cap = nat.capture(maps=False)
    nat.reset()
    for i in (1, 2):
        fsc = nat.iterate()
        torch.cuda.synchronize()
        if i == 1:
            # phase 0 of the first iteration depends on the inputs alone: weights of every support point of every image
            uR, uT = cap["uR"][0].cpu().numpy(), cap["uT"][0].cpu().numpy()
            gR, gT = gold["it1_uR"][0], gold["it1_uT"][0]
            eR = np.abs(uR - gR).max(1) / gR.max(1)
            eT = np.abs(uT - gT).max(1) / gT.max(1)
            print("fixture: phase-0 weights within %.1e / %.1e of each image's largest (median %.1e)" % (eR.max(), eT.max(), np.median(eR)))
            assert np.median(eR) <= 1e-3 and np.mean(eR <= 2e-2) >= 0.9 and np.mean(eT <= 2e-2) >= 0.9
        off, topR, _ = [x.cpu().numpy() for x in nat.state()]
        # (support points differ by the numerical latitude of Particle::perturb's mean frame: up to ~1e-3 rad, see the
        # mean-frame rule in tests/_iter_util.py; the same support point is the one within 5e-3)
        same = np.mean(np.abs(topR - gold["it%d_topR" % i]).max(1) <= 5e-3)
        sig = nat.fetch(nat.view().sig, np.float32, gold["it%d_sig" % i].shape)
        fs = [U.fsc_curve(O, nat.map(h).cpu().numpy(), gold["it%d_maps" % i][h], N, rU) for h in (0, 1)]
        print("fixture iteration %d: same top rotation for %.0f %% of the images; sigma within %.1e; map FSC vs fixture %s; FSC curve within %.1e"
              % (i, 100 * same, np.abs(sig / gold["it%d_sig" % i] - 1).max(), np.round(np.minimum(fs[0], fs[1])[:8], 4),
                 np.abs(fsc[:rU] - gold["it%d_fsc" % i]).max()))
        assert same >= (0.7 if i == 1 else 0.4)
        np.testing.assert_allclose(sig, gold["it%d_sig" % i], rtol=0.1)
        assert np.all(np.minimum(fs[0], fs[1])[:6] >= 0.99)
        np.testing.assert_allclose(fsc[:6], gold["it%d_fsc" % i][:6], atol=2e-2)
        assert np.sqrt(((off - gold["it%d_offset" % i]) ** 2).mean()) <= 0.5
    nat.close()
