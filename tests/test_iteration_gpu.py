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


def _check_iteration(O, nat, it, cap, inp, label):
    c = inp["cfg"]
    N, n, P, rU = c["N"], c["nImg"], 2 * c["N"], c["N"] // 2 - 2
    st0 = nat.stats(reset=True)
    fsc_dev = nat.iterate()
    torch.cuda.synchronize()
    capn = {k: v.cpu().numpy() for k, v in cap.items() if v is not None}
    fol = U.Follower(O, capn, c)
    out = it.iterate(fol)
    v = nat.view()
    # ---- the local search: every weight of every phase was checked inside the follower ----
    assert fol.n_checked == c["nPhase"] * n
    frac = len(fol.adopted) / float(fol.n_checked)
    print("%s: %d image-phases, weights within %.2g (bar %.2g..), %d adopted by the tie rule (%.2f %%): %s"
          % (label, fol.n_checked, fol.max_rel, U.weight_bar(0.0), len(fol.adopted), 100 * frac, fol.adopted[:8]))
    assert frac <= 0.05
    # Particle::calVari of every phase (fixed-point ACG sums in another order: 1e-6)
    np.testing.assert_allclose(capn["k123"], out["k"], rtol=2e-6)
    np.testing.assert_allclose(capn["s01"], out["s"], rtol=1e-10)
    # filter state after the iteration: support points (shifts already re-centred), top rotation, offsets
    off, topR, topT = [x.cpu().numpy() for x in nat.state()]
    assert np.abs(nat.fetch(v.r, np.float64, (n, c["mLR"], 4)) - out["q"]).max() <= 1e-12
    assert np.abs(nat.fetch(v.t, np.float64, (n, c["mLT"], 2)) - out["t"]).max() <= 1e-9
    assert np.abs(topR - out["topR"]).max() <= 1e-12 and np.abs(off - out["offset"]).max() <= 1e-9 and np.abs(topT).max() <= 1e-12
    # ---- allReduceSigma: shell sums in another order + 2-ulp ramps / CTF (the bar of test_sigma_update) ----
    sig = nat.fetch(v.sig, np.float32, (2, c["nGroup"], N // 2 - 1))
    np.testing.assert_allclose(sig, out["sig"], rtol=2e-5)
    # ---- insertion: F / T of both halves after prepareTF's normalisation, 1e-5 of the largest accumulated value ----
    volN = P * P * (P // 2 + 1)
    for h in (0, 1):
        Fd = nat.fetch(v.F, np.complex64, (P, P, P // 2 + 1), offset_elems=h * volN)
        Td = nat.fetch(v.T, np.float32, (P, P, P // 2 + 1), offset_elems=h * volN)
        eF, eT = _rel(Fd, out["F"][h]), _rel(Td, out["T"][h])
        print("%s: half %d F %.2e T %.2e (of max)" % (label, h, eF, eT))
        assert eF <= 1e-5 and eT <= 1e-5
    # ---- reconstructions: 1e-4 of max, FSC >= 0.9999 per shell, same number of balancing rounds ----
    st = nat.stats()
    assert st.balancingRounds == sum(out["rounds"]), (st.balancingRounds, out["rounds"])
    for h in (0, 1):
        for name, dv, ov in (("MAP off", capn["mapsFsc"][h], out["mapsFsc"][h]), ("final", nat.map(h).cpu().numpy(), out["maps"][h])):
            e = _rel(dv, ov)
            f = U.fsc_curve(O, dv, ov, N, rU)
            print("%s: half %d %s map %.2e of max, min FSC %.6f" % (label, h, name, e, f.min()))
            assert e <= 1e-4 and f.min() >= 0.9999
    # the FSC of the iteration (core-mask corrected: two more FFT round trips of maps that agree to 1e-4)
    assert np.all(fsc_dev[rU:] == 0)
    print("%s: FSC dev %s\n      oracle %s" % (label, np.round(fsc_dev[:rU], 4), np.round(out["fsc"], 4)))
    np.testing.assert_allclose(fsc_dev[:rU], out["fsc"], atol=2e-3)
    # ---- Model::refreshProj: the projector of the next iteration ----
    nv = P * P * (P // 2 + 1)
    for h in (0, 1):
        vd = nat.fetch(v.vols, np.complex64, (P, P, P // 2 + 1), offset_elems=h * nv)
        assert _rel(vd, out["vols"][h]) <= 1e-4
    # ---- reCentreImg + reMaskImg ----
    img = nat.fetch(v.img, np.complex64, (n, N, N // 2 + 1))
    sc = np.abs(out["img"]).reshape(n, -1).max(1)[:, None, None]
    assert (np.abs(img - out["img"]) / sc).max() <= 1e-5
    return out


@pytest.mark.parametrize("N,n,batch,snr", [(32, 240, 50, 0.5), (64, 200, 64, 0.2)])
def test_iteration_matches_oracle_chain(oracle, dev, N, n, batch, snr):
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
    P = 2 * N
    img = nat.fetch(v.img, np.complex64, (n, N, N // 2 + 1))
    sc = np.abs(it.img).reshape(n, -1).max(1)[:, None, None]
    assert (np.abs(img - it.img) / sc).max() <= 5e-6
    vd = nat.fetch(v.vols, np.complex64, (P, P, P // 2 + 1))
    assert _rel(vd, it.vols[0]) <= 2e-6
    assert (v.nPxl, v.nPxlM) == (it.pl["nPxl"], it.plM["nPxl"])
    out1 = _check_iteration(O, nat, it, cap, inp, "N=%d iteration 1" % N)
    # the MAP reconstruction of the first iteration used the all-ones FSC of Model::initProjReco, the second one uses out1's
    assert np.array_equal(it.fscReco, out1["fsc"].astype(np.float32))
    out2 = _check_iteration(O, nat, it, cap, inp, "N=%d iteration 2" % N)
    # and the chain does what an EM iteration should: the half maps agree with the generating map at low resolution
    for h in (0, 1):
        f = U.fsc_curve(O, out2["maps"][h], inp["ref"], N, 6)
        assert np.all(f[1:5] > 0.9), f
    nat.close()
