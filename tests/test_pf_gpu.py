"""The device particle filter (thx_pf.hip) against the oracle's restatement of src/Particle.cpp /
src/Geometry/DirectionalStat.cpp.  The device's Philox draws are replayed in numpy (tests/_philox.py), so the perturbation,
shuffle and resampling are compared element by element; summation orders differ (wave tree vs serial): 1e-9 relative."""
import ctypes as C

import numpy as np
import pytest

import _philox as PH

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _cloud(rng, nImg, n, spread):
    """n unit quaternions per image scattered around a random mean with anisotropic spread"""
    from thunder_amd import synth
    mean = synth.random_quats(nImg, rng)
    d = rng.standard_normal((nImg, n, 4)) * np.array([1.0, spread, 0.6 * spread, 0.3 * spread])
    d[..., 0] = 1.0
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    # mean * d: calVari estimates k1..k3 on conj(mean) * r (LEFT multiplication, src/Particle.cpp:1052-1058)
    q = synth.quat_mul(np.broadcast_to(mean[:, None, :], d.shape), d)
    return q / np.linalg.norm(q, axis=2, keepdims=True)


def _o(O, name, *args):
    getattr(O.lib(), name)(*args)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_acg_statistics(oracle, dev):
    from thunder_amd import capi
    O = oracle
    O.lib().orc_infer_acg.restype = C.c_int
    rng = np.random.default_rng(21)
    nImg, n = 12, 125
    q = _cloud(rng, nImg, n, 0.05)
    A = torch.empty((nImg, 16), dtype=torch.float64, device=dev)
    mean = torch.empty((nImg, 4), dtype=torch.float64, device=dev)
    k = torch.empty((nImg, 3), dtype=torch.float64, device=dev)
    wb = torch.empty((nImg, n), dtype=torch.float64, device=dev)
    rounds = torch.zeros((nImg, 2), dtype=torch.int32, device=dev)
    dq = T(q, dev)
    capi.call("thx_pf_acg_stats_dev", A.data_ptr(), mean.data_ptr(), k.data_ptr(), wb.data_ptr(), rounds.data_ptr(),
              dq.data_ptr(), nImg, n, capi.stream_ptr())
    A, mean, k, wb, rounds = [x.cpu().numpy() for x in (A, mean, k, wb, rounds)]
    same = 0
    for l in range(nImg):
        Aw = np.zeros(16)
        rw = O.lib().orc_infer_acg(_dp(Aw), _dp(np.ascontiguousarray(q[l])), n)
        # the fixed point stops on sum|A - B| <= 1e-3: a rounding-level difference can move the stop by one round
        tol = 1e-9 if rw == rounds[l, 0] else 2e-3
        same += rw == rounds[l, 0]
        assert np.abs(A[l] - Aw).sum() <= tol * max(1.0, np.abs(Aw).sum())
        mw = np.zeros(4)
        _o(O, "orc_sym4_top_eigvec", _dp(mw), _dp(Aw))
        assert min(np.abs(mean[l] - mw).max(), np.abs(mean[l] + mw).max()) <= max(tol, 1e-9) * 10
        kw, mw2, qq = np.zeros(3), np.zeros(4), np.ascontiguousarray(q[l].copy())
        _o(O, "orc_cal_vari_R", _dp(kw), _dp(mw2), _dp(qq), n)
        assert np.allclose(k[l], kw, rtol=1e-7 if (rw == rounds[l, 0]) else 5e-2)
        ww = np.zeros(n)
        _o(O, "orc_balance_weight_R", _dp(ww), _dp(np.ascontiguousarray(q[l])), n)
        assert np.allclose(wb[l], ww, rtol=1e-7 if rw == rounds[l, 0] else 5e-2) and abs(wb[l].sum() - 1) < 1e-12
    assert same >= nImg - 2
    # the estimated concentration follows the spread of the cloud: k ~ spread^2 ordering k1 > k2 > k3
    assert np.all(k[:, 0] > k[:, 1]) and np.all(k[:, 1] > k[:, 2]) and np.all(k > 0) and np.all(k < 0.05)


def test_acg_collapsed_clouds(oracle, dev):
    """inferACG on clouds as resampling leaves them when a few support points take most of the weight: 125 copies of 4 - 12 distinct
    rotations, one of them holding up to a third.  Tyler's fixed point then converges linearly towards a near-singular matrix --
    hundreds to tens of thousands of rounds, in the reference as here (no round limit there; 100 000 here) -- and one such image
    holds its whole k_pf_perturb launch.  Held: where the device runs the oracle's number of rounds it lands on its matrix (1e-5);
    where the stop falls in another round the matrices still agree to 5e-2; every result is finite and the launch is bounded."""
    import time
    from thunder_amd import capi
    O = oracle
    O.lib().orc_infer_acg.restype = C.c_int
    rng = np.random.default_rng(77)
    n = 125
    clouds = []
    for m, top in ((4, 0.34), (5, 0.3), (6, 0.25), (8, 0.3), (12, 0.2), (12, 0.1)):
        base = _cloud(rng, 1, m, 0.03)[0]
        w = rng.uniform(0.2, 1.0, m)
        w[0] = top / (1 - top) * w[1:].sum()
        idx = rng.choice(m, size=n, p=w / w.sum())
        idx[:m] = np.arange(m)                      # (every point at least once)
        clouds.append(base[idx])
    q = np.ascontiguousarray(np.stack(clouds))
    nImg = len(q)
    A = torch.empty((nImg, 16), dtype=torch.float64, device=dev)
    mean = torch.empty((nImg, 4), dtype=torch.float64, device=dev)
    k = torch.empty((nImg, 3), dtype=torch.float64, device=dev)
    wb = torch.empty((nImg, n), dtype=torch.float64, device=dev)
    rounds = torch.zeros((nImg, 2), dtype=torch.int32, device=dev)
    dq = T(q, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.call("thx_pf_acg_stats_dev", A.data_ptr(), mean.data_ptr(), k.data_ptr(), wb.data_ptr(), rounds.data_ptr(),
              dq.data_ptr(), nImg, n, capi.stream_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    A, rounds = A.cpu().numpy(), rounds.cpu().numpy()
    want, dA = [], []
    for l in range(nImg):
        Aw = np.zeros(16)
        rw = O.lib().orc_infer_acg(_dp(Aw), _dp(np.ascontiguousarray(q[l])), n)
        want.append(rw)
        dA.append(float(np.abs(A[l] - Aw).sum() / max(1.0, np.abs(Aw).sum())))
    print("collapsed clouds: rounds device %s oracle %s, |A - A_oracle| / |A_oracle| %s, %.1f ms for the launch (%.2f us per round of the slowest)"
          % (rounds[:, 0].tolist(), want, ["%.1e" % x for x in dA], dt * 1e3, dt * 1e6 / max(1, rounds[:, 0].max())))
    # measured: rounds device [3454, 28, 53, 10, 9, 8] oracle [189, 28, 112, 10, 9, 8]; |A - A_oracle| / |A_oracle| 9e-3, 1e-7, 3e-3, 2e-6,
    # 2e-8, 8e-12.  With 4 - 6 distinct points the iterates crawl along a nearly flat direction of the fixed-point map and WHEN
    # sum|A - B| first drops under 1e-3 is decided by rounding (the device forms the inverse from ten cofactors and a reciprocal,
    # the oracle from sixteen and divisions), yet the matrices returned agree to a per cent -- the regime of the chain tests'
    # collapsed-cloud rule
    for l in range(nImg):
        assert dA[l] <= (1e-5 if want[l] == rounds[l, 0] else 5e-2), (l, dA[l], want[l], rounds[l, 0])
    assert np.all(np.isfinite(A)) and max(want) > 100 and rounds.max() < 100000 and dt < 1.0


def _state(rng, nImg, nR, nT):
    q = _cloud(rng, nImg, nR, 0.03)
    t = rng.normal(0, 1.2, size=(nImg, nT, 2))
    wR = rng.uniform(0.5, 1.5, size=(nImg, nR)); wR /= wR.sum(1, keepdims=True)
    wT = rng.uniform(0.5, 1.5, size=(nImg, nT)); wT /= wT.sum(1, keepdims=True)
    k = rng.uniform(1e-4, 2e-3, size=(nImg, 3))
    s = rng.uniform(0.5, 1.5, size=(nImg, 2))
    return q, t, wR, wT, k, s


def test_perturb(oracle, dev):
    from thunder_amd import capi, synth
    O = oracle
    rng = np.random.default_rng(22)
    nImg, nR, nT, seed, call = 7, 125, 9, 0x1234567890ABCDEF, 5
    q, t, wR, wT, k, s = _state(rng, nImg, nR, nT)
    k[0] = (3.0, 0.5, 2.0)                               # exercises min(PERTURB_K_MAX = 1, k)
    pfR, pfT, transS, transQ = 2.0, 0.5, 2.0, 0.05
    dq, dt, dwR, dwT, dk, ds = T(q, dev), T(t, dev), T(wR, dev), T(wT, dev), T(k, dev), T(s, dev)
    capi.call("thx_pf_perturb_dev", dq.data_ptr(), dt.data_ptr(), dwR.data_ptr(), dwT.data_ptr(), dk.data_ptr(),
              ds.data_ptr(), nImg, nR, nT, pfR, pfT, transS, transQ, seed, call, None, capi.stream_ptr())
    gq, gt, gwR, gwT = [x.cpu().numpy() for x in (dq, dt, dwR, dwT)]
    for l in range(nImg):
        # Particle::perturb in the reference's own order: mean = inferACG(mean, r); r <- mean * (pert * (conj(mean) * r))
        # (src/Particle.cpp:1185-1243), shifts + reCentre (:1244-1272, 2473-2495), then balanceWeight for both
        gR = np.stack(PH.draw_n4(seed, l, call, 0, np.arange(nR)), axis=1)
        gT = np.stack(PH.draw_n4(seed, l, call, 1, np.arange(nT)), axis=1)
        wq, wt, wwR, wwT = O.pf_perturb(q[l], t[l], k[l], s[l], pfR, pfT, transS, transQ, gR, gT)
        # (the conjugation by the cloud's mean inherits the 1e-9 accuracy of the ACG estimate, see test_acg_statistics)
        assert np.abs(gq[l] - wq).max() <= 2e-9 and np.abs(np.linalg.norm(gq[l], axis=1) - 1).max() < 1e-12
        assert np.abs(gt[l] - wt).max() <= 1e-12
        assert np.allclose(gwR[l], wwR, rtol=1e-7) and abs(gwR[l].sum() - 1) < 1e-12
        assert np.allclose(gwT[l], wwT, rtol=1e-9)
        # the perturbation is NOT pert * r: it acts in the frame of the cloud's mean (conjugated by it)
        v = np.stack([gR[:, 0], gR[:, 1] * np.sqrt(pfR ** 2 * min(1.0, k[l, 0])), gR[:, 2] * np.sqrt(pfR ** 2 * min(1.0, k[l, 1])),
                      gR[:, 3] * np.sqrt(pfR ** 2 * min(1.0, k[l, 2]))], axis=1)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        assert np.abs(synth.quat_mul(v, q[l]) - wq).max() > 1e-6
    transM = transS * (-2.0 * np.log(transQ))
    assert np.all(np.hypot(gt[..., 0], gt[..., 1]) <= max(transM, 6 * transS))


def test_update_resample(oracle, dev):
    from thunder_amd import capi
    O = oracle
    rng = np.random.default_rng(23)
    nImg, nR, nT, seed, call, peak = 6, 125, 9, 987654321, 11, 1e-3
    q, t, wR, wT, k, s = _state(rng, nImg, nR, nT)
    uR = (rng.uniform(0, 1, (nImg, nR)) ** 8).astype(np.float32)      # peaked likelihood weights
    uT = rng.uniform(0.05, 1, (nImg, nT)).astype(np.float32)
    dq, dt, dwR, dwT, dk, ds = [T(x, dev) for x in (q, t, wR, wT, k, s)]
    topR = torch.zeros((nImg, 4), dtype=torch.float64, device=dev)
    topT = torch.zeros((nImg, 2), dtype=torch.float64, device=dev)
    duR, duT = T(uR, dev), T(uT, dev)
    capi.call("thx_pf_update_dev", dq.data_ptr(), dt.data_ptr(), dwR.data_ptr(), dwT.data_ptr(), duR.data_ptr(),
              duT.data_ptr(), dk.data_ptr(), ds.data_ptr(), topR.data_ptr(), topT.data_ptr(), nImg, nR, nT, peak,
              seed, call, None, capi.stream_ptr())
    gq, gt, gwR, gwT, gk, gs, gtopR, gtopT = [x.cpu().numpy() for x in (dq, dt, dwR, dwT, dk, ds, topR, topT)]
    for l in range(nImg):
        # rank-1st, variances
        # (_topR is taken after calVari's round trip through the mean frame, as resample() does: 1e-16 of noise)
        assert np.abs(gtopR[l] - q[l, uR[l].argmax()]).max() <= 1e-14 and np.array_equal(gtopT[l], t[l, uT[l].argmax()])
        kw, mw, qq = np.zeros(3), np.zeros(4), np.ascontiguousarray(q[l].copy())
        O.lib().orc_cal_vari_R(_dp(kw), _dp(mw), _dp(qq), nR)
        assert np.allclose(gk[l], kw, rtol=5e-2)
        sw = np.zeros(2)
        O.lib().orc_cal_vari_T(_dp(sw), _dp(np.ascontiguousarray(t[l])), nT)
        assert np.allclose(gs[l], sw, rtol=1e-12)
        # rotations: keepHalfHeightPeak, shuffle (replayed), systematic resampling with the replayed u0
        u = uR[l].astype(np.float64)
        O.lib().orc_keep_half_height_peak(_dp(u), nR, C.c_double(peak))
        rank = PH.shuffle_ranks(seed, l, call, 2, nR)
        inv = np.empty(nR, np.int64); inv[rank] = np.arange(nR)       # shuffled[j] = original[inv[j]]
        u0 = PH.draw_u4(seed, l, call, 3, 0)[0] / nR
        idx = np.zeros(nR, np.int32); wo = np.zeros(nR)
        O.lib().orc_resample(idx.ctypes.data_as(C.POINTER(C.c_int)), _dp(wo), _dp(np.ascontiguousarray(wR[l][inv])),
                             _dp(np.ascontiguousarray(u[inv])), nR, nR, C.c_double(float(u0)))
        src = inv[idx]
        # (calVari's round trip through the mean frame perturbs the quaternions at 1e-16)
        assert np.abs(gq[l] - q[l][src]).max() <= 1e-13
        assert np.allclose(gwR[l], wo, rtol=1e-10) and abs(gwR[l].sum() - 1) < 1e-12
        assert np.all(u[src] > 0)                                     # nothing below the half-height cut survives
        # shifts
        rank = PH.shuffle_ranks(seed, l, call, 4, nT)
        inv = np.empty(nT, np.int64); inv[rank] = np.arange(nT)
        u0 = PH.draw_u4(seed, l, call, 5, 0)[0] / nT
        idx = np.zeros(nT, np.int32); wo = np.zeros(nT)
        uu = uT[l].astype(np.float64)
        O.lib().orc_resample(idx.ctypes.data_as(C.POINTER(C.c_int)), _dp(wo), _dp(np.ascontiguousarray(wT[l][inv])),
                             _dp(np.ascontiguousarray(uu[inv])), nT, nT, C.c_double(float(u0)))
        assert np.array_equal(gt[l], t[l][inv[idx]]) and np.allclose(gwT[l], wo, rtol=1e-10)
    # reproducible: same seed / call -> same result; another call id -> another shuffle
    dq2, dt2, dwR2, dwT2, dk2, ds2 = [T(x, dev) for x in (q, t, wR, wT, k, s)]
    capi.call("thx_pf_update_dev", dq2.data_ptr(), dt2.data_ptr(), dwR2.data_ptr(), dwT2.data_ptr(),
              duR.data_ptr(), duT.data_ptr(), dk2.data_ptr(), ds2.data_ptr(), topR.data_ptr(),
              topT.data_ptr(), nImg, nR, nT, peak, seed, call, None, capi.stream_ptr())
    assert np.array_equal(dq2.cpu().numpy(), gq) and np.array_equal(dwT2.cpu().numpy(), gwT)


def test_stop_rule_matches_oracle(oracle, dev):
    """the per-image stop rule (src/Optimiser.cpp:1510-1615) on the device against the oracle's restatement, over random
    variance histories; the active mask makes perturb / expect_local / update leave stopped images untouched"""
    import torch
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(17)
    n, transS = 300, 2.0
    active, nP, state = ops.pf_stop_init(n, transS, dev)
    st_h = [O.stop_rule_init(transS) for _ in range(n)]
    alive_h = np.ones(n, bool)
    nP_h = np.zeros(n, np.int64)
    k = np.full((n, 3), 1e-3)
    s = np.full((n, 2), 1.0)
    for phase in range(3, 14):
        # variances shrink by a random factor around the rule's threshold, some stall, some bounce back
        k = k * rng.choice([0.7, 0.9, 0.93, 1.0, 1.1], size=(n, 3))
        s = s * rng.choice([0.8, 0.94, 0.96, 1.0, 1.05], size=(n, 2))
        cnt = ops.pf_stop_rule(active, nP, state, torch.from_numpy(k).to(dev), torch.from_numpy(s).to(dev), phase)
        for l in range(n):
            if alive_h[l] and O.stop_rule(st_h[l], k[l, 0], k[l, 1], k[l, 2], s[l, 0], s[l, 1]):
                alive_h[l] = False
                nP_h[l] = phase
        assert cnt == int(alive_h.sum())
        assert np.array_equal(active.cpu().numpy().astype(bool), alive_h)
    assert np.array_equal(nP.cpu().numpy(), nP_h) and 0 < alive_h.sum() < n or alive_h.sum() == 0
    assert nP_h[~alive_h].min() >= 4      # phase 3 always finds "room" (the initial minima are 1 and 5 transS)
    # masked kernels: a stopped image's filter state is not touched
    nR, nT, m = 20, 5, 8
    from thunder_amd import synth
    q = torch.from_numpy(synth.perturb_quats(synth.random_quats(m, rng), nR, 0.05, rng)).to(dev)
    t = torch.from_numpy(rng.normal(0, 1, size=(m, nT, 2))).to(dev)
    wR = torch.full((m, nR), 1.0 / nR, dtype=torch.float64, device=dev)
    wT = torch.full((m, nT), 1.0 / nT, dtype=torch.float64, device=dev)
    kk = ops.pf_acg_stats(q)[2]
    ss = t.std(dim=1).contiguous()
    act = torch.tensor([1, 0, 1, 0, 0, 1, 1, 0], dtype=torch.int32, device=dev)
    q0, t0 = q.clone(), t.clone()
    ops.pf_perturb(q, t, wR, wT, kk, ss, 2.0, 2.0, 2.0, 0.05, 5, 1, active=act)
    off = (act == 0)
    assert torch.equal(q[off], q0[off]) and torch.equal(t[off], t0[off])
    assert not torch.equal(q[~off], q0[~off])


def test_class_select_after_scan(oracle, dev):
    """PAR_C after the global scan (src/Optimiser.cpp:925-952): keepHalfHeightPeak(PAR_C) at PEAK_FACTOR_C = 0.99, resample(k, PAR_C),
    rand(cls) -- the device kernel against the oracle with replayed draws; clear winners give the arg-max, weights within
    1 % of the largest share the image between the classes"""
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(41)
    nImg, K, seed, call = 400, 4, 99991, 3
    uC = rng.uniform(0.0, 1.0, size=(nImg, K)).astype(np.float32) ** 6
    close = rng.choice(nImg, 120, replace=False)              # near-ties: a second class within 1 % of the best
    for l in close:
        a, b = rng.choice(K, 2, replace=False)
        uC[l, a] = uC[l].max() * 1.2
        uC[l, b] = uC[l, a] * np.float32(rng.uniform(0.991, 0.9999))
    wC = rng.uniform(0.5, 1.5, size=(nImg, K))
    wC /= wC.sum(1, keepdims=True)
    got = ops.pf_class_select(T(uC, dev), seed, call, wC=T(wC, dev)).cpu().numpy()
    want = np.empty(nImg, np.int64)
    for l in range(nImg):
        rank = PH.shuffle_ranks(seed, l, call, 6, K)
        u0 = PH.draw_u4(seed, l, call, 7, 0)[0] / K
        pick = min(int(PH.draw_u4(seed, l, call, 8, 0)[0] * K), K - 1)
        want[l] = O.pf_class_select(uC[l], wC[l], 1.0 - 1e-2, rank, u0, pick)
    assert np.array_equal(got, want)
    clear = np.setdiff1d(np.arange(nImg), close)
    assert np.array_equal(got[clear], uC[clear].argmax(1))
    second = np.mean(got[close] != uC[close].argmax(1))
    assert 0.02 < second < 0.6, second                         # the runner-up within 1 % keeps a share (u - 0.99 max)


def test_scan_support_points(oracle, dev):
    """thx_pf_scan_support_dev against the oracle's restatement of src/Optimiser.cpp:953-1008: the scanned grid (10 000 rotations /
    30 shifts in one case: the bitonic shuffle at its full size) with peaked scan weights of K classes -> keepHalfHeightPeak,
    resample(mLR) / resample(mLT) with the replayed shuffle and u0, calVari.  The cumulative sums are formed in the reference's
    order (serial), the normalising sum in a wave tree: a resampled index may sit on its threshold -- such a draw must then be an
    immediate neighbour in the shuffled order, and there are at most a handful of them."""
    from thunder_amd import ops, synth
    O = oracle
    for nImg, nK, nRin, nTin, mLR, mLT, seed_np in ((5, 2, 700, 30, 125, 9, 31), (2, 1, 10000, 30, 125, 9, 32)):
        rng = np.random.default_rng(seed_np)
        seed, call, peak = 13579, 7, 1e-3
        gridR = synth.random_quats(nRin, rng)
        gridT = np.ascontiguousarray(rng.normal(0, 3.0, size=(nTin, 2)))
        # scan posterior: a few strong modes on a weak background
        uR = (rng.uniform(0, 1, (nK, nImg, nRin)) ** 40).astype(np.float32)
        uT = (rng.uniform(0.01, 1, (nK, nImg, nTin)) ** 3).astype(np.float32)
        cls = rng.integers(0, nK, nImg).astype(np.int32)
        minK, minS = (2.0 * nRin ** (-1.0 / 3)) ** 2, 0.35          # the scanning phase's minimum spread (one of the k below it in case 2)
        st = ops.pf_scan_support(T(gridR, dev), T(gridT, dev), T(uR, dev), T(uT, dev), T(cls, dev), mLR, mLT, peak, seed, call, minK, minS)
        g = {k: v.cpu().numpy() for k, v in st.items()}
        near = 0
        for l in range(nImg):
            rankR, rankT = PH.shuffle_ranks(seed, l, call, 2, nRin), PH.shuffle_ranks(seed, l, call, 4, nTin)
            w = O.pf_scan_support(gridR, gridT, uR[cls[l], l], uT[cls[l], l], peak, mLR, mLT, rankR, PH.draw_u4(seed, l, call, 3, 0)[0] / mLR,
                                  rankT, PH.draw_u4(seed, l, call, 5, 0)[0] / mLT, minK, minS)
            assert np.array_equal(g["topR"][l], w["topR"]) and np.array_equal(g["topT"][l], w["topT"])
            # which grid point every device support point is (before calVari's round trip: 1e-16 of noise on the quaternion)
            dR = np.abs(g["r"][l][:, None, :] - gridR[None, w["srcR"], :]).max(axis=2)      # [mLR device][mLR oracle]
            same = dR.diagonal() <= 1e-13
            for j in np.nonzero(~same)[0]:     # a draw on its threshold: the neighbour in the shuffled order
                srcD = int(np.argmin(np.abs(gridR - g["r"][l][j]).max(axis=1)))
                assert abs(int(rankR[srcD]) - int(rankR[w["srcR"][j]])) <= 1, (l, j)
                near += 1
            if same.all():
                assert np.allclose(g["wR"][l], w["wR"], rtol=1e-9) and np.allclose(g["k"][l], w["k"], rtol=5e-2)
            assert np.all(w["uRk"][w["srcR"]] > 0) and abs(g["wR"][l].sum() - 1) < 1e-12
            assert np.array_equal(g["t"][l], w["t"]) and np.allclose(g["wT"][l], w["wT"], rtol=1e-9)
            assert np.allclose(g["s"][l], w["s"], rtol=1e-12)
        assert near <= 2
        # reproducible; another call id -> another shuffle
        st2 = ops.pf_scan_support(T(gridR, dev), T(gridT, dev), T(uR, dev), T(uT, dev), T(cls, dev), mLR, mLT, peak, seed, call, minK, minS)
        assert all(torch.equal(st[k], st2[k]) for k in st)
        st3 = ops.pf_scan_support(T(gridR, dev), T(gridT, dev), T(uR, dev), T(uT, dev), T(cls, dev), mLR, mLT, peak, seed, call + 1, minK, minS)
        assert not torch.equal(st["r"], st3["r"])


def test_defocus_filter(oracle, dev):
    """PAR_D of the CTF search: Particle::initD / perturb(PAR_D) + balanceWeight(PAR_D), then calRank1st / calVari / resample(PAR_D)
    -- device kernels against the oracle with replayed draws"""
    from thunder_amd import capi
    O = oracle
    rng = np.random.default_rng(51)
    nImg, nD, seed = 50, 9, 424242
    d = torch.zeros((nImg, nD), dtype=torch.float64, device=dev)
    wD = torch.zeros((nImg, nD), dtype=torch.float64, device=dev)
    capi.call("thx_pf_perturb_d_dev", d.data_ptr(), wD.data_ptr(), None, nImg, nD, 0.01, 1, seed, 1, None, capi.stream_ptr())
    d0, w0 = d.cpu().numpy(), wD.cpu().numpy()
    for l in range(nImg):
        g = PH.draw_n4(seed, l, 1, 10, np.arange(nD))[0]
        wd, ww = O.pf_perturb_d(None, 0.0, 0.01, True, g)
        assert np.abs(d0[l] - wd).max() <= 1e-15 and np.allclose(w0[l], ww, rtol=1e-10)
    assert abs(d0.mean() - 1) < 5e-3 and 0.007 < d0.std() < 0.013
    # a phase: likelihood weights -> top point, spread, resampling; then the next perturbation uses that spread
    uD = (rng.uniform(0, 1, (nImg, nD)) ** 3).astype(np.float32)
    sD = torch.zeros(nImg, dtype=torch.float64, device=dev)
    topD = torch.zeros(nImg, dtype=torch.float64, device=dev)
    duD = T(uD, dev)
    capi.call("thx_pf_update_d_dev", d.data_ptr(), wD.data_ptr(), duD.data_ptr(), sD.data_ptr(), topD.data_ptr(), nImg, nD, seed, 2,
              None, capi.stream_ptr())
    d1, w1, s1, t1 = d.cpu().numpy(), wD.cpu().numpy(), sD.cpu().numpy(), topD.cpu().numpy()
    for l in range(nImg):
        rank = PH.shuffle_ranks(seed, l, 2, 11, nD)
        u0 = PH.draw_u4(seed, l, 2, 12, 0)[0] / nD
        wd, ww, ws, wt, _ = O.pf_update_d(d0[l], w0[l], uD[l], rank, u0)
        assert np.array_equal(d1[l], wd) and np.allclose(w1[l], ww, rtol=1e-10) and abs(s1[l] - ws) <= 1e-15 and t1[l] == wt
    capi.call("thx_pf_perturb_d_dev", d.data_ptr(), wD.data_ptr(), sD.data_ptr(), nImg, nD, 0.5, 0, seed, 3, None, capi.stream_ptr())
    d2, w2 = d.cpu().numpy(), wD.cpu().numpy()
    for l in range(nImg):
        g = PH.draw_n4(seed, l, 3, 10, np.arange(nD))[0]
        wd, ww = O.pf_perturb_d(d1[l], s1[l], 0.5, False, g)
        assert np.abs(d2[l] - wd).max() <= 1e-15
        if np.std(wd) > 0:
            assert np.allclose(w2[l], ww, rtol=1e-9)


def _ctx(capi, symq_dev, nSym, img0):
    c = capi.PfCtx()
    c.symQuat, c.nSym, c.img0 = (symq_dev.data_ptr() if nSym else None), nSym, img0
    return c


@pytest.mark.parametrize("group", ["C4", "D2", "O"])
def test_filter_with_point_group(oracle, dev, group):
    """Particle::symmetrise / perturb / calVari / resample with a point group on the device (thx_pf_symmetrise_dev,
    thx_pf_perturb_ex_dev, thx_pf_update_ex_dev, thx_pf_cal_vari_dev with thx_pf_ctx) against the oracle's restatement of
    src/Particle.cpp:1030-1036,1234,2445-2470 and src/Geometry/Symmetry.cpp:309-336, draws replayed; img0 shifts the images'
    Philox numbering"""
    from thunder_amd import capi, synth
    O = oracle
    rng = np.random.default_rng(31)
    sym = O.symmetry(group)
    symq = T(sym["quat"], dev)
    nImg, nR, nT, seed, call, img0 = 6, 125, 9, 0xABCDEF0123, 7, 1000
    q, t, wR, wT, k, s = _state(rng, nImg, nR, nT)
    # scatter two of the clouds over symmetry-equivalent poses (what a global scan's resampling hands over)
    conj = np.concatenate([[[1.0, 0, 0, 0]], sym["quat"] * np.array([1.0, -1, -1, -1])])
    for l in (1, 4):
        pick = rng.integers(0, len(conj), nR)
        q[l] = np.stack([synth.quat_mul(conj[pick[i]][None], q[l, i][None])[0] for i in range(nR)])
    # ---- symmetrise: bit-exact, with ANCHOR_POINT_2 and with a per-image anchor ----
    anchor = synth.random_quats(nImg, rng)
    for an in (None, anchor):
        d = T(q, dev)
        capi.call("thx_pf_symmetrise_dev", d.data_ptr(), None if an is None else T(an, dev).data_ptr(), nImg, nR, symq.data_ptr(),
                  sym["n"], capi.stream_ptr())
        got = d.cpu().numpy()
        for l in range(nImg):
            assert np.array_equal(got[l], O.symmetrise(q[l], sym["quat"], None if an is None else an[l]))
    # ---- calVari on its own (Particle::load): anchor draw replayed, r symmetrised in place ----
    ctx = _ctx(capi, symq, sym["n"], img0)
    dq, dt = T(q, dev), T(t, dev)
    dk = torch.zeros((nImg, 3), dtype=torch.float64, device=dev)
    ds = torch.zeros((nImg, 2), dtype=torch.float64, device=dev)
    capi.call("thx_pf_cal_vari_dev", dq.data_ptr(), dt.data_ptr(), dk.data_ptr(), ds.data_ptr(), nImg, nR, nT, seed, call, C.byref(ctx),
              capi.stream_ptr())
    gq, gk, gs = dq.cpu().numpy(), dk.cpu().numpy(), ds.cpu().numpy()
    for l in range(nImg):
        iA = min(int(PH.draw_u4(seed, img0 + l, call, 13, 0)[0] * nR), nR - 1)
        kw, sw, qf = O.cal_vari(q[l], t[l], symQuat=sym["quat"], iAnchor=iA, return_q=True)
        assert np.abs(gq[l] - qf).max() <= 1e-13            # (the round trip through the mean frame: 1e-16 of noise)
        assert np.allclose(gk[l], kw, rtol=1e-6) and np.allclose(gs[l], sw, rtol=1e-12)
    # ---- perturb: the three passes, then symmetrise(&mean) ----
    q = gq.copy()                                           # continue from folded clouds (tight: a well-defined mean)
    k = np.maximum(gk, 1e-5)
    pfR, pfT, transS, transQ = 2.0, 0.5, 2.0, 0.05
    dq, dt, dwR, dwT, dk, ds = T(q, dev), T(t, dev), T(wR, dev), T(wT, dev), T(k, dev), T(s, dev)
    capi.call("thx_pf_perturb_ex_dev", dq.data_ptr(), dt.data_ptr(), dwR.data_ptr(), dwT.data_ptr(), dk.data_ptr(), ds.data_ptr(), nImg, nR,
              nT, pfR, pfT, transS, transQ, seed, call + 1, None, C.byref(ctx), capi.stream_ptr())
    pq, pt, pwR, pwT = [x.cpu().numpy() for x in (dq, dt, dwR, dwT)]
    for l in range(nImg):
        gR = np.stack(PH.draw_n4(seed, img0 + l, call + 1, 0, np.arange(nR)), axis=1)
        gT = np.stack(PH.draw_n4(seed, img0 + l, call + 1, 1, np.arange(nT)), axis=1)
        wq, wt, wwR, wwT = O.pf_perturb(q[l], t[l], k[l], s[l], pfR, pfT, transS, transQ, gR, gT, symQuat=sym["quat"])
        assert np.abs(pq[l] - wq).max() <= 2e-9 and np.abs(pt[l] - wt).max() <= 1e-12
        assert np.allclose(pwR[l], wwR, rtol=1e-6) and np.allclose(pwT[l], wwT, rtol=1e-9)
    # ---- update: calVari (symmetrise about a drawn anchor) + resample ----
    uR = (rng.uniform(0, 1, (nImg, nR)) ** 8).astype(np.float32)
    uT = rng.uniform(0.05, 1, (nImg, nT)).astype(np.float32)
    topR = torch.zeros((nImg, 4), dtype=torch.float64, device=dev)
    topT = torch.zeros((nImg, 2), dtype=torch.float64, device=dev)
    capi.call("thx_pf_update_ex_dev", dq.data_ptr(), dt.data_ptr(), dwR.data_ptr(), dwT.data_ptr(), T(uR, dev).data_ptr(), T(uT, dev).data_ptr(),
              dk.data_ptr(), ds.data_ptr(), topR.data_ptr(), topT.data_ptr(), nImg, nR, nT, 1e-3, seed, call + 2, None, C.byref(ctx),
              capi.stream_ptr())
    uq, ut, uwR, uk, utop = [x.cpu().numpy() for x in (dq, dt, dwR, dk, topR)]
    for l in range(nImg):
        c2, li = call + 2, img0 + l
        iA = min(int(PH.draw_u4(seed, li, c2, 13, 0)[0] * nR), nR - 1)
        own = O.pf_update(pq[l], pt[l], pwR[l], pwT[l], uR[l], uT[l], 1e-3, PH.shuffle_ranks(seed, li, c2, 2, nR),
                          PH.draw_u4(seed, li, c2, 3, 0)[0] / nR, PH.shuffle_ranks(seed, li, c2, 4, nT),
                          PH.draw_u4(seed, li, c2, 5, 0)[0] / nT, symQuat=sym["quat"], iAnchor=iA)
        assert np.abs(uq[l] - own["q"]).max() <= 1e-13 and np.array_equal(ut[l], own["t"])
        assert np.allclose(uwR[l], own["wR"], rtol=1e-10) and np.allclose(uk[l], own["k"], rtol=1e-6)
        assert np.abs(utop[l] - own["topR"]).max() <= 1e-13
