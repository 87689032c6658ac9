"""CPU checks of the particle-filter restatements in the oracle (src/Particle.cpp, src/Geometry/DirectionalStat.cpp) and of
the Philox replica the GPU tests replay the device's draws with."""
import ctypes as C

import numpy as np

import _philox as PH


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors: zero, all-ones and pi-digits inputs)"""
    z = PH.philox(0, 0, 0, 0, 0)
    assert [int(x) for x in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xFFFFFFFF
    o = PH.philox((f << 32) | f, f, f, f, f)
    assert [int(x) for x in o] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    p = PH.philox((0x299f31d0 << 32) | 0xa4093822, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)
    assert [int(x) for x in p] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = PH.draw_u4(123, np.arange(1000), 0, 0, 0)
    assert all(0 < x.min() and x.max() < 1 for x in u) and abs(np.mean(u[0]) - 0.5) < 0.05
    n = np.concatenate(PH.draw_n4(7, np.arange(5000), 1, 2, 3))
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1) < 0.03
    r = PH.shuffle_ranks(9, 4, 2, 2, 125)
    assert sorted(r.tolist()) == list(range(125))


def test_infer_acg_fixed_point_and_moments(oracle):
    O = oracle
    O.lib().orc_infer_acg.restype = C.c_int
    O.lib().orc_pdf_acg.restype = C.c_double
    rng = np.random.default_rng(2)
    n = 4000
    sig = np.diag([1.0, 0.02, 0.008, 0.002])
    x = rng.multivariate_normal(np.zeros(4), sig, n)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    A = np.zeros(16)
    rounds = O.lib().orc_infer_acg(_dp(A), _dp(np.ascontiguousarray(x)), n)
    A = A.reshape(4, 4)
    assert rounds > 2 and np.allclose(A, A.T, atol=1e-12)
    # Tyler's fixed point: A ~ 4 sum(x x^T / u) / sum(1 / u), u = x^T A^-1 x (one more round moves it by < 1e-3)
    u = np.einsum("ni,ij,nj->n", x, np.linalg.inv(A), x)
    B = 4 * (x[:, :, None] * x[:, None, :] / u[:, None, None]).sum(0) / (1 / u).sum()
    assert np.abs(A - B).sum() <= 1e-3
    # shape recovered up to scale
    assert np.allclose(np.diag(A)[1:] / A[0, 0], np.diag(sig)[1:], rtol=0.15)
    v = np.zeros(4)
    O.lib().orc_sym4_top_eigvec(_dp(v), _dp(np.ascontiguousarray(A)))
    w, V = np.linalg.eigh(A)
    assert min(np.abs(v - V[:, -1]).max(), np.abs(v + V[:, -1]).max()) < 1e-9
    p = O.lib().orc_pdf_acg(_dp(np.ascontiguousarray(x[0])), _dp(np.ascontiguousarray(A)))
    assert np.isclose(p, np.linalg.det(A) ** -0.5 * (x[0] @ np.linalg.inv(A) @ x[0]) ** -2, rtol=1e-10)


def test_cal_vari_and_weights(oracle):
    O = oracle
    from thunder_amd import synth
    rng = np.random.default_rng(3)
    n = 125
    mean = synth.random_quats(1, rng)[0]
    d = rng.standard_normal((n, 4)) * np.array([1, 0.05, 0.03, 0.01])
    d[:, 0] = 1
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # calVari looks at conj(mean) * r (LEFT multiplication, src/Particle.cpp:1052-1058): a cloud mean * d shows d's axes
    q = np.ascontiguousarray(synth.quat_mul(np.broadcast_to(mean, d.shape), d))
    q0 = q.copy()
    k, m = np.zeros(3), np.zeros(4)
    O.lib().orc_cal_vari_R(_dp(k), _dp(m), _dp(q), n)
    assert np.abs(q - q0).max() < 1e-14                        # rotated to the mean frame and back
    assert min(np.abs(m - mean).max(), np.abs(m + mean).max()) < 0.02
    # (the cloud is Gaussian in the tangent plane, not ACG: Tyler's estimator sees ~0.5-0.7 of sigma^2)
    assert np.all(k < [0.05 ** 2, 0.03 ** 2, 0.01 ** 2]) and np.all(k > 0.3 * np.array([0.05 ** 2, 0.03 ** 2, 0.01 ** 2]))
    assert k[0] > k[1] > k[2]
    t = rng.normal(0, [1.5, 0.7], size=(9, 2))
    s = np.zeros(2)
    O.lib().orc_cal_vari_T(_dp(s), _dp(np.ascontiguousarray(t)), 9)
    assert np.allclose(s, t.std(0, ddof=1), rtol=1e-12)
    w = np.zeros(9)
    O.lib().orc_balance_weight_T(_dp(w), _dp(np.ascontiguousarray(t)), 9)
    z = (t - t.mean(0)) / t.std(0, ddof=1)
    want = np.exp((z ** 2).sum(1) / 2)
    assert np.allclose(w, want / want.sum(), rtol=1e-10)
    wr = np.zeros(n)
    O.lib().orc_balance_weight_R(_dp(wr), _dp(q0), n)
    assert abs(wr.sum() - 1) < 1e-12 and np.all(wr > 0)
    # points far from the mode get the larger weight (1 / pdf)
    dist = 1 - np.abs(q0 @ m)
    assert np.corrcoef(dist, wr)[0, 1] > 0.5


def test_resample_and_peak(oracle):
    O = oracle
    rng = np.random.default_rng(4)
    n = 50
    u = rng.uniform(0, 1, n) ** 4
    u2 = u.copy()
    O.lib().orc_keep_half_height_peak(_dp(u2), n, C.c_double(0.2))
    hh = 0.2 * u.max()
    assert np.allclose(u2, np.where(u < hh, 0, u - hh))
    w = np.full(n, 1.0 / n)
    idx = np.zeros(n, np.int32)
    wo = np.zeros(n)
    O.lib().orc_resample(idx.ctypes.data_as(C.POINTER(C.c_int)), _dp(wo), _dp(w), _dp(u2), n, n, C.c_double(0.5 / n))
    assert np.all(np.diff(idx) >= 0) and np.all(u2[idx] > 0)
    cnt = np.bincount(idx, minlength=n)
    expect = n * u2 / u2.sum()
    assert np.all(np.abs(cnt - expect) <= 1.0 + 1e-9)          # systematic resampling: counts within 1 of n * p
    assert np.allclose(wo, (1 / u2[idx]) / (1 / u2[idx]).sum())


def test_classification_stages_match_committed_fixture():
    """tests/golden/classify_n16.npz against a fresh run of its generator: the oracle's scan weights over K classes, the class of
    every image and the support points of the local search (replayed Philox draws) have not moved"""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden_classify as G
    from oracle import oracle as O
    g = np.load(os.path.join(here, "golden", "classify_n16.npz"))
    out = G.compute(O)
    assert set(out) == set(g.files)
    for k in g.files:
        if k in ("wC", "wR", "wT", "base", "k123"):
            np.testing.assert_allclose(out[k], g[k], rtol=1e-6, atol=1e-30, err_msg=k)   # (expf / the compiler's reassociation-free sums: same binary, same bits here)
        else:
            assert np.array_equal(out[k], g[k]), k
    assert (g["cls"] == g["cls_true"]).all()


def test_classify_stages_properties():
    """oracle.ClassifyStages on a small seeded case (numpy + oracle only): the scan recovers class, rotation and shift of images that
    sit on the scanned grid; the class weights are the per-class maxima relative to the best class (one of them is 1); the support
    points are grid points carrying scan weight, their priors sum to 1, the top point is the scan's best; the minimum spread of the
    scanning phase bounds k1..k3 and s0, s1 from below; another call id draws other support points from the same weights"""
    import _classify_util as U
    from oracle import oracle as O
    N, K, nImg, nR, nT, rScan, rL, mLR, mLT, seed = 16, 2, 10, 30, 3, 5, 1, 6, 3, 4711
    cs = U.make_case(O, N, K, nImg, nR, nT, 123, noise=0.2)
    st = U.stages(O, N, K, cs["vols"], cs["quat"], cs["shifts"], rScan, rL)
    s2m = U.sub_rows(cs["plM"], st.plS)
    wC, wR, wT, base = st.scan(cs["datM"][:, s2m], cs["ctfM"][:, s2m], cs["sigM"][:, s2m])
    assert np.all(np.isfinite(base)) and np.all(wC >= 0) and np.all(wR >= 0) and np.all(wT >= 0)
    cls = st.classes(wC, seed, 1.0 - 1e-2)
    assert np.array_equal(cls, cs["cls_true"]) and np.array_equal(cls, wC.argmax(1))
    assert np.array_equal(wR[cls, np.arange(nImg)].argmax(1), cs["r_true"])
    minK, minS = 0.4, 0.3
    for l in range(nImg):
        ws, rankR = st.support(wR, wT, cls, l, seed, 1e-3, mLR, mLT, minK, minS)
        assert sorted(rankR) == list(range(nR))
        assert np.all(wR[cls[l], l][ws["srcR"]] > 0) and np.all(wT[cls[l], l][ws["srcT"]] > 0)
        assert abs(ws["wR"].sum() - 1) < 1e-12 and abs(ws["wT"].sum() - 1) < 1e-12
        assert np.array_equal(ws["topR"], cs["quat"][cs["r_true"][l]])
        assert np.all(ws["k"] >= minK) and np.all(ws["s"] >= minS)
        assert np.array_equal(ws["t"], cs["shifts"][ws["srcT"]])
    a, _ = st.support(wR, wT, cls, 0, seed, 1e-3, mLR, mLT, 0.0, 0.0, call=2)
    b, _ = st.support(wR, wT, cls, 0, seed, 1e-3, mLR, mLT, 0.0, 0.0, call=3)
    assert np.array_equal(a["topR"], b["topR"])       # the top point does not depend on the draws


# ---- point-group symmetry (Symmetry::init, symmetryCounterpart, Particle::symmetrise) ----
GROUPS = [("C1", 1), ("C2", 2), ("C4", 4), ("C7", 7), ("D2", 4), ("D3", 6), ("D7", 14), ("T", 12), ("O", 24), ("I1", 60), ("I2", 60),
          ("I3", 60), ("I4", 60)]


def _quat_to_R(q):
    from oracle import oracle as O
    return O.rotate3D(q).reshape(3, 3).T


def test_symmetry_groups_are_groups(oracle):
    """the restated Symmetry::init: order of every point group, closure under products (completePointGroup), the quaternion of
    every element is the quaternion of its matrix, and Cn is what Symmetry::fillLR gives for RotationSO(n, 0, 0, 1) with its
    RFLOAT angle (src/Geometry/Symmetry.cpp:146-171)"""
    O = oracle
    for name, order in GROUPS:
        s = O.symmetry(name)
        assert s["n"] == order - 1, name
        R = [np.eye(3)] + [m.reshape(3, 3).T for m in s["R"]]
        for a in R:
            assert np.abs(a @ a.T - np.eye(3)).max() < 2e-5 and abs(np.linalg.det(a) - 1) < 2e-5
            for b in R:
                c = a @ b
                assert min(np.abs(c - x).max() for x in R) < 1e-4, name      # closed (axes carry 7 digits)
        for m, q in zip(s["R"], s["quat"]):
            assert abs(np.linalg.norm(q) - 1) < 1e-5
            # quaternion(dvec4&, const dmat33&) (src/Geometry/Euler.cpp:112-123) takes w = sqrt(1 + trace) / 2 and the SIGNS of x, y, z
            # from differences of off-diagonal elements: for the two-fold elements of T / O / I, whose axes are given to 6 - 7
            # digits, w is sqrt(5e-6) = 1e-3 instead of 0 and the signs come from rounding noise, so Symmetry::quat(i) can be the
            # quaternion of ANOTHER two-fold element of the group.  The reference's own arithmetic, kept: what matters to
            # symmetryCounterpart is that every quat is (close to) an element of the group.
            Rq = _quat_to_R(q / np.linalg.norm(q))
            if name[0] in "CD":
                assert np.abs(Rq - m.reshape(3, 3).T).max() < 2e-5
            else:
                assert min(np.abs(Rq - x).max() for x in R) < 5e-3, name
    c4 = O.symmetry("C4")
    ang = np.float32(2 * np.pi / 4)
    for j in (1, 2, 3):
        a = float(np.float32(ang * np.float32(j)))
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        assert np.abs(c4["R"][j - 1].reshape(3, 3).T - Rz).max() < 1e-15
    try:
        O.symmetry("X9")
        assert False
    except ValueError:
        pass


def test_symmetry_host_twin_is_identical(oracle):
    """thx_symmetry_host (the product's Symmetry::init, host code of libthunder_amd.so) == the oracle's, bit for bit"""
    from thunder_amd import capi
    capi.load()
    O = oracle
    for name, order in GROUPS:
        n = C.c_int(0)
        capi.call("thx_symmetry_host", name.encode(), None, None, 0, C.byref(n))
        assert n.value == order - 1
        R, q = np.zeros((max(n.value, 1), 9)), np.zeros((max(n.value, 1), 4))
        capi.call("thx_symmetry_host", name.encode(), R.ctypes.data, q.ctypes.data, n.value, C.byref(n))
        s = O.symmetry(name)
        assert np.array_equal(R[:n.value], s["R"]) and np.array_equal(q[:n.value], s["quat"]), name
    n = C.c_int(0)
    with np.testing.assert_raises(capi.ThxError):
        capi.call("thx_symmetry_host", b"Q5", None, None, 0, C.byref(n))


def test_symmetry_counterpart_properties(oracle):
    """symmetryCounterpart picks, among q and conj(g) q, the one closest (|<., anchor>|) to the anchor: the result is an
    equivalent pose (its slice of a symmetric volume is the same), no group element brings it closer, and symmetrising twice
    changes nothing"""
    O = oracle
    from thunder_amd import synth
    rng = np.random.default_rng(5)
    for name in ("C4", "D2", "D7", "O"):
        s = O.symmetry(name)
        q = synth.random_quats(200, rng)
        anchor = synth.random_quats(1, rng)[0]
        for an in (None, anchor):
            qs = O.symmetrise(q, s["quat"], an)
            a = np.array([1.0, 0, 0, 0]) if an is None else an
            conj = s["quat"] * np.array([1.0, -1, -1, -1])
            for i in range(len(q)):
                cands = np.concatenate([q[i][None], synth.quat_mul(conj, q[i][None])])
                d = np.abs(cands @ a)
                assert np.abs(cands - qs[i]).max(1).min() < 1e-15                       # one of the candidates
                assert np.float32(abs(qs[i] @ a)) >= np.float32(d.max()) - np.float32(1e-7)
            if name != "O":   # (Symmetry::quat of O is not closed under products -- see test_symmetry_groups_are_groups)
                assert np.array_equal(O.symmetrise(qs, s["quat"], an), qs)
    # C1: untouched
    assert np.array_equal(O.symmetrise(q, np.zeros((0, 4))), q)


def test_perturb_and_cal_vari_with_symmetry(oracle):
    """Particle::perturb / calVari with a point group: the perturbed cloud is the C1 cloud moved to the counterparts next to
    the cloud's mean; calVari's spread of a cloud scattered over symmetry-equivalent poses is the spread of the folded cloud"""
    O = oracle
    from thunder_amd import synth
    rng = np.random.default_rng(9)
    s = O.symmetry("D2")
    base = synth.random_quats(1, rng)
    q = synth.perturb_quats(base, 60, 0.03, rng)[0]
    t = rng.normal(0, 1, (9, 2))
    k0, s0 = O.cal_vari(q, t)
    # scatter the cloud over the group: every point replaced by a random equivalent pose  conj(g) q
    conj = np.concatenate([[[1.0, 0, 0, 0]], s["quat"] * np.array([1.0, -1, -1, -1])])
    pick = rng.integers(0, len(conj), len(q))
    qx = np.stack([synth.quat_mul(conj[pick[i]][None], q[i][None])[0] for i in range(len(q))])
    kx, _ = O.cal_vari(qx, t)                       # C1 arithmetic on the scattered cloud: a huge spread
    assert kx.max() > 50 * k0.max()
    ks, _, qf = O.cal_vari(qx, t, symQuat=s["quat"], iAnchor=3, return_q=True)
    assert np.allclose(ks, k0, rtol=1e-6)
    d = np.abs(np.einsum("ij,ij->i", qf, np.broadcast_to(qf[3], qf.shape)))
    assert d.min() > 0.99                           # folded next to the anchor
    g = rng.standard_normal((len(q), 4))
    gT = rng.standard_normal((9, 4))
    q1, _, w1, _ = O.pf_perturb(q, t, k0, s0, 2.0, 2.0, 2.0, 0.05, g, gT)
    q2, _, w2, _ = O.pf_perturb(q, t, k0, s0, 2.0, 2.0, 2.0, 0.05, g, gT, symQuat=s["quat"])
    # same perturbations, each then replaced by an equivalent pose (mostly itself: the cloud is tight)
    same = np.abs(q1 - q2).max(1) < 1e-15
    assert same.mean() > 0.9
    for i in np.nonzero(~same)[0]:
        cands = synth.quat_mul(conj, q1[i][None])
        assert np.abs(cands - q2[i]).max(1).min() < 1e-14
