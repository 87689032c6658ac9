"""The oracle's chain of a whole EM iteration (oracle.Iteration) on the CPU: regression against the committed fixture
tests/golden/iteration_n32.npz, what the chain does as an EM iteration, and the sensitivity of the reference's gridding stop
rule that the GPU chain test (tests/test_iteration_gpu.py) has to live with."""
import os
import sys

import numpy as np

import _iter_util as U

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_iteration_chain_matches_golden_and_refines(oracle):
    import make_golden_iteration as G
    O = oracle
    gold = np.load(os.path.join(HERE, "golden", "iteration_n32.npz"))
    inp, out = G.compute()
    assert bytes(out["input_sha256"]) == bytes(gold["input_sha256"]), "the seeded inputs changed"
    for k in gold.files:
        a, b = out[k], gold[k]
        assert a.shape == b.shape, k
        if a.dtype.kind in "iu":
            assert np.array_equal(a, b), k
        else:   # same machine arithmetic up to libm / FFT-library rounding
            assert np.allclose(a, b, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(b).max()))), k
    # what an EM iteration should do: shifts recovered by the re-centring, half maps agree with the generating map and
    # with each other at low resolution, the Wiener term of iteration 2 came from iteration 1's curve
    N = G.PARAMS["N"]
    assert np.sqrt(((-out["it2_offset"] - inp["shift"]) ** 2).mean()) < 0.5
    for h in (0, 1):
        f = U.fsc_curve(O, out["it2_maps"][h], inp["ref"], N, 6)
        assert np.all(f[1:4] > 0.9) and f[4] > 0.8, f      # (120 particles at 32^3: seed to seed the fifth shell moves between 0.84 and 0.97)
    assert out["it1_fsc"][1] > 0.99 and out["it1_fsc"][-1] < 0.5
    assert np.all((out["it1_rounds"] > 10) & (out["it1_rounds"] <= 30))


def test_forced_rounds_equal_the_stop_rule_where_it_stops(oracle):
    """oracle.reconstruct(force_rounds = k) runs the same loop: forcing the round count the rule chose reproduces the map"""
    O = oracle
    F, T, N = _thin_accumulators(O, 32, 40, seed=5)
    P, rU = 2 * N, N // 2 - 2
    O.normalise_TF(F, T, P)
    m0, it0, d0, _ = O.reconstruct(F, T, P, N, 2, rU, MAP=False, gridCorr=True, return_iters=True)
    m1, it1, d1, _ = O.reconstruct(F, T, P, N, 2, rU, MAP=False, gridCorr=True, return_iters=True, force_rounds=it0)
    assert it1 == it0 and np.array_equal(m0, m1) and d0 == d1
    m2, it2, _, _ = O.reconstruct(F, T, P, N, 2, rU, MAP=False, gridCorr=True, return_iters=True, force_rounds=it0 + 3)
    assert it2 == it0 + 3 and not np.array_equal(m0, m2)


def _thin_accumulators(O, N, nImg, seed, draws=8):
    """F / T of a small data set: nImg noisy images x `draws` rotations of a 0.03 rad cloud each"""
    from thunder_amd import synth
    rng = np.random.default_rng(seed)
    P, rU = 2 * N, N // 2 - 2
    vol = O.set_projectee(synth.blob_map(N, seed=9, nblob=14), 2)
    pl = O.pixel_list(N, rU, 0, 2)
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    T = np.zeros((P, P, P // 2 + 1), np.float32)
    quat, attr = synth.random_quats(nImg, rng), synth.ctf_params(nImg, rng)
    for l in range(nImg):
        c = O.ctf(1.32, *attr[l], N, pl["iCol"], pl["iRow"])
        s = O.project(vol, P, 2, O.rotate3D(quat[l]), pl["iCol"], pl["iRow"]) * c
        s = (s + (rng.standard_normal(len(s)) + 1j * rng.standard_normal(len(s))) * np.abs(s).std() * 2).astype(np.complex64)
        for q in synth.perturb_quats(quat[l:l + 1], draws, 0.03, rng)[0]:
            O.insertP(F, T, P, s, c, O.rotate3D(q), np.float32(1.0 / draws), pl["iColPad"], pl["iRowPad"])
    return F, T, N


def test_stop_rule_is_noise_sensitive(oracle):
    """Reconstructor::reconstruct ends its balancing loop when max | |C| - 1 | over the sphere has not dropped by 5 % twice
    (src/Reconstructor.cpp:1530-1551).  On a few hundred images that norm sits on barely covered rim voxels and jumps
    around; relative noise of 1e-6 on F / T -- a few ulp of RFLOAT, what the reference's unordered `omp atomic` adds give run
    to run -- is enough to make the rule fire in another round, and the map then differs by per cent of its maximum although
    the inputs agree to 1e-6.  This is why the chain-level GPU test compares maps after the SAME round (force_rounds) and
    why a round count is reported, not required, there."""
    O = oracle
    F, T, N = _thin_accumulators(O, 64, 100, seed=3)
    P, rU = 2 * N, N // 2 - 2

    def reco(F_, T_):
        F_, T_ = F_.copy(), T_.copy()
        O.normalise_TF(F_, T_, P)
        m, it, d, _ = O.reconstruct(F_, T_, P, N, 2, rU, MAP=False, gridCorr=True, return_iters=True)
        return m, it
    m0, it0 = reco(F, T)
    rng = np.random.default_rng(11)
    rounds, moved = [], []
    for _ in range(8):
        g = (1 + 1e-6 * rng.standard_normal(T.shape)).astype(np.float32)   # F and T of a voxel move together
        m1, it1 = reco(F * g, T * g)
        rounds.append(it1)
        moved.append(float(np.abs(m1 - m0).max() / np.abs(m0).max()))
    same = [m for r, m in zip(rounds, moved) if r == it0]
    other = [m for r, m in zip(rounds, moved) if r != it0]
    print("rounds", it0, rounds, "map moved by", ["%.1e" % m for m in moved])
    assert all(m < 1e-3 for m in same)        # same round: the map follows its inputs
    assert other and min(other) > 1e-2        # another round: per cent of the maximum
    m1, it1 = reco(F, T)
    assert it1 == it0 and np.array_equal(m0, m1)   # (the oracle itself is deterministic)


# ---- the chain beyond the one-class local search: point groups, K classes with a global search, normCorrection ----
def _rot_map_z90(m):
    """the map rotated by +90 degrees about z (wrapped-index layout [z][y][x]): m'(x, y, z) = m(y, -x, z)"""
    n = m.shape[0]
    return np.transpose(m, (0, 2, 1))[:, (-np.arange(n)) % n, :]


def test_chain_with_point_group(oracle):
    """C4 through the chain (Particle::symmetrise in perturb / calVari, prepareTF's symmetrizeT / symmetrizeF): every image's
    support points stay next to each other although the group scatters equivalent poses, F / T after prepareTF are invariant
    under the group, T is NOT divided by the group order, and the maps carry the symmetry"""
    O = oracle
    N, n = 16, 48
    inp = U.make_inputs(O, N, n, seed=411, mLR=24, mLT=5, nPhase=2, mReco=8, batch=16, snr=2.0, rL=1, sym="C4")
    sym = inp["cfg"]["sym"]
    # start half of the clouds from symmetry-equivalent poses of their own support points
    rng = np.random.default_rng(3)
    from thunder_amd import synth
    conj = np.concatenate([[[1.0, 0, 0, 0]], sym["quat"] * np.array([1.0, -1, -1, -1])])
    for l in range(0, n, 2):
        pick = rng.integers(0, len(conj), inp["quat0"].shape[1])
        inp["quat0"][l] = np.stack([synth.quat_mul(conj[pick[i]][None], inp["quat0"][l, i][None])[0] for i in range(len(pick))])
    it = U.oracle_chain(O, inp)
    # Particle::load -> calVari folds the clouds: the spread is that of a tight cloud again
    assert it.k.max() < 0.05
    r = it.iterate()
    assert np.all(np.isfinite(it.k)) and it.k.max() < 0.5      # calVari of every phase saw folded clouds
    P, rU = 2 * N, N // 2 - 2
    for h in (0, 1):
        Fs, Ts, Fr, Tr = r["F_sym"][h][0], r["T_sym"][h][0], r["F_raw"][h][0], r["T_raw"][h][0]
        # the sum over the group of the normalised accumulators: the origin voxel is hit by every element
        assert np.isclose(Ts[0, 0, 0], (1 + sym["n"]) * 1.0, rtol=1e-5)
        m = r["maps"][h][0]
        assert np.abs(_rot_map_z90(m) - m).max() <= 0.12 * np.abs(m).max()
        f = U.fsc_curve(O, m, inp["ref"], N, 5)
        assert np.all(f[1:4] > 0.85), f
    # the same particles through a C1 chain: no such invariance
    inp1 = dict(inp, cfg=dict(inp["cfg"], sym=None))
    r1 = U.oracle_chain(O, inp1).iterate()
    m1 = r1["maps"][0][0]
    assert np.abs(_rot_map_z90(m1) - m1).max() > np.abs(_rot_map_z90(r["maps"][0][0]) - r["maps"][0][0]).max()


def test_chain_global_search_classifies(oracle):
    """K = 3 references, global search: the scan assigns the classes, the local phases run with phase index 1.. (no large
    perturbation), every image's draws go to the F / T of its class, the halves of a class are averaged everywhere, no
    re-centring after a global search; an empty class takes over another class's reference (balanceClass)"""
    O = oracle
    N, n, K = 16, 60, 3
    inp = U.make_inputs(O, N, n, seed=97, mLR=12, mLT=4, nPhase=2, mReco=6, batch=32, snr=20.0, rL=1, K=K,
                        scan=dict(nR=60, nT=4, rScan=5), balance=1)
    # nobody belongs to class 2: its particles are redrawn from classes 0 / 1
    it = U.oracle_chain(O, inp)
    r = it.iterate(search="global")
    assert (r["cls"] == inp["cls_true"]).mean() >= 0.9
    assert np.all(r["offset"] == 0) and np.array_equal(it.img, it._remask(it.imgOri))        # no reCentreImg / reMaskImg
    assert r["avgR"] == -1
    for k in range(K):
        assert np.array_equal(r["maps"][0][k], r["maps"][1][k]) or not np.any(r["cls"] == k)  # A = B = (A + B) / 2
        for h in (0, 1):
            sel = np.nonzero(r["cls"][slice(*it.ranges[h])] == k)[0]
            # T(0,0,0) counts the draws that went to the class: every image adds mReco x w x ctf(0)^2 = amplitudeContrast^2 = 0.01
            assert np.isclose(r["T_raw"][h][k][0, 0, 0], 0.01 * len(sel), rtol=1e-4, atol=1e-9)
    # class maps resemble their own reference
    for k in range(K):
        own = [U.fsc_curve(O, r["maps"][0][k], inp["refs"][j], N, 4)[1:4].mean() for j in range(K)]
        assert int(np.argmax(own)) == k, (k, own)
    # second iteration: local search in the assigned classes, with re-centring
    r2 = it.iterate(search="local")
    assert np.array_equal(r2["cls"], r["cls"]) and np.abs(r2["offset"]).max() > 0
    # balanceClass: a data set in which class 2 is empty
    inp2 = U.make_inputs(O, N, n, seed=98, mLR=12, mLT=4, nPhase=1, mReco=6, batch=32, snr=20.0, rL=1, K=K,
                         scan=dict(nR=60, nT=4, rScan=5), balance=1)
    it2 = U.oracle_chain(O, inp2)
    it2.ref[2] = 0.0                                    # a reference nothing matches
    it2.reset()
    rb = it2.iterate(search="global")
    if not np.any(rb["cls"] == 2):
        assert rb["bm"][2] in (0, 1) and rb["bm"][0] == -1 and rb["bm"][1] == -1
        for h in (0, 1):
            assert np.array_equal(rb["mapsFsc"][h][2], rb["mapsFsc"][h][rb["bm"][2]])
            assert np.array_equal(rb["maps"][h][2], rb["maps"][h][rb["bm"][2]])


def test_chain_norm_correction(oracle):
    """Optimiser::normCorrection in the chain (src/Optimiser.cpp:3405-3413,6201-6394): not in the first iteration; from the second on
    every image is rescaled by sqrt(median / norm) with norm = its residual power against the top pose's slice inside
    rNorm = min(r, resolutionP(0.75) of the previous FSC), both stacks by the same factor, the M-step rows cut from the rescaled
    stack; an image four times too strong is turned down"""
    O = oracle
    N, n = 16, 200
    inp = U.make_inputs(O, N, n, seed=5, mLR=16, mLT=4, nPhase=2, mReco=6, batch=32, snr=1.0, rL=1, norm_correction=1)
    inp["imgOri"][7] *= 4.0
    it = U.oracle_chain(O, inp)
    ori0 = it.imgOri.copy()
    r1 = it.iterate()
    assert "norm" not in r1 and np.array_equal(it.imgOri, ori0)
    assert O.res_p(r1["fsc"][0], 0.75, 1, 1, False) >= 2
    r2 = it.iterate()
    assert "norm" in r2 and r2["rNorm"] == min(N // 2 - 2, O.res_p(r1["fsc"][0], 0.75, 1, 1, False))
    scale = np.sqrt(np.float32(r2["normMedian"]) / r2["norm"])
    assert np.allclose(it.imgOri, ori0 * scale[:, None, None], rtol=2e-6, atol=0)
    assert r2["normMedian"] == O.median(r2["norm"]) and scale[7] < 1
    # the insertion saw the rescaled rows: T is unchanged by the scale (CTF^2 weights), F scales with the images
    assert np.all(np.isfinite(r2["F_raw"][0][0])) and np.all(np.isfinite(r2["maps"][0][0]))


def test_chain_per_image_stop_rule(oracle):
    """the per-image stop rule in the chain (src/Optimiser.cpp:1183,1510-1615; maxPhase > nPhase): every image runs phases 0 ..
    nPhase unconditionally -- the rule is first asked AFTER the phase with index nPhase = MIN_N_PHASE_PER_ITER_LOCAL, so an image
    runs at least nPhase + 1 phases --, ends in the first phase in which none of its variances fell by 5 % (squared for k1..k3)
    below the smallest seen so far, and at maxPhase at the latest; images stop in different phases"""
    O = oracle
    N, n, nPhase, maxPhase = 16, 60, 2, 7
    inp = U.make_inputs(O, N, n, seed=11, mLR=16, mLT=4, nPhase=nPhase, mReco=6, batch=32, snr=1.0, rL=1, max_phase=maxPhase)
    it = U.oracle_chain(O, inp)
    r = it.iterate()
    ran, nP = r["phases"], r["nP"]
    assert ran.min() >= nPhase + 1 and ran.max() <= maxPhase and len(np.unique(ran)) > 1
    stopped = nP > 0                                      # (phase INDEX in which the rule ended the search; 0: it never did)
    assert np.all(nP[stopped] == ran[stopped] - 1) and np.all(ran[~stopped] == maxPhase) and stopped.any() and (~stopped).any()
    # the decision, restated: k / s of the phases an image ran, against the running minima from (1, 1, 1, 5 transS, 5 transS)
    for l in range(n):
        st = O.stop_rule_init(inp["cfg"]["transS"])
        for p in range(int(ran[l])):
            if p >= nPhase:
                end = O.stop_rule(st, *r["k"][p, l], *r["s"][p, l])
                assert end == (p == ran[l] - 1 and stopped[l])
    # rows of phases an image did not run stay empty
    for l in range(n):
        assert not np.any(r["uR"][ran[l]:, l]) and np.all(r["uR"][:ran[l], l].sum(axis=1) > 0)


def test_reconstruct_on_a_resized_grid(oracle):
    """Reconstructor::resizeSpace (src/Reconstructor.cpp:184-198; Model::resetReco src/Model.cpp:1113): the same slices inserted into
    the (pf size)^3 grid and into the (pf N)^3 grid, reconstructed on each.  The two are not the same computation -- convoluteC keeps
    QUAD_3 / (N pf)^2 as the kernel's argument on the smaller grid (:2639-2645), so the balancing kernel differs -- but both are
    reconstructions of the same data: they agree with each other and with the generating map inside the cut-off, and the small
    grid's map has nothing beyond it that the large one lacks.  (Pins the placement of F W into the (N pf)^3 padDst, :1677-1701.)"""
    from thunder_amd import synth
    O = oracle
    N, pf, rU = 32, 2, 8
    rng = np.random.default_rng(3)
    ref = synth.blob_map(N, nblob=10)
    vol = O.set_projectee(ref, pf)
    size = min(N, (rU + 2) * 2)
    PF, PN = pf * size, pf * N
    assert (size, PF) == (20, 40)
    pl = O.pixel_list(N, rU, 0, pf)
    quat = synth.random_quats(300, rng)
    maps = {}
    for P in (PF, PN):
        F, T = np.zeros((P, P, P // 2 + 1), np.complex64), np.zeros((P, P, P // 2 + 1), np.float32)
        for q in quat:
            R = O.rotate3D(q)
            O.insertP(F, T, P, O.project(vol, PN, pf, R, pl["iCol"], pl["iRow"]), np.ones(pl["nPxl"], np.float32), R, 1.0, pl["iColPad"], pl["iRowPad"])
        O.normalise_TF(F, T, P)
        maps[P], it, _, _ = O.reconstruct(F, T, P, N, pf, rU, MAP=False, gridCorr=True, return_iters=True)
        assert 10 < it <= 30
    f = U.fsc_curve(O, maps[PF], maps[PN], N, rU)
    assert f.min() >= 0.995, f
    fr = U.fsc_curve(O, maps[PF], ref, N, rU)
    assert fr[:rU - 1].min() >= 0.99, fr
    assert np.abs(maps[PF] - maps[PN]).max() <= 3e-2 * np.abs(maps[PN]).max()


def test_iteration_chain_with_cutoffs(oracle):
    """oracle.Iteration.set_cutoff(r, rU): the lists, the grid of F / T and the length of the FSC follow the cut-offs of every iteration;
    the MAP reconstruction of an iteration uses the previous iteration's curve with the previous rU entries (Model::resetReco,
    src/Model.cpp:1122); cut-offs at Nyquist are the chain without the call, bit for bit."""
    O = oracle
    N, n = 32, 60
    inp = U.make_inputs(O, N, n, seed=21, mLR=30, mLT=4, nPhase=2, mReco=6, snr=2.0)
    a, b = U.oracle_chain(O, inp), U.oracle_chain(O, inp)
    b.set_cutoff(N // 2 - 2, N // 2 - 2)
    oa, ob = a.iterate(), b.iterate()
    assert np.array_equal(oa["fsc"], ob["fsc"]) and np.array_equal(oa["maps"][0][0], ob["maps"][0][0])
    it = U.oracle_chain(O, inp)
    it.set_cutoff(7, 8)
    assert (it.rE, it.rU, it.size, it.PF) == (7, 8, 20, 40) and it.fscReco.shape == (1, 8)
    assert (it.pl["nPxl"], it.plM["nPxl"]) == (O.pixel_list(N, 7, 2)["nPxl"], O.pixel_list(N, 8, 0)["nPxl"])
    o1 = it.iterate()
    assert o1["fsc"].shape == (1, 8) and o1["F"][0][0].shape == (40, 40, 21) and it.fscReco.shape == (1, 8)
    it.set_cutoff(10, 12)
    assert it.fscReco.shape == (1, 8) and it.PF == 56        # the reconstructor keeps the curve it was handed; the grid follows rU
    o2 = it.iterate()
    assert o2["fsc"].shape == (1, 12) and o2["T"][1][0].shape == (56, 56, 29)
    for h in (0, 1):
        f = U.fsc_curve(O, o2["maps"][h][0], inp["ref"], N, 6)
        assert np.all(f[1:4] > 0.85), f
    # nothing of the data beyond the cut-off reached the accumulators
    k = np.fft.fftfreq(56, 1.0 / 56)
    kk = np.sqrt(k[:, None, None] ** 2 + k[None, :, None] ** 2 + np.arange(29)[None, None, :] ** 2)
    assert not np.any(o2["T_raw"][0][0][kk > 2 * 12 + 2])
