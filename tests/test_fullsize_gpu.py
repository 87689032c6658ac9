"""GPU parity tests at the bench's own sizes and settings (BASELINE configs[1] / [4]): the HIP path exactly as `bench.py`
drives it -- brick-sorted insertion with resampled particle-filter draws, cell-packed projector, Morton-ordered pixel
list, particle-filter priors, 125 x 9 support points, occupancy cap 2 -- against the CPU oracle on the same inputs.

These are the regimes the small-box tests do not reach: at P = 512 / 1024 the volume has 35 k / 280 k bricks, a region of
the pixel list touches tens of them per pass and the grid's rim clips samples; the E-step runs the packed gather in
pixel-visit order with non-uniform priors.  Oracle cost: a few seconds per case (C, single thread).
"""
import time

import numpy as np
import pytest

from _util import quat_to_mat

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _filter_draws(rng, synth, quat0, shift0, nImg, mLR, mLT, mReco, spread):
    """what Particle::resample + Particle::rand leave (src/Particle.cpp:2109-2300): a cloud of mLR rotations / mLT shifts
    around the pose, resampled by (peaked) weights -- so support points repeat -- and mReco uniform picks among them"""
    cloudR = synth.perturb_quats(quat0, mLR, spread, rng)                      # [nImg][mLR][4]
    cloudT = shift0[:, None, :] + rng.normal(0, 0.6, size=(nImg, mLT, 2))
    quat = np.empty((nImg, mReco, 4))
    tran = np.empty((nImg, mReco, 2))
    for l in range(nImg):
        wR = rng.exponential(size=mLR) ** 3
        wT = rng.exponential(size=mLT) ** 2
        rsR = rng.choice(mLR, size=mLR, p=wR / wR.sum())                        # resample
        rsT = rng.choice(mLT, size=mLT, p=wT / wT.sum())
        quat[l] = cloudR[l][rsR[rng.integers(0, mLR, mReco)]]                   # rand
        tran[l] = cloudT[l][rsT[rng.integers(0, mLT, mReco)]]
    return quat, tran


def _noisy_rows(O, vol_h, P, N, pl, quat0, shift0, attr, rng, pixelSize=1.32):
    nImg = quat0.shape[0]
    dat = np.zeros((nImg, pl["nPxl"]), np.complex64)
    ctf = np.zeros((nImg, pl["nPxl"]), np.float32)
    for l in range(nImg):
        s = O.project(vol_h, P, 2, O.rotate3D(quat0[l]), pl["iCol"], pl["iRow"])
        ctf[l] = O.ctf(pixelSize, *attr[l], N, pl["iCol"], pl["iRow"])
        sig = s * ctf[l] * O.translate(shift0[l, 0], shift0[l, 1], N, pl["iCol"], pl["iRow"])
        sd = np.sqrt(np.mean(np.abs(sig) ** 2)) * 3.0
        dat[l] = (sig + (rng.normal(size=sig.shape) + 1j * rng.normal(size=sig.shape)) * sd / np.sqrt(2)).astype(np.complex64)
    return dat, ctf


def _oracle_insert(O, P, N, pl, dat, ctf, quat, tran, offS, w):
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    Tt = np.zeros((P, P, P // 2 + 1), np.float32)
    for l in range(len(w)):
        for m in range(quat.shape[1]):
            t = tran[l, m] - offS[l]
            src = O.translate(np.float32(-t[0]), np.float32(-t[1]), N, pl["iCol"], pl["iRow"], src=dat[l])
            O.insertP(F, Tt, P, src, ctf[l], O.rotate3D(quat[l, m]), w[l], pl["iColPad"], pl["iRowPad"])
    return F, Tt


def _insert_vs_oracle(O, dev, N, nImg, mReco, seed, spread=0.01):
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list
    rng = np.random.default_rng(seed)
    P = 2 * N
    pl = pixel_list(N, N // 2 - 2, 0)
    plan = ops.RecoPlan(N, N, 2)
    vol_h = plan.set_projectee(T(synth.blob_map(N, nblob=8), dev)).cpu().numpy()
    plan.close()
    quat0 = synth.random_quats(nImg, rng)
    shift0 = rng.normal(0, 2.0, size=(nImg, 2))
    attr = synth.ctf_params(nImg, rng)
    dat, ctf = _noisy_rows(O, vol_h, P, N, pl, quat0, shift0, attr, rng)
    del vol_h
    quat, tran = _filter_draws(rng, synth, quat0, shift0, nImg, 125, 9, mReco, spread)
    nGroups = [len({q.tobytes() for q in quat[l]}) for l in range(nImg)]
    assert max(nGroups) < mReco, "the draws are meant to repeat support points"
    offS = rng.normal(0, 0.8, size=(nImg, 2))
    w = (rng.uniform(0.5, 1.0, size=nImg) / mReco).astype(np.float32)
    t0 = time.perf_counter()
    Fw, Tw = _oracle_insert(O, P, N, pl, dat, ctf, quat, tran, offS, w)
    t_or = time.perf_counter() - t0
    F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev)).reshape(nImg, mReco, 9)
    ops.insert(F, Tt, P, T(dat, dev), T(ctf, dev), T(w, dev), rot, T(tran, dev), T(pl["iCol"], dev), T(pl["iRow"], dev), 2, N,
               offS=T(offS, dev))
    Fg, Tg = F.cpu().numpy(), Tt.cpu().numpy()
    del F, Tt
    print("insert N=%d: %d images x %d draws (%s groups), oracle %.1f s" % (N, nImg, mReco, nGroups, t_or))
    # bar of SURVEY 8c (7): 1e-5 of the largest accumulated magnitude (atomic order is arbitrary in the reference too)
    eF, eT = np.abs(Fg - Fw).max() / np.abs(Fw).max(), np.abs(Tg - Tw).max() / np.abs(Tw).max()
    assert eF <= 1e-5 and eT <= 1e-5, (eF, eT)
    # no voxel outside the oracle's support is touched; voxels left at zero carry at most a sub-quantum contribution
    assert not np.any((Fw == 0) & (Fg != 0)) and not np.any((Tw == 0) & (Tg != 0))
    assert np.abs(Tw[Tg == 0]).max(initial=0) <= 1e-6 * np.abs(Tw).max()
    # total mass: sum T = sum over in-grid samples of w ctf^2 (weights of a cell sum to 1)
    np.testing.assert_allclose(Tg.sum(dtype=np.float64), Tw.sum(dtype=np.float64), rtol=1e-6)


def test_insert_vs_oracle_n256(oracle, dev):
    """configs[1] box: 3 images x 100 resampled draws (~1 degree spread, repeated support points, non-zero offS)"""
    _insert_vs_oracle(oracle, dev, 256, 3, 100, seed=2560, spread=0.01)


def test_insert_vs_oracle_n256_wide_cloud(oracle, dev):
    """a cloud with a 4-degree spread: draws leaving the first draw's plane by tens of voxels"""
    _insert_vs_oracle(oracle, dev, 256, 2, 40, seed=2561, spread=0.04)


def test_insert_vs_oracle_n512(oracle, dev):
    """configs[4] box (P = 1024): one image x 20 draws"""
    _insert_vs_oracle(oracle, dev, 512, 1, 20, seed=5120, spread=0.008)


@pytest.mark.parametrize("N", [32, 64])
def test_insert_far_posterior_modes(oracle, dev, N):
    """draws of ONE image from far-apart posterior modes: exactly / nearly 90 degrees from the first draw, 30-60 degrees away,
    and the first draw itself (the case that broke the per-image window kernel of rounds 1-2: unbounded shear slopes).  Must
    equal the oracle, in bounded time -- the brick-sorted form has no reference plane, every draw is binned the same way."""
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list
    O = oracle
    rng = np.random.default_rng(900 + N)
    P = 2 * N
    pl = pixel_list(N, N // 2 - 2, 0)
    nImg = 3
    q0 = synth.random_quats(nImg, rng)
    q0[0] = [1.0, 0, 0, 0]                       # identity: normal = z exactly, shear axis z

    def qmul(a, b):
        return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])

    def axis_angle(ax, deg):
        h = np.deg2rad(deg) / 2
        return np.concatenate([[np.cos(h)], np.sin(h) * np.asarray(ax, float) / np.linalg.norm(ax)])
    turns = [axis_angle([1, 0, 0], 90.0), axis_angle([0, 1, 0], 90.0), axis_angle([1, 1, 0], 89.9), axis_angle([1, 0, 0], 75.0),
             axis_angle([0, 1, 0], 45.0), axis_angle([1, 2, 3], 30.0), axis_angle([0, 0, 1], 90.0), axis_angle([1, 0, 0], 1.0)]
    mReco = len(turns) + 2
    quat = np.empty((nImg, mReco, 4))
    for l in range(nImg):
        quat[l, 0] = q0[l]
        for m, t in enumerate(turns):
            quat[l, 1 + m] = qmul(t, q0[l])
        quat[l, -1] = quat[l, 1]               # a repeated far draw (grouped)
    tran = rng.normal(0, 1.0, size=(nImg, mReco, 2))
    tran[:, -1] = tran[:, 1]
    dat = (rng.normal(size=(nImg, pl["nPxl"])) + 1j * rng.normal(size=(nImg, pl["nPxl"]))).astype(np.complex64)
    ctf = rng.uniform(-1, 1, size=(nImg, pl["nPxl"])).astype(np.float32)
    offS = rng.normal(0, 0.3, size=(nImg, 2))
    w = (rng.uniform(0.5, 1.0, size=nImg) / mReco).astype(np.float32)
    Fw, Tw = _oracle_insert(O, P, N, pl, dat, ctf, quat, tran, offS, w)
    F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev)).reshape(nImg, mReco, 9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.insert(F, Tt, P, T(dat, dev), T(ctf, dev), T(w, dev), rot, T(tran, dev), T(pl["iCol"], dev), T(pl["iRow"], dev), 2, N,
               offS=T(offS, dev))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert dt < 5.0, "insertion of far-apart draws took %.1f s" % dt
    assert np.abs(F.cpu().numpy() - Fw).max() <= 1e-5 * np.abs(Fw).max()
    assert np.abs(Tt.cpu().numpy() - Tw).max() <= 1e-5 * np.abs(Tw).max()


def test_expect_local_bench_settings_n256(oracle, dev):
    """configs[1] E-step exactly as bench.py launches it: cell-packed projector, Morton-ordered rL = 2 pixel list,
    particle-filter priors (non-uniform pR / pT), 125 rotations x 9 shifts (~1 degree cloud), occupancy cap 2 -- two
    images against the oracle's replay of src/Optimiser.cpp:1225-1406"""
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list, pixel_visit_order
    O = oracle
    rng = np.random.default_rng(2562)
    N, P, nImg, nR, nT = 256, 512, 2, 125, 9
    pl = pixel_list(N, N // 2 - 2, 2)
    assert pl["nPxl"] == 24742
    order = pixel_visit_order(pl, N)
    assert order is not None
    for k in ("iCol", "iRow", "iPxl", "iSig", "iColPad", "iRowPad"):
        pl[k] = np.ascontiguousarray(pl[k][order])
    plan = ops.RecoPlan(N, N, 2)
    vol = plan.set_projectee(T(synth.blob_map(N, nblob=8), dev))
    plan.close()
    vol_h = vol.cpu().numpy()
    quat0 = synth.random_quats(nImg, rng)
    shift0 = rng.normal(0, 2.0, size=(nImg, 2))
    attr = synth.ctf_params(nImg, rng)
    dat, ctf = _noisy_rows(O, vol_h, P, N, pl, quat0, shift0, attr, rng)
    sig = np.broadcast_to((-0.5 / np.mean(np.abs(dat) ** 2, axis=0, keepdims=True)).astype(np.float32), dat.shape).copy()
    q = synth.perturb_quats(quat0, nR, 0.01, rng)
    q[:, 0] = quat0
    rot_h = np.stack([[O.rotate3D(x) for x in qs] for qs in q])
    tran_h = shift0[:, None, :] + rng.normal(0, 0.5, size=(nImg, nT, 2))
    pR = rng.exponential(size=(nImg, nR))
    pR /= pR.sum(1, keepdims=True)
    pT = rng.exponential(size=(nImg, nT))
    pT /= pT.sum(1, keepdims=True)
    cells = ops.pack_projector(vol[None].contiguous(), P)
    res = ops.expect_local(cells, P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(dat, dev), T(ctf, dev), T(sig, dev),
                           T(rot_h, dev), T(tran_h, dev), pR=T(pR, dev), pT=T(pT, dev), want_logW=True, packed=True,
                           wg_per_cu=2)
    del cells
    for l in range(nImg):
        t0 = time.perf_counter()
        want = O.expect_local(vol_h, P, 2, N, pl["iCol"], pl["iRow"], dat[l], ctf[l], sig[l], rot_h[l], tran_h[l], pR=pR[l],
                              pT=pT[l])
        print("expect_local N=256 oracle: %.1f s" % (time.perf_counter() - t0))
        wl = want["logW"][:, :, 0].T            # [nT][nR]
        got = res.logW[l, 0].cpu().numpy()
        # The oracle's (= the reference's scalar) sequential float sum of 24742 terms of like sign carries ~5e-6 |L| of
        # rounding drift; the stated bar between two summation orders is 1e-5 max|L| (src/Optimiser.cpp:25-79).  Against
        # the fp64 value of the same sum the device must be within 5e-7 |C| (C = sum sigRcp |dat|^2).
        scale = np.abs(wl).max()
        assert np.abs(got - wl).max() <= 1e-5 * scale, (np.abs(got - wl).max(), scale)
        Cabs = abs(float(np.sum(sig[l].astype(np.float64) * np.abs(dat[l].astype(np.complex128)) ** 2)))
        for (ir, it) in [(0, 0), (7, 3), (64, 8), (124, 5), (33, 1)]:
            sl = O.project(vol_h, P, 2, rot_h[l, ir], pl["iCol"], pl["iRow"])
            ramp = O.translate(np.float32(tran_h[l, it, 0]), np.float32(tran_h[l, it, 1]), N, pl["iCol"], pl["iRow"])
            exact = O.logDataVSPrior_f64(dat[l], ramp * sl, ctf[l], sig[l])
            assert abs(got[it, ir] - exact) <= 5e-7 * Cabs, (got[it, ir], exact, Cabs)
            assert abs(got[it, ir] - exact) <= abs(wl[it, ir] - exact) + 2e-7 * Cabs   # no worse than the reference's own sum
        tolw = max(2e-5 * scale, 1e-4)
        for name in ("wR", "wT", "wC"):
            g = getattr(res, name)[l].cpu().numpy().reshape(-1)
            np.testing.assert_allclose(g, np.asarray(want[name]).reshape(-1), rtol=3 * tolw, atol=1e-30, err_msg=name)
        # the device's best rotation is (one of) the oracle's best
        wRo = want["wR"].reshape(-1)
        assert wRo[int(res.wR[l].argmax())] >= (1 - 3 * tolw) * wRo.max()


@pytest.mark.parametrize("nT", [16, 32])
def test_expect_local_many_shifts(oracle, dev, nT):
    """the NT = 16 and NT = 32 instantiations of the local-search kernel (the 32-shift stage table needs 68.6 KB of
    dynamic LDS): against the oracle at N = 32, standard and cell-packed layouts"""
    from thunder_amd import ops, synth
    from _util import make_case, make_images
    O = oracle
    rng = np.random.default_rng(40 + nT)
    N, P, nImg, nR = 32, 64, 3, 70
    ref, vol, pl = make_case(O, N, rL=2)
    im = make_images(O, vol, pl, N, nImg, rng)
    q = synth.perturb_quats(im["quat"], nR, 0.05, rng)
    rot_h = np.stack([[O.rotate3D(x) for x in qs] for qs in q])
    tran_h = im["shift"][:, None, :] + rng.normal(0, 0.7, size=(nImg, nT, 2))
    args = (P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(im["dat"], dev), T(im["ctf"], dev), T(im["sigRcp"], dev),
            T(rot_h, dev), T(tran_h, dev))
    a = ops.expect_local(T(vol, dev), *args, want_logW=True)
    b = ops.expect_local(ops.pack_projector(T(vol, dev)[None].contiguous(), P), *args, want_logW=True, packed=True)
    assert torch.equal(a.logW, b.logW)
    for l in range(nImg):
        want = O.expect_local(vol, P, 2, N, pl["iCol"], pl["iRow"], im["dat"][l], im["ctf"][l], im["sigRcp"][l], rot_h[l], tran_h[l])
        wl = want["logW"][:, :, 0].T
        np.testing.assert_allclose(a.logW[l, 0].cpu().numpy(), wl, rtol=0, atol=1e-5 * np.abs(wl).max())
        np.testing.assert_allclose(a.wT[l].cpu().numpy(), want["wT"].reshape(-1), rtol=2e-3)


def test_insert_chunking_bit_identical_n256(dev, knob_env):
    """the bench's box with the record buffer of the brick-sorted insertion cut to ~1 / 12 of what the 48 images need
    (THX_INSERT_SCRATCH_MB=128: a dozen chunks, bricks flushed a dozen times) against the default (one chunk): F and T must be
    bit for bit the same -- every term is rounded once to the session's quanta, whatever chunk it travels in"""
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list
    rng = np.random.default_rng(78)
    N, P, nImg, mReco = 256, 512, 48, 100
    pl = pixel_list(N, N // 2 - 2, 0)
    quat0 = synth.random_quats(nImg, rng)
    quat, tran = _filter_draws(rng, synth, quat0, rng.normal(0, 2.0, size=(nImg, 2)), nImg, 125, 9, mReco, 0.03)
    dat = T((rng.normal(size=(nImg, pl["nPxl"])) + 1j * rng.normal(size=(nImg, pl["nPxl"]))).astype(np.complex64), dev)
    ctf = T(rng.uniform(-1, 1, size=(nImg, pl["nPxl"])).astype(np.float32), dev)
    w = T((rng.uniform(0.2, 1.0, size=nImg) / mReco).astype(np.float32), dev)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev)).reshape(nImg, mReco, 9)
    trn, iCol, iRow = T(tran, dev), T(pl["iCol"], dev), T(pl["iRow"], dev)
    outs = []
    for mb in (None, "128"):
        knob_env("THX_INSERT_SCRATCH_MB", mb)
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, dat, ctf, w, rot, trn, iCol, iRow, 2, N)
        outs.append((F, Tt))
    knob_env("THX_INSERT_SCRATCH_MB", None)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].sum(dtype=torch.float64)) > 0


def test_insert_bit_reproducible_n256(dev):
    """the insertion accumulates in 64-bit fixed point end to end (LDS bricks AND the global volumes, thx_insert.h:acc_add), so
    F and T are bit-identical from run to run whatever the scheduling -- 96 images x 100 filter draws at the bench's box"""
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list
    rng = np.random.default_rng(77)
    N, P, nImg, mReco = 256, 512, 96, 100
    pl = pixel_list(N, N // 2 - 2, 0)
    quat0 = synth.random_quats(nImg, rng)
    quat, tran = _filter_draws(rng, synth, quat0, rng.normal(0, 2.0, size=(nImg, 2)), nImg, 125, 9, mReco, 0.03)
    dat = T((rng.normal(size=(nImg, pl["nPxl"])) + 1j * rng.normal(size=(nImg, pl["nPxl"]))).astype(np.complex64), dev)
    ctf = T(rng.uniform(-1, 1, size=(nImg, pl["nPxl"])).astype(np.float32), dev)
    w = T((rng.uniform(0.2, 1.0, size=nImg) / mReco).astype(np.float32), dev)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev)).reshape(nImg, mReco, 9)
    trn, iCol, iRow = T(tran, dev), T(pl["iCol"], dev), T(pl["iRow"], dev)
    outs = []
    for rep in range(3):
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, dat, ctf, w, rot, trn, iCol, iRow, 2, N)
        if rep == 1:   # perturb the scheduling: other work in flight on a second stream
            s2 = torch.cuda.Stream()
            with torch.cuda.stream(s2):
                junk = torch.randn(1 << 26, device=dev).cumsum(0)
            s2.synchronize()
            del junk
        outs.append((F, Tt))
    for F, Tt in outs[1:]:
        assert torch.equal(F, outs[0][0]) and torch.equal(Tt, outs[0][1])
    mass = float(Tt.sum(dtype=torch.float64))
    want = float((w.double()[:, None] * ctf.double() ** 2).sum()) * mReco
    assert abs(mass - want) <= 2e-5 * want


def test_reconstruct_vs_oracle_p512(oracle, dev):
    """Reconstructor::reconstruct at the bench's grid (N = 256, P = 512; src/Reconstructor.cpp:1129-1831, convoluteC :2595-2674,
    checkC :2563-2592): thx_reco_reconstruct_dev -- the hand-written radix-8 passes -- against the oracle on the F / T of a
    bench-like insertion (3 000 images x 100 resampled filter draws), MAP off and MAP on (joinHalf, a decaying FSC).
    The balancing loop's stop rule fires on a max norm that jitters (tests/test_iteration_cpu.py::test_stop_rule_is_noise_
    sensitive); on IDENTICAL inputs both sides are expected to stop after the same round, and that is asserted.
    Bars of SURVEY 8c (9): 1e-4 of max, FSC >= 0.9999 per shell, checkC's distance to 3 digits."""
    import os
    import scipy.fft as sfft
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list
    O = oracle
    rng = np.random.default_rng(78)
    N, P, nImg, mReco = 256, 512, 3000, 100       # enough views for the balancing loop to converge instead of jittering
    rU = N // 2 - 2
    pl = pixel_list(N, rU, 0)
    iCol, iRow = T(pl["iCol"], dev), T(pl["iRow"], dev)
    plan = ops.RecoPlan(N, N, 2)
    vol = plan.set_projectee(T(synth.blob_map(N, nblob=8), dev))
    quat0 = synth.random_quats(nImg, rng)
    shift0 = rng.normal(0, 2.0, size=(nImg, 2))
    attr = T(synth.ctf_params(nImg, rng), dev)
    F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev).manual_seed(9)
    for b0 in range(0, nImg, 500):          # rows made by the product's own kernels (data generation, not the thing tested)
        b1 = min(nImg, b0 + 500)
        ctf = ops.ctf(attr[b0:b1].contiguous(), 1.32, iCol, iRow, N)
        sig = ops.project(vol, ops.rotmat(T(quat0[b0:b1], dev)), iCol, iRow, 2) * ctf * ops.translate(T(shift0[b0:b1], dev), iCol, iRow, N)
        dat = (sig + torch.view_as_complex(torch.randn(sig.shape + (2,), generator=g, device=dev)) * (3.0 * float(sig.abs().pow(2).mean().sqrt()) / np.sqrt(2))).contiguous()
        quat, tran = _filter_draws(rng, synth, quat0[b0:b1], shift0[b0:b1], b1 - b0, 125, 9, mReco, 0.03)
        rot = ops.rotmat(T(quat.reshape(-1, 4), dev)).reshape(b1 - b0, mReco, 9)
        w = torch.full((b1 - b0,), 1.0 / mReco, dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, dat, ctf.contiguous(), w, rot, T(tran, dev), iCol, iRow, 2, N)
    del vol
    ops.normalise_TF(F, Tt, P)
    Fh, Th = F.cpu().numpy(), Tt.cpu().numpy()
    fsc = np.clip(1.2 - np.arange(rU) / (0.6 * rU), 0.02, 1.0).astype(np.float32)
    t0 = time.perf_counter()
    with sfft.set_workers(os.cpu_count() or 1):
        for MAP in (False, True):
            m_dev = plan.reconstruct(F, Tt, rU, FSC=fsc, joinHalf=True, MAP=MAP, gridCorr=True).cpu().numpy()   # T in place, as the reference
            k_dev, d_dev = plan.last_iters, plan.last_diffC
            m_or, k_or, diffs, _ = O.reconstruct(Fh, Th, P, N, 2, rU, FSC=fsc, joinHalf=True, MAP=MAP, gridCorr=True, return_iters=True,
                                                 T_inplace=True)
            note = ""
            assert k_or == k_dev, "round counts differ on identical inputs: device %d oracle %d, diffC %s" % (k_dev, k_or, diffs[-4:])
            e = np.abs(m_dev - m_or).max() / np.abs(m_or).max()
            fs = O.fsc(sfft.rfftn(m_dev).astype(np.complex64), sfft.rfftn(m_or).astype(np.complex64), N, rU)
            print("P = 512, MAP %s: rounds device %d oracle %d%s, diffC device %.5f oracle %.5f, map %.2e of max, min FSC %.6f"
                  % ("on" if MAP else "off", k_dev, k_or, note, d_dev, diffs[-1], e, fs.min()))
            assert abs(d_dev - diffs[-1]) <= 1e-3 * diffs[-1]
            assert e <= 1e-4 and fs.min() >= 0.9999
    print("oracle + device reconstructions at P = 512: %.0f s" % (time.perf_counter() - t0))
    plan.close()


def _expect_local_vs_oracle(O, dev, N, nImg, nR, nT, K, seed, spread):
    """k_expect_local (cell-packed, Morton list, filter priors, occupancy cap 2, volIdx when K > 1) against the oracle's replay of
    src/Optimiser.cpp:1225-1406 for every image: logW of every (rotation, shift) at 1e-5 max|L|, the weights, the best rotation"""
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list, pixel_visit_order
    rng = np.random.default_rng(seed)
    P = 2 * N
    pl = pixel_list(N, N // 2 - 2, 2)
    order = pixel_visit_order(pl, N)
    for k in ("iCol", "iRow", "iPxl", "iSig", "iColPad", "iRowPad"):
        pl[k] = np.ascontiguousarray(pl[k][order])
    plan = ops.RecoPlan(N, N, 2)
    vols = torch.stack([plan.set_projectee(T(synth.blob_map(N, seed=seed + 10 * k, nblob=8), dev)) for k in range(K)]).contiguous()
    plan.close()
    vol_h = [vols[k].cpu().numpy() for k in range(K)]
    cls = (np.arange(nImg) % K).astype(np.int32)[::-1].copy()           # every class is used; not in storage order
    quat0 = synth.random_quats(nImg, rng)
    shift0 = rng.normal(0, 2.0, size=(nImg, 2))
    attr = synth.ctf_params(nImg, rng)
    dat = np.zeros((nImg, pl["nPxl"]), np.complex64)
    ctf = np.zeros((nImg, pl["nPxl"]), np.float32)
    for l in range(nImg):
        d, c = _noisy_rows(O, vol_h[cls[l]], P, N, pl, quat0[l:l + 1], shift0[l:l + 1], attr[l:l + 1], rng)
        dat[l], ctf[l] = d[0], c[0]
    sig = np.broadcast_to((-0.5 / np.mean(np.abs(dat) ** 2, axis=0, keepdims=True)).astype(np.float32), dat.shape).copy()
    q = synth.perturb_quats(quat0, nR, spread, rng)
    q[:, 0] = quat0
    rot_h = np.stack([[O.rotate3D(x) for x in qs] for qs in q])
    tran_h = shift0[:, None, :] + rng.normal(0, 0.5, size=(nImg, nT, 2))
    pR = rng.exponential(size=(nImg, nR)); pR /= pR.sum(1, keepdims=True)
    pT = rng.exponential(size=(nImg, nT)); pT /= pT.sum(1, keepdims=True)
    cells = ops.pack_projector(vols, P)
    del vols
    torch.cuda.empty_cache()
    res = ops.expect_local(cells, P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(dat, dev), T(ctf, dev), T(sig, dev),
                           T(rot_h, dev), T(tran_h, dev), pR=T(pR, dev), pT=T(pT, dev), want_logW=True, packed=True, wg_per_cu=2,
                           volIdx=T(cls, dev) if K > 1 else None)
    torch.cuda.synchronize()
    del cells
    torch.cuda.empty_cache()
    for l in range(nImg):
        t0 = time.perf_counter()
        want = O.expect_local(vol_h[cls[l]], P, 2, N, pl["iCol"], pl["iRow"], dat[l], ctf[l], sig[l], rot_h[l], tran_h[l], pR=pR[l], pT=pT[l])
        wl = want["logW"][:, :, 0].T            # [nT][nR]
        got = res.logW[l, 0].cpu().numpy()
        scale = np.abs(wl).max()
        e = np.abs(got - wl).max()
        print("expect_local N=%d K=%d image %d (class %d): |logW - oracle| %.2e of max|L|, oracle %.1f s" % (N, K, l, cls[l], e / scale, time.perf_counter() - t0))
        # the oracle's (= the reference's scalar) SEQUENTIAL float sum drifts with the number of terms: 1e-5 max|L| is the bar the
        # reference states between its own two summation orders (src/Optimiser.cpp:25-79) at the 24 742 pixels of a 256^3 box; a
        # 512^3 box sums 100 941 terms (measured here: 1.1e-5), so the bar follows sqrt(terms).  What does NOT scale is the
        # device's distance to the fp64 value of the same sum, held to 5e-7 |C| below -- and to the oracle's own distance.
        bar = 1e-5 * max(1.0, np.sqrt(pl["nPxl"] / 24742.0))
        assert e <= bar * scale, (e, scale, bar)
        Cabs = abs(float(np.sum(sig[l].astype(np.float64) * np.abs(dat[l].astype(np.complex128)) ** 2)))
        for (ir, it) in [(0, 0), (nR - 1, nT - 1), (nR // 2, 1)]:
            sl = O.project(vol_h[cls[l]], P, 2, rot_h[l, ir], pl["iCol"], pl["iRow"])
            ramp = O.translate(np.float32(tran_h[l, it, 0]), np.float32(tran_h[l, it, 1]), N, pl["iCol"], pl["iRow"])
            exact = O.logDataVSPrior_f64(dat[l], ramp * sl, ctf[l], sig[l])
            assert abs(got[it, ir] - exact) <= 5e-7 * Cabs, (got[it, ir], exact, Cabs)
            assert abs(got[it, ir] - exact) <= abs(wl[it, ir] - exact) + 2e-7 * Cabs   # no worse than the reference's own sum
        tolw = max(2e-5 * scale, 1e-4)
        for name in ("wR", "wT", "wC"):
            g = getattr(res, name)[l].cpu().numpy().reshape(-1)
            np.testing.assert_allclose(g, np.asarray(want[name]).reshape(-1), rtol=3 * tolw, atol=1e-30, err_msg=name)
        wRo = want["wR"].reshape(-1)
        assert wRo[int(res.wR[l].argmax())] >= (1 - 3 * tolw) * wRo.max()
        if K > 1:      # against ANOTHER class's volume the likelihoods are different numbers: volIdx really selects
            other = O.expect_local(vol_h[(cls[l] + 1) % K], P, 2, N, pl["iCol"], pl["iRow"], dat[l], ctf[l], sig[l], rot_h[l][:2], tran_h[l][:2])
            assert np.abs(other["logW"][:, :, 0].T - got[:2, :2]).max() > 1e-3 * scale


def test_expect_local_vs_oracle_n512(oracle, dev):
    """configs[4] box: the cell-packed E-step at N = 512 (P = 1024: 34 GB of cells, element offsets beyond 2^32) against the
    oracle -- 2 images x 12 rotations x 3 shifts, logW and weights (until round 5 this size was held packed == unpacked only)"""
    _expect_local_vs_oracle(oracle, dev, 512, 2, 12, 3, 1, seed=5121, spread=0.006)


def test_expect_local_vs_oracle_k4_n256(oracle, dev):
    """configs[3] at full size: the local phase with volIdx over FOUR 4.3 GB cell-packed references at 256^3 -- 4 images (one per
    class), 125 x 9 support points with a 3-degree cloud (after a scan), against the oracle on each image's own reference"""
    _expect_local_vs_oracle(oracle, dev, 256, 4, 125, 9, 4, seed=2563, spread=0.03)


def test_reconstruct_vs_oracle_p1024(oracle, dev):
    """configs[4] grid: thx_reco_reconstruct_dev at P = 1024 -- the <3, 2> instances of the hand-written passes, incl. the z pass that
    runs one workgroup per CU -- against the oracle (src/Reconstructor.cpp:1129-1831) with the number of balancing rounds pinned
    to 3 on both sides (thx_reco_set_balance_rounds / force_rounds), MAP on with joinHalf.  Inputs analytic (a smooth sampling
    density; F = reference x T).  Bars of SURVEY 8c (9): 1e-4 of max, FSC >= 0.9999 per shell."""
    import os
    import scipy.fft as sfft
    from thunder_amd import ops, synth
    O = oracle
    N, P = 512, 1024
    rU = N // 2 - 2
    plan = ops.RecoPlan(N, N, 2)
    vol = plan.set_projectee(T(synth.blob_map(N, nblob=8), dev))
    ax = torch.fft.fftfreq(P, d=1.0 / P, device=dev)
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :P // 2 + 1] ** 2)
    Tt = (1.0 / (1.0 + r / 8.0)).to(torch.float32).contiguous()
    Tt[r >= rU * 2 + 1] = 0
    del r
    # (a white component on top of the blobs' transform: every shell up to rU carries signal, so that the per-shell FSC below compares
    # maps and not the rounding noise of empty shells)
    g = torch.Generator(device=dev).manual_seed(1024)
    amp = float(vol.abs()[:8, :8, :8].mean()) * 1e-3
    vol += torch.view_as_complex(torch.randn(vol.shape + (2,), generator=g, device=dev, dtype=torch.float32)) * amp
    F = (vol * Tt).contiguous()
    del vol
    fsc = np.clip(1.2 - np.arange(rU) / (0.6 * rU), 0.02, 1.0).astype(np.float32)
    Fh, Th = F.cpu().numpy(), Tt.cpu().numpy()
    plan.set_balance_rounds(3, 3)
    m_dev = plan.reconstruct(F, Tt, rU, FSC=fsc, joinHalf=True, MAP=True, gridCorr=True).cpu().numpy()
    assert plan.last_iters == 3
    d_dev = plan.last_diffC
    plan.close()
    del F, Tt
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    with sfft.set_workers(os.cpu_count() or 1):
        m_or, k_or, diffs, _ = O.reconstruct(Fh, Th, P, N, 2, rU, FSC=fsc, joinHalf=True, MAP=True, gridCorr=True, return_iters=True,
                                             force_rounds=3)
        e = np.abs(m_dev - m_or).max() / np.abs(m_or).max()
        fs = O.fsc(sfft.rfftn(m_dev).astype(np.complex64), sfft.rfftn(m_or).astype(np.complex64), N, rU)
    print("P = 1024, 3 rounds: diffC device %.5f oracle %.5f, map %.2e of max, min FSC %.6f, oracle %.0f s" % (d_dev, diffs[-1], e, fs.min(), time.perf_counter() - t0))
    assert k_or == 3 and abs(d_dev - diffs[-1]) <= 1e-3 * diffs[-1]
    assert e <= 1e-4 and fs.min() >= 0.9999
