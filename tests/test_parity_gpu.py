"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars:
  * integer / index work and the trilinear gather (project, symmetrise, normalise): bit-exact;
  * transcendental rows (CTF, ramps): |delta| <= 5e-7 (device sincosf / cosf vs glibc, <= 2 ulp);
  * sums whose order the reference itself does not fix (likelihood, atomically inserted F/T, FSC, FFT pipelines):
    tolerance stated at each assert.  The only tolerance the reference states for this path is |delta| < 1e-5 between its
    own scalar and SIMD likelihoods (src/Optimiser.cpp:25-79).
"""
import ctypes as C

import numpy as np
import pytest
import scipy.fft as sfft

from _util import edge_rotations, make_case, make_images, quat_to_mat

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bits(a):
    return np.ascontiguousarray(a).view(np.int32 if a.dtype.itemsize in (4, 8) and a.dtype.kind != "c" else np.int32)


def assert_bit_equal(a, b, what):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, what
    same = (a.view(np.uint8) == b.view(np.uint8))
    if not same.all():
        # +0 / -0 are the same number; anything else is a failure
        assert np.array_equal(a, b), "%s: %d differing elements, max |d| %g" % (
            what, int((a != b).sum()), float(np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max()))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [16, 32, 64])
def test_project_bit_exact(oracle, dev, N):
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(100 + N)
    ref, vol, pl = make_case(O, N)
    P = 2 * N
    # a dense random volume exercises every voxel, not only the smooth blob spectrum
    vol = (vol + (rng.normal(size=vol.shape) + 1j * rng.normal(size=vol.shape)).astype(np.complex64)).astype(np.complex64)
    mats = edge_rotations(rng, n_random=10)
    want = np.stack([O.project(vol, P, 2, m, pl["iCol"], pl["iRow"]) for m in mats])
    got = ops.project(T(vol, dev), T(mats, dev), T(pl["iCol"], dev), T(pl["iRow"], dev), 2).cpu().numpy()
    assert_bit_equal(got, want, "project N=%d" % N)


def test_project_empty_and_ragged(oracle, dev):
    from thunder_amd import ops
    O = oracle
    ref, vol, pl = make_case(O, 16, rU=6, rL=2)
    assert pl["nPxl"] > 0 and pl["nPxl"] % 64 != 0
    mats = edge_rotations(np.random.default_rng(1), 2)[:3]
    got = ops.project(T(vol, dev), T(mats, dev), T(pl["iCol"], dev), T(pl["iRow"], dev), 2).cpu().numpy()
    want = np.stack([O.project(vol, 32, 2, m, pl["iCol"], pl["iRow"]) for m in mats])
    assert_bit_equal(got, want, "ragged pixel list")
    # zero rotations: a no-op that must not fail
    out = ops.project(T(vol, dev), torch.empty((0, 9), dtype=torch.float64, device=dev), T(pl["iCol"], dev),
                      T(pl["iRow"], dev), 2)
    assert out.shape[0] == 0


def test_rotmat_translate_ctf(oracle, dev):
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(3)
    N = 64
    pl = O.pixel_list(N, N // 2 - 2, 0)
    q = synth.random_quats(50, rng)
    got = ops.rotmat(T(q, dev)).cpu().numpy()
    want = np.stack([O.rotate3D(x) for x in q])
    np.testing.assert_allclose(got, want, rtol=0, atol=4e-16)
    shifts = np.concatenate([rng.normal(0, 3, size=(6, 2)), [[0, 0], [N / 2, -N / 2]]])
    got = ops.translate(T(shifts, dev), T(pl["iCol"], dev), T(pl["iRow"], dev), N).cpu().numpy()
    want = np.stack([O.translate(np.float32(s[0]), np.float32(s[1]), N, pl["iCol"], pl["iRow"]) for s in shifts])
    assert np.abs(got - want).max() <= 5e-7
    attr = synth.ctf_params(5, rng)
    attr[4, 6] = 0.3  # phase plate
    got = ops.ctf(T(attr, dev), 1.32, T(pl["iCol"], dev), T(pl["iRow"], dev), N).cpu().numpy()
    want = np.stack([O.ctf(1.32, *a, N, pl["iCol"], pl["iRow"]) for a in attr])
    # chi is a float of up to ~100-200 rad here (ulp 7.6e-6 .. 1.5e-5) built with identical operations on both sides, so
    # most values are bit-equal; at isolated pixels the device's cosf(2 angle) and glibc's differ by one ulp, the
    # defocus term rounds the other way and chi moves by 1-2 ulp.  Measured (tools/ctf_debug.py): max 8e-6 at N = 64,
    # 1.5e-5 at N = 256, mean 1.2e-8; both are 5e-5 from the double-precision value of the formula.  Bar: 4 ulp of chi.
    assert np.abs(got - want).max() <= 4 * 7.7e-6
    assert np.abs(got - want).mean() <= 2e-7
    assert np.mean(got == want) >= 0.7     # 82 % bit-equal here (N = 64, one phase-plate row), 99 % in tools/ctf_debug.py's set


def test_gather_pixels(oracle, dev):
    from thunder_amd import ops
    N = 32
    pl = oracle.pixel_list(N, 12, 1)
    rng = np.random.default_rng(0)
    img = (rng.normal(size=(3, N, N // 2 + 1)) + 1j * rng.normal(size=(3, N, N // 2 + 1))).astype(np.complex64)
    got = ops.gather_pixels(T(img, dev), T(pl["iPxl"], dev), N).cpu().numpy()
    want = img.reshape(3, -1)[:, pl["iPxl"]]
    assert_bit_equal(got, want, "allocPreCal gather")


def test_logDataVSPrior(oracle, dev):
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(11)
    N = 64
    ref, vol, pl = make_case(O, N)
    im = make_images(O, vol, pl, N, 2, rng)
    pri = np.stack([O.project(vol, 2 * N, 2, im["rot"][k % 2], pl["iCol"], pl["iRow"]) for k in range(5)])
    got = ops.logDataVSPrior(T(im["dat"][0], dev), T(pri, dev), T(im["ctf"][0], dev), T(im["sigRcp"][0], dev)).cpu().numpy()
    for k in range(5):
        exact = O.logDataVSPrior_f64(im["dat"][0], pri[k], im["ctf"][0], im["sigRcp"][0])
        ref32 = O.logDataVSPrior(im["dat"][0], pri[k], im["ctf"][0], im["sigRcp"][0])
        # the device tree sum must be at least as close to the exact value as the reference's own
        # sequential float sum, up to 2 ulp of the result
        tol = abs(ref32 - exact) + 2 * np.spacing(np.float32(abs(exact)))
        assert abs(got[k] - exact) <= max(tol, 1e-6 * abs(exact))


@pytest.mark.parametrize("N,nR,nT", [(32, 20, 9), (64, 125, 9), (32, 70, 3), (32, 5, 12)])
def test_expect_local(oracle, dev, N, nR, nT):
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(200 + N + nR)
    ref, vol, pl = make_case(O, N, rL=1)
    P = 2 * N
    nImg = 3
    im = make_images(O, vol, pl, N, nImg, rng, snr_sigma=2.0)
    quat = synth.perturb_quats(im["quat"], nR, 0.04, rng)            # [nImg][nR][4]
    rot = np.stack([[O.rotate3D(q) for q in qs] for qs in quat])      # [nImg][nR][9]
    tran = im["shift"][:, None, :] + rng.normal(0, 0.5, size=(nImg, nT, 2))
    pR = rng.uniform(0.5, 1.5, size=(nImg, nR))
    pT = rng.uniform(0.5, 1.5, size=(nImg, nT))
    pC = rng.uniform(0.5, 1.5, size=nImg)
    res = ops.expect_local(T(vol, dev), P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(im["dat"], dev),
                           T(im["ctf"], dev), T(im["sigRcp"], dev), T(rot, dev), T(tran, dev), nD=1, pC=T(pC, dev),
                           pR=T(pR, dev), pT=T(pT, dev), want_logW=True)
    logW = res.logW.cpu().numpy()  # [nImg][1][nT][nR]
    for l in range(nImg):
        want = O.expect_local(vol, P, 2, N, pl["iCol"], pl["iRow"], im["dat"][l], im["ctf"][l], im["sigRcp"][l],
                              rot[l], tran[l], nD=1, pC=pC[l], pR=pR[l], pT=pT[l])
        wl = want["logW"][:, :, 0].T  # [nT][nR]
        scale = np.abs(wl).max()
        # log-likelihoods: float sums of nPxl terms; 1e-5 relative covers both summation orders
        np.testing.assert_allclose(logW[l, 0], wl, rtol=0, atol=1e-5 * scale)
        assert abs(res.baseLine[l].item() - want["baseLine"]) <= 1e-5 * scale
        # weights are exp(L - max): an absolute error e in L is a relative error e in the weight
        tolw = max(2e-5 * scale, 1e-4)
        for name in ("wR", "wT", "wC"):
            g = getattr(res, name)[l].cpu().numpy().reshape(-1)
            w = want[name].reshape(-1)
            np.testing.assert_allclose(g, w, rtol=3 * tolw, atol=1e-30, err_msg=name)
        np.testing.assert_allclose(res.wD[l].cpu().numpy(), want["wD"], rtol=3 * tolw)


def test_expect_local_dsearch_and_volidx(oracle, dev):
    """nD > 1 (CTF search rows) and two reference volumes selected per image"""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(5)
    N, nR, nT, nD, nImg = 32, 10, 4, 3, 2
    ref, vol, pl = make_case(O, N)
    _, vol2, _ = make_case(O, N, seed=99)
    P = 2 * N
    im = make_images(O, vol, pl, N, nImg, rng)
    rot = np.stack([[O.rotate3D(q) for q in qs] for qs in synth.perturb_quats(im["quat"], nR, 0.05, rng)])
    tran = im["shift"][:, None, :] + rng.normal(0, 0.5, size=(nImg, nT, 2))
    dfac = 1.0 + rng.normal(0, 0.02, size=(nImg, nD))
    ctfD = np.stack([[O.ctf(1.32, a[0], np.float32(a[1] * d), np.float32(a[2] * d), *a[3:], N, pl["iCol"], pl["iRow"])
                      for d in dfac[l]] for l, a in enumerate(im["attr"])])  # [nImg][nD][nPxl]
    pD = rng.uniform(0.5, 1.5, size=(nImg, nD))
    vols = np.stack([vol, vol2])
    volIdx = np.array([1, 0], np.int32)
    res = ops.expect_local(T(vols, dev), P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(im["dat"], dev),
                           T(ctfD, dev), T(im["sigRcp"], dev), T(rot, dev), T(tran, dev), nD=nD, volIdx=T(volIdx, dev),
                           pD=T(pD, dev), want_logW=True)
    for l in range(nImg):
        want = O.expect_local(vols[volIdx[l]], P, 2, N, pl["iCol"], pl["iRow"], im["dat"][l], ctfD[l], im["sigRcp"][l],
                              rot[l], tran[l], nD=nD, pD=pD[l], cSearch=True)
        wl = np.transpose(want["logW"], (2, 1, 0))  # [nD][nT][nR]
        scale = np.abs(wl).max()
        np.testing.assert_allclose(res.logW[l].cpu().numpy(), wl, rtol=0, atol=1e-5 * scale)
        for name in ("wR", "wT", "wD", "wC"):
            np.testing.assert_allclose(getattr(res, name)[l].cpu().numpy().reshape(-1), want[name].reshape(-1),
                                       rtol=1e-3, err_msg=name)


@pytest.mark.parametrize("nR,nT,nImg,form", [(70, 11, 5, "small"), (301, 11, 37, "tiled"), (301, 11, 37, "simple"),
                                                (257, 35, 8, "tiled")])   # 35 shifts: the general fold behind the tiled contraction
def test_expect_global(oracle, dev, knob_env, nR, nT, nImg, form):
    """the scanning stage against the oracle: the rotation-per-thread kernel (small problems), the LDS-tiled contraction
    (nImg * nT >= 256 and nR >= 256; ragged tiles on purpose) and the former forced onto the latter's sizes -- the two forms
    accumulate in the same order, so their outputs must be bit-identical"""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(17)
    N, nK = 32, 2
    if form == "simple":
        knob_env("THX_SCAN", "simple")
    ref, vol, pl = make_case(O, N, rU=10)
    _, vol2, _ = make_case(O, N, seed=31, rU=10)
    P = 2 * N
    im = make_images(O, vol, pl, N, nImg, rng)
    mats = np.stack([O.rotate3D(q) for q in synth.random_quats(nR, rng)])
    mats[:nImg] = im["rot"]
    shifts = rng.normal(0, 2, size=(nT, 2))
    pR = rng.uniform(0.5, 1.5, size=(nImg, nR))
    pT = rng.uniform(0.5, 1.5, size=(nImg, nT))
    traP_h = np.stack([O.translate(np.float32(s[0]), np.float32(s[1]), N, pl["iCol"], pl["iRow"]) for s in shifts])
    wC = np.zeros((nImg, nK), np.float32)
    wR = np.zeros((nK, nImg, nR), np.float32)
    wT = np.zeros((nK, nImg, nT), np.float32)
    base = np.full(nImg, np.nan, np.float32)
    d_wC, d_wR, d_wT, d_base = T(wC, dev), T(wR, dev), T(wT, dev), T(base, dev)
    dat_pm = np.ascontiguousarray(im["dat"].T)
    ctf_pm = np.ascontiguousarray(im["ctf"].T)
    sig_pm = np.ascontiguousarray(im["sigRcp"].T)
    iCol, iRow = T(pl["iCol"], dev), T(pl["iRow"], dev)
    traP = ops.translate(T(shifts, dev), iCol, iRow, N)
    for k, v in enumerate((vol, vol2)):
        rotP_h = np.stack([O.project(v, P, 2, m, pl["iCol"], pl["iRow"]) for m in mats])
        O.expect_global(rotP_h, traP_h, dat_pm, ctf_pm, sig_pm, nK, k, pR, pT, wC, wR, wT, base)
        rotP = ops.project(T(v, dev), T(mats, dev), iCol, iRow, 2)
        ops.expect_global(rotP, traP, T(im["dat"], dev), T(im["ctf"], dev), T(im["sigRcp"], dev), T(pR, dev), T(pT, dev),
                          d_wC, d_wR, d_wT, d_base, k, nK)
    scale = np.abs(base).max()
    assert np.abs(d_base.cpu().numpy() - base).max() <= 1e-5 * scale
    tol = max(6e-5 * scale, 3e-4)
    np.testing.assert_allclose(d_wC.cpu().numpy(), wC, rtol=tol)
    np.testing.assert_allclose(d_wR.cpu().numpy(), wR, rtol=tol, atol=1e-30)
    np.testing.assert_allclose(d_wT.cpu().numpy(), wT, rtol=tol, atol=1e-30)
    got = {k: v.cpu().numpy().copy() for k, v in (("wC", d_wC), ("wR", d_wR), ("wT", d_wT), ("base", d_base))}
    if form == "tiled":
        _scan_forms[(nR, nT, nImg)] = got
    elif form == "simple" and (nR, nT, nImg) in _scan_forms:
        for k, v in got.items():
            assert np.array_equal(v, _scan_forms[(nR, nT, nImg)][k]), k


_scan_forms = {}


def test_expect_global_forms_bit_identical_scan_size(dev, knob_env):
    """the two forms of the scanning stage at the classification scan's pixel count (866) and shift count (30), ragged in
    images and rotations, two classes swept twice (so the carried baseline is rescaled): the f32-MFMA contraction must
    reproduce the rotation-per-thread kernel bit for bit -- its sums run over k in the same order with one rounding per
    product"""
    from thunder_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    nImg, nT, nR, nPxl, nK = 129, 30, 1001, 866, 2

    def cplx(*shape, scale=1.0):
        return (torch.randn(*shape, 2, generator=g) * scale).view(*shape, 2)

    rot = [torch.view_as_complex(cplx(nR, nPxl).contiguous()).to(dev) for _ in range(nK)]
    ang = torch.rand(nT, nPxl, generator=g, dtype=torch.float64) * 6.283185307179586
    tra = torch.complex(torch.cos(ang), torch.sin(ang)).to(torch.complex64).to(dev)
    dat = torch.view_as_complex(cplx(nImg, nPxl).contiguous()).to(dev)
    ctf = (torch.rand(nImg, nPxl, generator=g) * 2 - 1).to(dev)
    sig = (-0.5 / (0.5 + torch.rand(nImg, nPxl, generator=g))).to(dev) * 1e-2
    pR = (0.5 + torch.rand(nImg, nR, generator=g, dtype=torch.float64)).to(dev)
    pT = (0.5 + torch.rand(nImg, nT, generator=g, dtype=torch.float64)).to(dev)
    out = {}
    for form in ("simple", "tiled"):
        knob_env("THX_SCAN", "simple" if form == "simple" else None)
        wC = torch.zeros((nImg, nK), dtype=torch.float32, device=dev)
        wR = torch.zeros((nK, nImg, nR), dtype=torch.float32, device=dev)
        wT = torch.zeros((nK, nImg, nT), dtype=torch.float32, device=dev)
        base = torch.full((nImg,), float("nan"), dtype=torch.float32, device=dev)
        for sweep in range(2):
            for k in range(nK):
                ops.expect_global(rot[(k + sweep) % nK], tra, dat, ctf, sig, pR, pT, wC, wR, wT, base, k, nK)
        out[form] = [t.cpu() for t in (wC, wR, wT, base)]
        assert all(torch.isfinite(t).all() for t in out[form])
    for a, b, name in zip(out["simple"], out["tiled"], ("wC", "wR", "wT", "base")):
        assert torch.equal(a, b), name
    assert out["tiled"][1].abs().max() > 0


# ---------------------------------------------------------------------------------------------
def _insert_case(O, N, nImg, mReco, rng, nK=1):
    from thunder_amd import synth
    ref, vol, pl = make_case(O, N)
    im = make_images(O, vol, pl, N, nImg, rng, snr_sigma=0.2)
    quat = synth.perturb_quats(im["quat"], mReco, 0.03, rng)
    tran = im["shift"][:, None, :] + rng.normal(0, 0.3, size=(nImg, mReco, 2))
    offS = rng.normal(0, 0.2, size=(nImg, 2))
    w = (rng.uniform(0.5, 1.0, size=nImg) / mReco).astype(np.float32)
    cls = rng.integers(0, nK, size=(nImg, mReco)).astype(np.int32)
    return ref, vol, pl, im, quat, tran, offS, w, cls


def _oracle_insert(O, P, N, pl, im, quat, tran, offS, w, cls, nK):
    F = np.zeros((nK, P, P, P // 2 + 1), np.complex64)
    Tt = np.zeros((nK, P, P, P // 2 + 1), np.float32)
    Osum = np.zeros(3)
    for l in range(len(w)):
        for m in range(quat.shape[1]):
            R = O.rotate3D(quat[l, m])
            t = tran[l, m] - offS[l]
            src = O.translate(np.float32(-t[0]), np.float32(-t[1]), N, pl["iCol"], pl["iRow"], src=im["dat"][l])
            O.insertP(F[cls[l, m]], Tt[cls[l, m]], P, src, im["ctf"][l], R, w[l], pl["iColPad"], pl["iRowPad"])
            Rm = R.reshape(3, 3).T
            Osum += -(Rm @ np.array([t[0], t[1], 0.0]))
    return F, Tt, Osum


@pytest.mark.parametrize("N,nK", [(32, 1), (32, 3), (64, 1)])
def test_insert(oracle, dev, N, nK):
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(300 + N + nK)
    nImg, mReco = 6, 8
    P = 2 * N
    ref, vol, pl, im, quat, tran, offS, w, cls = _insert_case(O, N, nImg, mReco, rng, nK)
    Fw, Tw, Ow = _oracle_insert(O, P, N, pl, im, quat, tran, offS, w, cls, nK)
    F = torch.zeros((nK, P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((nK, P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    Od = torch.zeros(3, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev))
    ops.insert(F, Tt, P, T(im["dat"], dev), T(im["ctf"], dev), T(w, dev), rot, T(tran, dev), T(pl["iCol"], dev),
               T(pl["iRow"], dev), 2, N, O=Od, counter=cnt, offS=T(offS, dev), cls=T(cls, dev), nK=nK)
    Fg, Tg = F.cpu().numpy(), Tt.cpu().numpy()
    # atomic summation order differs run to run; each voxel sums <= nImg*mReco*few terms of like magnitude.
    # bar: 1e-5 of the largest accumulated magnitude -- the ramp sincosf adds ~1e-7 relative.
    assert np.abs(Fg - Fw).max() <= 1e-5 * np.abs(Fw).max()
    assert np.abs(Tg - Tw).max() <= 1e-5 * np.abs(Tw).max()
    # untouched voxels stay exactly zero; voxels the device left at zero carry at most a sub-quantum contribution
    # (every term is rounded to the session's 64-bit quantum, 2^-30 or less of the largest possible term)
    assert not np.any((Fw == 0) & (Fg != 0)) and not np.any((Tw == 0) & (Tg != 0))
    assert np.abs(Fw[Fg == 0]).max(initial=0) <= 1e-6 * np.abs(Fw).max()
    assert np.abs(Tw[Tg == 0]).max(initial=0) <= 1e-6 * np.abs(Tw).max()
    assert cnt.item() == nImg * mReco
    np.testing.assert_allclose(Od.cpu().numpy(), Ow, rtol=1e-12, atol=1e-12)


def test_insert_unrelated_draws_and_plain_kernel(oracle, dev, knob_env):
    """draws that are NOT nearby orientations (every draw its own plane through the volume: many bricks per pass of k_bin);
    the plain kernel (THX_INSERT_PLAIN=1) and the brick-sorted form must both match the oracle"""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(77)
    N, nImg, mReco = 32, 3, 6
    P = 2 * N
    ref, vol, pl, im, quat, tran, offS, w, cls = _insert_case(O, N, nImg, mReco, rng)
    quat = synth.random_quats(nImg * mReco, rng).reshape(nImg, mReco, 4)
    Fw, Tw, _ = _oracle_insert(O, P, N, pl, im, quat, tran, offS, w, np.zeros_like(cls), 1)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev))
    for plain in ("0", "1"):
        knob_env("THX_INSERT_PLAIN", plain)
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, T(im["dat"], dev), T(im["ctf"], dev), T(w, dev), rot, T(tran, dev), T(pl["iCol"], dev),
                   T(pl["iRow"], dev), 2, N, offS=T(offS, dev))
        assert np.abs(F.cpu().numpy() - Fw[0]).max() <= 1e-5 * np.abs(Fw).max(), plain
        assert np.abs(Tt.cpu().numpy() - Tw[0]).max() <= 1e-5 * np.abs(Tw).max(), plain


def test_insert_sorted_special_paths(oracle, dev, knob_env):
    """the brick-sorted insertion (thx_insert_sort.hip) off its common path: more unique shifts per image than a thread keeps
    ramps for (every draw its own shift: k_bin's member-by-member ramp sum), images spread over many chunks of the record
    buffer (THX_INSERT_SCRATCH_MB=1: one image's worst case per chunk) and a descriptor table that is full at once
    (THX_INSERT_SEG_CAP=8: workgroups insert their segments themselves).  All against the oracle, and the three runs bit for
    bit equal to each other -- every term is rounded once, to the session's quanta, whichever way it travels."""
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(4242)
    N, nImg, mReco, nK = 32, 7, 24, 2
    P = 2 * N
    ref, vol, pl, im, quat, tran, offS, w, cls = _insert_case(O, N, nImg, mReco, rng, nK)
    assert len({t.tobytes() for t in tran[0]}) == mReco > 16
    Fw, Tw, _ = _oracle_insert(O, P, N, pl, im, quat, tran, offS, w, cls, nK)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev))
    outs = []
    for knob, val in ((None, None), ("THX_INSERT_SCRATCH_MB", "1"), ("THX_INSERT_SEG_CAP", "8")):
        if knob:
            knob_env(knob, val)
        F = torch.zeros((nK, P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((nK, P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, T(im["dat"], dev), T(im["ctf"], dev), T(w, dev), rot, T(tran, dev), T(pl["iCol"], dev),
                   T(pl["iRow"], dev), 2, N, offS=T(offS, dev), cls=T(cls, dev), nK=nK)
        if knob:
            knob_env(knob, None)
        assert np.abs(F.cpu().numpy() - Fw).max() <= 1e-5 * np.abs(Fw).max(), knob
        assert np.abs(Tt.cpu().numpy() - Tw).max() <= 1e-5 * np.abs(Tw).max(), knob
        outs.append((F, Tt))
    for F, Tt in outs[1:]:
        assert torch.equal(F, outs[0][0]) and torch.equal(Tt, outs[0][1])


def test_insert_sorted_many_unrelated_groups(oracle, dev):
    """160 draws per image, every one its own rotation anywhere on the sphere: 20 passes of k_bin per region, each touching tens of
    bricks all over the volume -- the workgroup's descriptor list (2 048 entries) fills up and is flushed in the middle of its
    passes.  Against the oracle."""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(515)
    N, nImg, mReco = 64, 2, 160
    P = 2 * N
    ref, vol, pl, im, quat, tran, offS, w, cls = _insert_case(O, N, nImg, mReco, rng)
    quat = synth.random_quats(nImg * mReco, rng).reshape(nImg, mReco, 4)
    tran = np.repeat(tran[:, :4], mReco // 4, axis=1)                  # 4 unique shifts per image: the register-ramp path
    Fw, Tw, _ = _oracle_insert(O, P, N, pl, im, quat, tran, offS, w, np.zeros_like(cls), 1)
    F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    ops.insert(F, Tt, P, T(im["dat"], dev), T(im["ctf"], dev), T(w, dev), ops.rotmat(T(quat.reshape(-1, 4), dev)), T(tran, dev),
               T(pl["iCol"], dev), T(pl["iRow"], dev), 2, N, offS=T(offS, dev))
    assert np.abs(F.cpu().numpy() - Fw[0]).max() <= 1e-5 * np.abs(Fw).max()
    assert np.abs(Tt.cpu().numpy() - Tw[0]).max() <= 1e-5 * np.abs(Tw).max()
    np.testing.assert_allclose(Tt.sum(dtype=torch.float64).item(), Tw.sum(dtype=np.float64), rtol=1e-6)


def test_insert_linearity_and_csearch(oracle, dev):
    """size-independent properties: insert(a)+insert(b) == insert(a and b); cSearch with dfac == 1 equals the
    precomputed-CTF path to rounding of the on-device CTF"""
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(9)
    N, nImg, mReco = 32, 4, 5
    P = 2 * N
    ref, vol, pl, im, quat, tran, offS, w, cls = _insert_case(O, N, nImg, mReco, rng)
    iCol, iRow = T(pl["iCol"], dev), T(pl["iRow"], dev)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev)).reshape(nImg, mReco, 9)
    dat, ctf, wd, trd = T(im["dat"], dev), T(im["ctf"], dev), T(w, dev), T(tran, dev)

    def run(sel, **kw):
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, dat[sel].contiguous(), ctf[sel].contiguous(), wd[sel].contiguous(), rot[sel].contiguous(),
                   trd[sel].contiguous(), iCol, iRow, 2, N, **kw)
        return F, Tt
    Fa, Ta = run(slice(0, 2))
    Fb, Tb = run(slice(2, 4))
    Fab, Tab = run(slice(0, 4))
    assert (Fa + Fb - Fab).abs().max().item() <= 1e-5 * Fab.abs().max().item()
    assert (Ta + Tb - Tab).abs().max().item() <= 1e-5 * Tab.abs().max().item()
    attr = T(im["attr"], dev)
    dfac = torch.ones((nImg, mReco), dtype=torch.float64, device=dev)
    Fc, Tc = run(slice(0, 4), attr=attr, dfac=dfac, cSearch=True, pixelSize=1.32)
    assert (Fc - Fab).abs().max().item() <= 1e-3 * Fab.abs().max().item()


def test_prepareTF_bit_exact(oracle, dev):
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(21)
    N, P = 32, 64
    maxRadius = N // 2 - 2
    ref, vol, pl, im, quat, tran, offS, w, cls = _insert_case(O, N, 10, 6, rng)
    F, Tt, _ = _oracle_insert(O, P, N, pl, im, quat, tran, offS, w, cls, 1)
    F, Tt = F[0].copy(), Tt[0].copy()
    Fd, Td = T(F, dev), T(Tt, dev)
    O.normalise_TF(F, Tt, P)
    ops.normalise_TF(Fd, Td, P)
    assert_bit_equal(Fd.cpu().numpy(), F, "normalise F")
    assert_bit_equal(Td.cpu().numpy(), Tt, "normalise T")
    for nsym in (1, 4, 7):
        sym = synth.cn_symmetry(nsym)
        r = maxRadius * 2 + 1
        Fs = O.symmetrize(F, P, sym, r) if nsym > 1 else F
        Ts = O.symmetrize(Tt, P, sym, r) if nsym > 1 else Tt
        assert_bit_equal(ops.symmetrize(Fd, P, sym, r).cpu().numpy(), Fs, "symmetrize F C%d" % nsym)
        assert_bit_equal(ops.symmetrize(Td, P, sym, r).cpu().numpy(), Ts, "symmetrize T C%d" % nsym)


def _fsc_np(O, a, b, N):
    return O.fsc(sfft.rfftn(a).astype(np.complex64), sfft.rfftn(b).astype(np.complex64), N, N // 2)


@pytest.mark.parametrize("MAP,gridCorr,joinHalf,N", [(False, True, False, 32), (True, True, True, 32),
                                                     (True, False, False, 32), (False, False, False, 32),
                                                     (True, True, False, 24),    # N = 24: non-power-of-two grid (rocFFT)
                                                     (True, True, False, 64)])   # P = 128: radix-2 pre-stage + 8 x 8
def test_reconstruct(oracle, dev, MAP, gridCorr, joinHalf, N):
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(33)
    P = 2 * N
    maxRadius = N // 2 - 2
    ref, vol, pl = make_case(O, N)
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    Tt = np.zeros((P, P, P // 2 + 1), np.float32)
    ctf1 = np.ones(pl["nPxl"], np.float32)
    from thunder_amd import synth
    for q in synth.random_quats(400, rng):
        R = O.rotate3D(q)
        O.insertP(F, Tt, P, O.project(vol, P, 2, R, pl["iCol"], pl["iRow"]), ctf1, R, 1.0, pl["iColPad"], pl["iRowPad"])
    O.normalise_TF(F, Tt, P)
    fscv = np.clip(np.linspace(1.0, 0.05, N // 2), 0, 1).astype(np.float32)
    want, it_w, diffs, _ = O.reconstruct(F, Tt, P, N, 2, maxRadius, FSC=fscv, joinHalf=joinHalf, MAP=MAP,
                                         gridCorr=gridCorr, return_iters=True)
    plan = ops.RecoPlan(N, N, 2)
    got = plan.reconstruct(T(F, dev), T(Tt, dev), maxRadius, FSC=fscv, joinHalf=joinHalf, MAP=MAP,
                           gridCorr=gridCorr).cpu().numpy()
    if gridCorr:
        assert plan.last_iters == it_w
        assert abs(plan.last_diffC - diffs[-1]) <= 1e-3 * max(1.0, diffs[-1])
    # voxel-wise <= 1e-4 max|map| and FSC >= 0.9999 on every shell
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    assert _fsc_np(O, got, want, N)[: maxRadius].min() >= 0.9999
    plan.close()


def test_set_projectee_and_roundtrip(oracle, dev):
    """Projector::setProjectee parity, then the reference-sanctioned self-consistency check
    (thunder_project -> thunder_reconstruct) entirely on the device at N = 64."""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(44)
    N, P = 64, 128
    maxRadius = N // 2 - 2
    ref, vol_w, pl = make_case(O, N)
    plan = ops.RecoPlan(N, N, 2)
    vol = plan.set_projectee(T(ref, dev))
    assert (vol.cpu() - torch.from_numpy(vol_w)).abs().max().item() <= 2e-6 * np.abs(vol_w).max()
    nImg = 1500
    quat = synth.random_quats(nImg, rng)
    rot = ops.rotmat(T(quat, dev))
    iCol, iRow = T(pl["iCol"], dev), T(pl["iRow"], dev)
    slices = ops.project(vol, rot, iCol, iRow, 2)
    F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    ones = torch.ones((nImg, pl["nPxl"]), dtype=torch.float32, device=dev)
    w = torch.ones(nImg, dtype=torch.float32, device=dev)
    ops.insert(F, Tt, P, slices, ones, w, rot.reshape(nImg, 1, 9), torch.zeros((nImg, 1, 2), dtype=torch.float64, device=dev),
               iCol, iRow, 2, N)
    ops.normalise_TF(F, Tt, P)
    rec = plan.reconstruct(F, Tt, maxRadius, MAP=False, gridCorr=True).cpu().numpy()
    f = _fsc_np(O, rec, ref, N)
    assert f[:20].min() >= 0.995, f
    scale = float((rec * ref).sum() / (ref * ref).sum())
    assert 0.9 < scale < 1.1
    plan.close()


def test_fsc(oracle, dev):
    from thunder_amd import ops
    rng = np.random.default_rng(2)
    N = 32
    a = rng.normal(size=(N, N, N)).astype(np.float32)
    b = (a + 0.5 * rng.normal(size=(N, N, N))).astype(np.float32)
    A, B = sfft.rfftn(a).astype(np.complex64), sfft.rfftn(b).astype(np.complex64)
    want = oracle.fsc(A, B, N, N // 2)
    got = ops.fsc(T(A, dev), T(B, dev), N, N // 2).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-6)


def test_host_interface_entry_points(oracle, dev):
    """the Interface.h-shaped host-pointer entry points agree with the device path"""
    from thunder_amd import capi, ops, synth
    O = oracle
    rng = np.random.default_rng(8)
    N, P = 32, 64
    ref, vol, pl = make_case(O, N)
    mats = edge_rotations(rng, 4)
    out = np.zeros((len(mats), pl["nPxl"]), np.complex64)
    capi.call("thx_ExpectProject_host", vol.ctypes.data, out.ctypes.data, mats.ctypes.data, pl["iCol"].ctypes.data,
              pl["iRow"].ctypes.data, len(mats), 2, 1, P, pl["nPxl"])
    want = np.stack([O.project(vol, P, 2, m, pl["iCol"], pl["iRow"]) for m in mats])
    assert_bit_equal(out, want, "ExpectProject_host")
    # InsertFT -> PrepareTF -> reconstructG on host arrays
    nImg, mReco = 5, 4
    _, _, _, im, quat, tran, offS, w, cls = _insert_case(O, N, nImg, mReco, rng)
    Fw, Tw, _ = _oracle_insert(O, P, N, pl, im, quat, tran, offS, w, np.zeros_like(cls), 1)
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    Tc = np.zeros((P, P, P // 2 + 1), np.complex64)
    O3 = np.zeros(3)
    cnt = np.zeros(1, np.int32)
    quat_c, tran_c = np.ascontiguousarray(quat), np.ascontiguousarray(tran)
    capi.call("thx_InsertFT_host", F.ctypes.data, Tc.ctypes.data, O3.ctypes.data, cnt.ctypes.data, im["dat"].ctypes.data,
              im["ctf"].ctypes.data, None, offS.ctypes.data, w.ctypes.data, quat_c.ctypes.data, tran_c.ctypes.data, None,
              None, pl["iCol"].ctypes.data, pl["iRow"].ctypes.data, 1.32, 0, 2, pl["nPxl"], mReco, N, P, 1, nImg)
    assert np.abs(F - Fw[0]).max() <= 1e-5 * np.abs(Fw).max()
    assert np.abs(Tc.real - Tw[0]).max() <= 1e-5 * np.abs(Tw).max() and np.all(Tc.imag == 0)
    assert cnt[0] == nImg * mReco
    sym = synth.cn_symmetry(2)
    capi.call("thx_PrepareTF_host", 0, F.ctypes.data, Tc.ctypes.data, P, sym.ctypes.data, len(sym), N // 2 - 2, 2)
    Fo, To = Fw[0].copy(), Tw[0].copy()
    O.normalise_TF(Fo, To, P)
    Fo, To = O.symmetrize(Fo, P, sym, (N // 2 - 2) * 2 + 1), O.symmetrize(To, P, sym, (N // 2 - 2) * 2 + 1)
    assert np.abs(F - Fo).max() <= 2e-5 * np.abs(Fo).max()
    # reconstructG on a well-covered data set (20 sparse inserts leave the W iteration ill-conditioned: its output then
    # amplifies last-bit differences of the FFTs), identical host inputs on both sides
    Fo = np.zeros((P, P, P // 2 + 1), np.complex64)
    To = np.zeros((P, P, P // 2 + 1), np.float32)
    one = np.ones(pl["nPxl"], np.float32)
    for q in synth.random_quats(400, rng):
        R = O.rotate3D(q)
        O.insertP(Fo, To, P, O.project(vol, P, 2, R, pl["iCol"], pl["iRow"]), one, R, 1.0, pl["iColPad"], pl["iRowPad"])
    O.normalise_TF(Fo, To, P)
    Tin = np.zeros((P, P, P // 2 + 1), np.complex64)
    Tin.real = To
    dst = np.zeros((N, N, N), np.float32)
    capi.call("thx_ReconstructG_host", 0, Fo.ctypes.data, Tin.ctypes.data, N, N, 2, N // 2 - 2, 1.9, 15.0, None, 0, 0, 0,
              1, dst.ctypes.data)
    want = O.reconstruct(Fo, To, P, N, 2, N // 2 - 2, MAP=False, gridCorr=True)
    assert np.abs(dst - want).max() <= 1e-4 * np.abs(want).max()


def test_cpp_mirror_roundtrip(dev, tmp_path):
    """host code in the reference's own style (include/thunder_amd/Projector.hpp + Reconstructor.hpp over the C ABI),
    compiled with g++ and run as a separate process: thunder_project -> thunder_reconstruct round trip"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mirror_rt")
    libdir = os.path.join(root, "thunder_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "mirror_roundtrip.cpp"), "-o", exe, "-L" + libdir,
                           "-lthunder_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), (out.stdout, out.stderr[-2000:])


def test_insert_grouped_duplicates(oracle, dev):
    """draws that repeat rotations / shifts (a resampled particle filter) go through the insert plan's grouping:
    the result must equal the oracle's draw-by-draw insertion; with cSearch the defocus factor is part of the key"""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(123)
    N, nImg, mReco = 32, 4, 24
    P = 2 * N
    ref, vol, pl, im, quat, tran, offS, w, cls = _insert_case(O, N, nImg, mReco, rng, nK=2)
    # 5 distinct rotations, 3 distinct shifts per image, classes tied to the rotation index parity
    base_q = synth.perturb_quats(im["quat"], 5, 0.03, rng)
    base_t = im["shift"][:, None, :] + rng.normal(0, 0.4, size=(nImg, 3, 2))
    iR = rng.integers(0, 5, size=(nImg, mReco))
    iT = rng.integers(0, 3, size=(nImg, mReco))
    quat = np.take_along_axis(base_q, iR[:, :, None], axis=1)
    tran = np.take_along_axis(base_t, iT[:, :, None], axis=1)
    cls = (iR % 2).astype(np.int32)
    Fw, Tw, Ow = _oracle_insert(O, P, N, pl, im, quat, tran, offS, w, cls, 2)
    F = torch.zeros((2, P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((2, P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev))
    ops.insert(F, Tt, P, T(im["dat"], dev), T(im["ctf"], dev), T(w, dev), rot, T(tran, dev), T(pl["iCol"], dev),
               T(pl["iRow"], dev), 2, N, offS=T(offS, dev), cls=T(cls, dev), nK=2)
    assert np.abs(F.cpu().numpy() - Fw).max() <= 1e-5 * np.abs(Fw).max()
    assert np.abs(Tt.cpu().numpy() - Tw).max() <= 1e-5 * np.abs(Tw).max()
    # cSearch: same rotation, different defocus factors must NOT be merged
    dfac = 1.0 + 0.05 * rng.integers(0, 2, size=(nImg, mReco)).astype(np.float64)
    F2 = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    T2 = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    ops.insert(F2, T2, P, T(im["dat"], dev), T(im["ctf"], dev), T(w, dev), rot, T(tran, dev), T(pl["iCol"], dev),
               T(pl["iRow"], dev), 2, N, offS=T(offS, dev), attr=T(im["attr"], dev), dfac=T(dfac, dev), cSearch=True,
               pixelSize=1.32)
    Fo = np.zeros((P, P, P // 2 + 1), np.complex64)
    To = np.zeros((P, P, P // 2 + 1), np.float32)
    for l in range(nImg):
        for m in range(mReco):
            a = im["attr"][l]
            c = O.ctf(1.32, a[0], np.float32(a[1] * dfac[l, m]), np.float32(a[2] * dfac[l, m]), *a[3:], N, pl["iCol"], pl["iRow"])
            t = tran[l, m] - offS[l]
            src = O.translate(np.float32(-t[0]), np.float32(-t[1]), N, pl["iCol"], pl["iRow"], src=im["dat"][l])
            O.insertP(Fo, To, P, src, c, O.rotate3D(quat[l, m]), w[l], pl["iColPad"], pl["iRowPad"])
    assert np.abs(F2.cpu().numpy() - Fo).max() <= 2e-4 * np.abs(Fo).max()   # on-device CTF: chi rounding (see CTF test)
    assert np.abs(T2.cpu().numpy() - To).max() <= 2e-4 * np.abs(To).max()


def test_full_size_properties_n256(oracle, dev):
    """BASELINE box size (256^3, P = 512, nPxl = 24747): slices bit-exact against the oracle, log-likelihoods within the
    stated bar, and the size-independent properties of insertion (mass conservation, linearity)"""
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list
    O = oracle
    rng = np.random.default_rng(256)
    N, P = 256, 512
    pl = pixel_list(N, N // 2 - 2, 0)
    assert pl["nPxl"] == 24747
    ref = synth.blob_map(N, nblob=8)
    plan = ops.RecoPlan(N, N, 2)
    vol = plan.set_projectee(T(ref, dev))
    vol_h = vol.cpu().numpy()
    iCol, iRow = T(pl["iCol"], dev), T(pl["iRow"], dev)
    quat = synth.perturb_quats(synth.random_quats(1, rng), 6, 0.01, rng)[0]
    mats = np.stack([O.rotate3D(q) for q in quat])
    sl = ops.project(vol, T(mats, dev), iCol, iRow, 2).cpu().numpy()
    want = np.stack([O.project(vol_h, P, 2, m, pl["iCol"], pl["iRow"]) for m in mats])
    assert_bit_equal(sl, want, "project at N=256")
    # one image = slice 0 + noise; 6 rotations x 3 shifts
    dat = (sl[0] + 0.5 * np.abs(sl[0]).mean() * (rng.normal(size=sl[0].shape) + 1j * rng.normal(size=sl[0].shape))).astype(np.complex64)
    ctf = np.ones(pl["nPxl"], np.float32)
    sig = np.full(pl["nPxl"], -0.5 / float(np.mean(np.abs(dat) ** 2)), np.float32)
    tran = np.array([[0.0, 0.0], [0.4, -0.3], [-1.0, 0.7]])
    res = ops.expect_local(vol, P, 2, N, iCol, iRow, T(dat[None], dev), T(ctf[None], dev), T(sig[None], dev),
                           T(mats[None], dev), T(tran[None], dev), want_logW=True)
    w = O.expect_local(vol_h, P, 2, N, pl["iCol"], pl["iRow"], dat, ctf, sig, mats, tran)
    wl = w["logW"][:, :, 0].T
    exact = np.array([[O.logDataVSPrior_f64(dat, O.translate(np.float32(t[0]), np.float32(t[1]), N, pl["iCol"], pl["iRow"]) * s_, ctf, sig)
                       for s_ in want] for t in tran])
    got = res.logW[0, 0].cpu().numpy()
    # float sums of ~25k terms: the error scale is eps * sum|terms| ~ eps * |C|, C = sum sigRcp |dat|^2 (the expanded form
    # of the device kernel carries C explicitly; at cryo-EM SNR |L| ~ |C|).  Bar: 5e-7 |C| on L, and -- what the particle
    # filter consumes -- the DIFFERENCES L - max L to 2e-7 |C| (C is kept out of the exponent on the device).
    Cabs = abs(float(np.sum(sig.astype(np.float64) * np.abs(dat.astype(np.complex128)) ** 2)))
    assert np.abs(got - exact).max() <= 5e-7 * Cabs
    assert np.abs(wl - exact).max() <= 5e-7 * Cabs      # the oracle's (= reference's) own float sum meets the same bar
    dg, de = got - got.max(), exact - exact.max()
    assert np.abs(dg - de).max() <= 5e-7 * Cabs
    wRx = np.exp(de).sum(axis=0)
    np.testing.assert_allclose(res.wR[0].cpu().numpy(), wRx, rtol=max(2e-6 * Cabs, 1e-4))
    # the true pose is (one of) the best: within the float-sum tolerance of the maximum
    wRg, wTg = res.wR[0].cpu().numpy(), res.wT[0].cpu().numpy()
    assert wRg[0] >= (1 - max(2e-6 * Cabs, 1e-4)) * wRg.max() * 0.99 and wTg[0] >= 0.99 * wTg.max()
    # insertion: mass conservation (every in-grid sample adds weights summing to 1) and linearity
    nImg, mReco = 3, 10
    q2 = synth.perturb_quats(synth.random_quats(nImg, rng), mReco, 0.01, rng)
    rot = ops.rotmat(T(q2.reshape(-1, 4), dev)).reshape(nImg, mReco, 9)
    ones_c = torch.ones((nImg, pl["nPxl"]), dtype=torch.complex64, device=dev)
    ones_f = torch.ones((nImg, pl["nPxl"]), dtype=torch.float32, device=dev)
    wgt = torch.full((nImg,), 0.5, dtype=torch.float32, device=dev)
    tr0 = torch.zeros((nImg, mReco, 2), dtype=torch.float64, device=dev)

    def run(sel):
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, ones_c[sel].contiguous(), ones_f[sel].contiguous(), wgt[sel].contiguous(), rot[sel].contiguous(),
                   tr0[sel].contiguous(), iCol, iRow, 2, N)
        return F, Tt
    Fa, Ta = run(slice(0, 1))
    Fb, Tb = run(slice(1, 3))
    Fab, Tab = run(slice(0, 3))
    assert abs(Tab.sum(dtype=torch.float64).item() - 0.5 * nImg * mReco * pl["nPxl"]) <= 1e-4 * 0.5 * nImg * mReco * pl["nPxl"]
    assert (Ta + Tb - Tab).abs().max().item() <= 1e-5 * Tab.abs().max().item()
    assert (Fa + Fb - Fab).abs().max().item() <= 1e-5 * Fab.abs().max().item()
    assert abs(Fab.real.sum(dtype=torch.float64).item() - Tab.sum(dtype=torch.float64).item()) <= 1e-4 * Tab.sum().item()
    plan.close()


def test_insert_then_reconstruct_matches_float_path(dev, knob_env):
    """The gridding reconstruction is ill-conditioned where coverage is thin, so the bar on the inserted F/T (1e-5 max)
    does not by itself bound the map.  The LDS-brick kernel (fixed-point accumulation, tiny terms routed as floats) must
    give the same MAP-off, grid-corrected map as the plain float-atomic kernel (the reference's own arithmetic)."""
    from thunder_amd import ops
    from thunder_amd.refine import RefineShard
    sh = RefineShard(64, 1200, dev)
    wR, wT = sh.expectation(0)
    rot, tran = sh.draw_reco(0, wR, wT)
    maps = {}
    for plain in ("1", "0"):
        knob_env("THX_INSERT_PLAIN", plain)
        sh.insertion(0, rot, tran)
        ops.normalise_TF(sh.F[0], sh.T[0], sh.P)
        maps[plain] = sh.plans[0].reconstruct(sh.F[0].clone(), sh.T[0].clone(), sh.maxRadius, MAP=False, gridCorr=True)
    d = (maps["0"] - maps["1"]).abs().max().item() / maps["1"].abs().max().item()
    assert d <= 5e-4, d


def test_expect_local_packed_projector_is_bit_identical(oracle, dev):
    """the cell-packed projector layout changes where the 8 neighbours are read from, not what is computed"""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(77)
    N, P, nImg, nR, nT = 32, 64, 5, 70, 9
    ref, vol, pl = make_case(O, N, rL=2)
    im = make_images(O, vol, pl, N, nImg, rng)
    q = synth.perturb_quats(im["quat"], nR, 0.05, rng)
    rot = ops.rotmat(T(q.reshape(-1, 4), dev)).reshape(nImg, nR, 9)
    tr = T(im["shift"][:, None, :] + rng.normal(0, 0.7, size=(nImg, nT, 2)), dev)
    v2 = torch.stack([T(vol, dev), T(vol[::-1].copy(), dev)]).contiguous()          # two "classes"
    volIdx = T(np.array([0, 1, 1, 0, 1], np.int32), dev)
    args = (P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(im["dat"], dev), T(im["ctf"], dev), T(im["sigRcp"], dev), rot, tr)
    a = ops.expect_local(v2, *args, volIdx=volIdx, want_logW=True)
    cells = ops.pack_projector(v2, P)
    assert cells.shape == (2, P, P, P // 2 + 1, 16)
    b = ops.expect_local(cells, *args, volIdx=volIdx, want_logW=True, packed=True)
    for k in ("wR", "wT", "wC", "baseLine", "logW"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


def test_expect_local_split_form_is_bit_identical(oracle, dev, knob_env):
    """THX_EXPECT_SPLIT = m: the near-slab / tail form of the local-search kernel (samples within m voxels of the wave's mean slab
    are requested one pixel ahead, the rest one at a time) reads the same cells and accumulates in the same order: every weight,
    the base line and every log-likelihood are bit-identical to the default form -- for clouds that sit inside the slab, straddle
    it, and lie outside it altogether"""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(78)
    N, P, nImg, nR, nT = 64, 128, 6, 125, 9
    ref, vol, pl = make_case(O, N, rL=2)
    im = make_images(O, vol, pl, N, nImg, rng)
    spread = np.array([0.005, 0.02, 0.05, 0.2, 1.0, 0.03])
    q = np.stack([synth.perturb_quats(im["quat"][l:l + 1], nR, spread[l], rng)[0] for l in range(nImg)])
    rot = ops.rotmat(T(q.reshape(-1, 4), dev)).reshape(nImg, nR, 9)
    tr = T(im["shift"][:, None, :] + rng.normal(0, 0.7, size=(nImg, nT, 2)), dev)
    cells = ops.pack_projector(T(vol, dev)[None].contiguous(), P)
    args = (P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(im["dat"], dev), T(im["ctf"], dev), T(im["sigRcp"], dev), rot, tr)
    a = ops.expect_local(cells, *args, want_logW=True, packed=True)
    for m in ("1.5", "6", "1000"):
        knob_env("THX_EXPECT_SPLIT", m)
        b = ops.expect_local(cells, *args, want_logW=True, packed=True)
        for k in ("wR", "wT", "wC", "baseLine", "logW"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (m, k)
    knob_env("THX_EXPECT_SPLIT", None)


def test_expect_local_cloud_order_is_bit_identical(oracle, dev, knob_env):
    """THX_EXPECT_ORDER: lane <-> rotation of the local-search kernel follows a ranking of every image's cloud (k_cloud_order: by
    the in-plane angle -- the default -- or one of the two tilt components); results go back to the rotation's own place, so every
    weight and every log-likelihood is bit-identical to the storage order (0), for tight clouds, wide ones, a cloud of identical
    rotations (all keys tie) and 70 / 125 / 200 rotations (one partial wave, two, four)"""
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(79)
    N, P, nImg, nT = 64, 128, 6, 9
    ref, vol, pl = make_case(O, N, rL=2)
    im = make_images(O, vol, pl, N, nImg, rng)
    spread = np.array([0.005, 0.02, 0.05, 0.2, 1.0, 0.0])
    cells = ops.pack_projector(T(vol, dev)[None].contiguous(), P)
    for nR in (70, 125, 200):
        q = np.stack([synth.perturb_quats(im["quat"][l:l + 1], nR, spread[l], rng)[0] for l in range(nImg)])
        rot = ops.rotmat(T(q.reshape(-1, 4), dev)).reshape(nImg, nR, 9)
        tr = T(im["shift"][:, None, :] + rng.normal(0, 0.7, size=(nImg, nT, 2)), dev)
        args = (P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(im["dat"], dev), T(im["ctf"], dev), T(im["sigRcp"], dev), rot, tr)
        knob_env("THX_EXPECT_ORDER", "0")
        a = ops.expect_local(cells, *args, want_logW=True, packed=True)
        for m in ("1", "2", "3"):
            knob_env("THX_EXPECT_ORDER", m)
            b = ops.expect_local(cells, *args, want_logW=True, packed=True)
            for k in ("wR", "wT", "wC", "baseLine", "logW"):
                assert torch.equal(getattr(a, k), getattr(b, k)), (nR, m, k)
    knob_env("THX_EXPECT_ORDER", None)


def test_full_size_properties_n512(oracle, dev):
    """BASELINE config (4): 512^3 box (P = 1024, nPxl = 100941; 4.3 GB projector, 34 GB cell-packed, element offsets
    above 2^32).  Slices bit-exact against the oracle, packed == unpacked E-step, insertion mass / linearity, and a
    project -> insert -> reconstruct round trip (FSC against the input map)."""
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list
    O = oracle
    rng = np.random.default_rng(512)
    N, P = 512, 1024
    pl = pixel_list(N, N // 2 - 2, 0)
    assert pl["nPxl"] == 100941
    ref = synth.blob_map(N, nblob=10)
    plan = ops.RecoPlan(N, N, 2)
    vol = plan.set_projectee(T(ref, dev))
    iCol, iRow = T(pl["iCol"], dev), T(pl["iRow"], dev)
    # ---- slices vs the oracle (rotations chosen to reach the far corner of the half grid: k, j < 0 wrap) ----
    quat = np.concatenate([synth.random_quats(2, rng), np.array([[1.0, 0, 0, 0]])])
    mats = np.stack([O.rotate3D(q) for q in quat])
    sl = ops.project(vol, T(mats, dev), iCol, iRow, 2).cpu().numpy()
    vol_h = vol.cpu().numpy()
    want = np.stack([O.project(vol_h, P, 2, m, pl["iCol"], pl["iRow"]) for m in mats])
    del vol_h
    assert_bit_equal(sl, want, "project at N=512")
    # ---- E-step: cell-packed (offsets up to 8.6e9 floats) == standard layout, bit for bit ----
    nR, nT = 12, 3
    q = synth.perturb_quats(quat[:2], nR, 0.01, rng)
    rot = ops.rotmat(T(q.reshape(-1, 4), dev)).reshape(2, nR, 9)
    dat = sl[:2] + (0.5 * np.abs(sl[:2]).mean() * (rng.normal(size=sl[:2].shape) + 1j * rng.normal(size=sl[:2].shape)))
    dat = dat.astype(np.complex64)
    ctf = np.ones((2, pl["nPxl"]), np.float32)
    sig = np.full((2, pl["nPxl"]), -0.5 / float(np.mean(np.abs(dat) ** 2)), np.float32)
    tr = T(rng.normal(0, 0.7, size=(2, nT, 2)), dev)
    args = (P, 2, N, iCol, iRow, T(dat, dev), T(ctf, dev), T(sig, dev), rot, tr)
    a = ops.expect_local(vol, *args, want_logW=True)
    cells = ops.pack_projector(vol, P)
    b = ops.expect_local(cells, *args, want_logW=True, packed=True)
    del cells
    for k in ("wR", "wT", "wC", "baseLine", "logW"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert int(a.wR[0].argmax()) < nR and torch.isfinite(a.logW).all()
    # ---- insertion: mass conservation and linearity on the 1024^3 half grid ----
    nImg, mReco = 3, 6
    q2 = synth.perturb_quats(synth.random_quats(nImg, rng), mReco, 0.01, rng)
    rot2 = ops.rotmat(T(q2.reshape(-1, 4), dev)).reshape(nImg, mReco, 9)
    ones_c = torch.ones((nImg, pl["nPxl"]), dtype=torch.complex64, device=dev)
    ones_f = torch.ones((nImg, pl["nPxl"]), dtype=torch.float32, device=dev)
    wgt = torch.full((nImg,), 0.5, dtype=torch.float32, device=dev)
    tr0 = torch.zeros((nImg, mReco, 2), dtype=torch.float64, device=dev)

    def run(sel):
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, ones_c[sel].contiguous(), ones_f[sel].contiguous(), wgt[sel].contiguous(), rot2[sel].contiguous(),
                   tr0[sel].contiguous(), iCol, iRow, 2, N)
        return F, Tt
    Fa, Ta = run(slice(0, 1))
    Fb, Tb = run(slice(1, 3))
    Fab, Tab = run(slice(0, 3))
    mass = 0.5 * nImg * mReco * pl["nPxl"]
    assert abs(Tab.sum(dtype=torch.float64).item() - mass) <= 1e-4 * mass
    assert (Ta + Tb - Tab).abs().max().item() <= 1e-5 * Tab.abs().max().item()
    assert (Fa + Fb - Fab).abs().max().item() <= 1e-5 * Fab.abs().max().item()
    del Fa, Ta, Fb, Tb, Fab, Tab
    # ---- round trip: 1500 noiseless slices -> insert -> normalise -> reconstruct; FSC against the input ----
    nS = 1500
    qs = synth.random_quats(nS, rng)
    rots = ops.rotmat(T(qs, dev)).reshape(nS, 1, 9)
    F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
    one_f = torch.ones((100, pl["nPxl"]), dtype=torch.float32, device=dev)
    w1 = torch.ones((100,), dtype=torch.float32, device=dev)
    t1 = torch.zeros((100, 1, 2), dtype=torch.float64, device=dev)
    for s0 in range(0, nS, 100):
        slices = ops.project(vol, rots[s0:s0 + 100].reshape(100, 9).contiguous(), iCol, iRow, 2)
        ops.insert(F, Tt, P, slices, one_f, w1, rots[s0:s0 + 100].contiguous(), t1, iCol, iRow, 2, N)
    ops.normalise_TF(F, Tt, P)
    got = plan.reconstruct(F, Tt, N // 2 - 2, MAP=False, gridCorr=True)
    assert 1 <= plan.last_iters <= 30 and torch.isfinite(got).all()
    A = ops.fft3d_fw(got.contiguous())
    B = ops.fft3d_fw(T(ref, dev))
    fsc = ops.fsc(A, B, N, N // 2 - 2).cpu().numpy()
    # 1500 slices sample shell r with ~1500 * 2 pi r / (4 pi r^2) = 750 / r samples per voxel; the blob map's power
    # falls steeply with r, so the (fixed) interpolation error weighs more from shell to shell
    assert fsc[1:32].min() >= 0.999 and fsc[1:64].min() >= 0.97, fsc[:64]
    plan.close()


@pytest.mark.parametrize("N", [64, 128, 256, 512])
def test_hand_fft_passes_match_rocfft(dev, knob_env, N):
    """The gridding iteration on power-of-two grids (P = 128 ... 1024 here; the BASELINE grid is P = 512) runs on the
    hand-written FFT passes of thx_fft8.h (strided radix-8 passes with a radix-2 / 4 pre-stage where needed, x transform
    fused with the kernel multiply, z transform fused with the weight update); at P = 64 and 128 the same code is checked
    against the oracle by test_reconstruct.  Here: identical inputs through THX_FFT=rocfft (library
    transforms + separate elementwise kernels) and the default path must give the same number of rounds, the same diffC
    and the same map.  The inputs are analytic (T = a smooth sampling density, F = reference x T), so that the run is
    deterministic and well conditioned: with sparse inserted data the max-norm stop rule amplifies last-bit differences
    (of the FFTs, or of the atomics' order from run to run) into different round counts -- for rocFFT against itself too."""
    from thunder_amd import ops, synth
    P = 2 * N
    plan = ops.RecoPlan(N, N, 2)
    vol = plan.set_projectee(T(synth.blob_map(N, nblob=12), dev))
    ax = torch.fft.fftfreq(P, d=1.0 / P, device=dev)
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :P // 2 + 1] ** 2)
    Tt = (1.0 / (1.0 + r / 8.0)).to(torch.float32).contiguous()        # ~ the 1/r density of randomly oriented slices
    Tt[r >= (N // 2 - 2) * 2 + 1] = 0
    F = (vol * Tt).contiguous()
    fscv = np.clip(np.linspace(1.0, 0.1, N // 2), 0, 1).astype(np.float32)
    out = {}
    for mode in ("rocfft", "hand", "hand_natural", "hand_hoststop") + (("hand_x2",) if P == 1024 else ()):
        knob_env("THX_FFTZ_WAVES", {"hand_x2": "16", "hand": "4" if P == 1024 else None}.get(mode))
        knob_env("THX_FFT", "rocfft" if mode == "rocfft" else None)
        knob_env("THX_RECO_WT", "natural" if mode == "hand_natural" else None)
        knob_env("THX_RECO_STOP", "host" if mode == "hand_hoststop" else None)
        m = plan.reconstruct(F.clone(), Tt.clone(), N // 2 - 2, FSC=fscv, MAP=True, gridCorr=True)
        out[mode] = (m, plan.last_iters, plan.last_diffC)
    (ma, ia, da), (mb, ib, db) = out["rocfft"], out["hand"]
    assert ia == ib and abs(da - db) <= 1e-3 * max(1.0, da), (ia, ib, da, db)
    assert (ma - mb).abs().max().item() <= 1e-4 * ma.abs().max().item()
    # W / T tiled by z column for the loop (the default) or in the volume's layout: the same arithmetic, bit for bit
    mc, ic, dc = out["hand_natural"]
    assert ic == ib and dc == db and torch.equal(mc, mb)
    # the stop rule evaluated on the device (default: every round queued, the rounds after the stop fall through) or on the host
    # with one read-back per round: the same round count, the same bits
    md, id_, dd = out["hand_hoststop"]
    assert id_ == ib and dd == db and torch.equal(md, mb)
    knob_env("THX_RECO_STOP", None)
    if P == 1024:   # the two forms of the z pass at P = 1024: eight points per thread in 1 024-thread workgroups ("hand" above, forced), sixteen in 512-thread ones
        me, ie, de = out["hand_x2"]
        assert ie == ib and de == db and torch.equal(me, mb)
    knob_env("THX_FFTZ_WAVES", None)
    plan.close()


def test_expect_local_fused_defocus_search(oracle, dev, knob_env):
    """CTF search at the reference's sizes (mLR 125, mLT 9, mLD 9): the fused kernel (one gather serves the 9 defocus
    factors) against one sweep per factor (THX_EXPECT_ND=sweep) and, for one image, against the oracle"""
    import time
    from thunder_amd import ops, synth
    O = oracle
    rng = np.random.default_rng(91)
    N, nR, nT, nD, nImg = 64, 125, 9, 9, 6
    ref, vol, pl = make_case(O, N, rL=2)
    P = 2 * N
    im = make_images(O, vol, pl, N, nImg, rng)
    rot_h = np.stack([[O.rotate3D(q) for q in qs] for qs in synth.perturb_quats(im["quat"], nR, 0.03, rng)])
    tran_h = im["shift"][:, None, :] + rng.normal(0, 0.5, size=(nImg, nT, 2))
    dfac = 1.0 + rng.normal(0, 0.02, size=(nImg, nD))
    ctfD = np.stack([[O.ctf(1.32, a[0], np.float32(a[1] * d), np.float32(a[2] * d), *a[3:], N, pl["iCol"], pl["iRow"])
                      for d in dfac[l]] for l, a in enumerate(im["attr"])])  # [nImg][nD][nPxl]
    pD = rng.uniform(0.5, 1.5, size=(nImg, nD))
    pR = rng.uniform(0.5, 1.5, size=(nImg, nR))
    args = (T(vol, dev), P, 2, N, T(pl["iCol"], dev), T(pl["iRow"], dev), T(im["dat"], dev), T(ctfD, dev), T(im["sigRcp"], dev),
            T(rot_h, dev), T(tran_h, dev))
    out = {}
    for mode in ("sweep", "fused"):
        knob_env("THX_EXPECT_ND", "sweep" if mode == "sweep" else None)
        ops.expect_local(*args, nD=nD, pD=T(pD, dev), pR=T(pR, dev), want_logW=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out[mode] = ops.expect_local(*args, nD=nD, pD=T(pD, dev), pR=T(pR, dev), want_logW=True)
        torch.cuda.synchronize()
        print("expect_local nD = 9, %s: %.3f ms" % (mode, (time.perf_counter() - t0) * 1e3))
    a, b = out["sweep"], out["fused"]
    scale = a.logW.abs().max().item()
    assert (a.logW - b.logW).abs().max().item() <= 1e-5 * scale
    for k in ("wR", "wT", "wD", "wC"):
        np.testing.assert_allclose(getattr(b, k).cpu().numpy(), getattr(a, k).cpu().numpy(), rtol=max(3e-5 * scale, 3e-4), err_msg=k)
    want = O.expect_local(vol, P, 2, N, pl["iCol"], pl["iRow"], im["dat"][0], ctfD[0], im["sigRcp"][0], rot_h[0], tran_h[0],
                          nD=nD, pD=pD[0], pR=pR[0], cSearch=True)
    wl = np.transpose(want["logW"], (2, 1, 0))  # [nD][nT][nR]
    np.testing.assert_allclose(b.logW[0].cpu().numpy(), wl, rtol=0, atol=1e-5 * np.abs(wl).max())
    for k in ("wR", "wT", "wD", "wC"):
        np.testing.assert_allclose(getattr(b, k)[0].cpu().numpy().reshape(-1), want[k].reshape(-1), rtol=2e-3, err_msg=k)
