"""CPU tests of the oracle's re-centre / re-mask / sigma-update restatements:
independent numpy cross-checks and invariants (the oracle is PARITY-UNPINNED)."""
import numpy as np
import scipy.fft as sfft

from _next_util import full_images
from _util import make_case


def test_soft_mask(oracle):
    O = oracle
    N, r, ew = 32, 9.0, 6.0
    m = O.soft_mask(N, r, ew)
    jj, ii = np.meshgrid(np.fft.fftfreq(N, 1.0 / N), np.fft.fftfreq(N, 1.0 / N), indexing="ij")
    u = np.hypot(ii, jj)
    assert np.all(m[u < r] == 1) and np.all(m[u > r + ew] == 0)
    edge = (u >= r) & (u <= r + ew)
    want = 0.5 + 0.5 * np.cos((u[edge] - r) / ew * np.pi)
    assert np.abs(m[edge] - want).max() < 1e-6
    assert np.array_equal(m, m.T)


def test_remask_matches_numpy_double(oracle):
    O = oracle
    rng = np.random.default_rng(3)
    N = 32
    rl = rng.standard_normal((3, N, N)).astype(np.float32)
    ft = sfft.rfft2(rl).astype(np.complex64)
    out = O.remask(ft, 13.2, 1.32, 6.0)       # radius 10 px
    want = np.fft.rfft2(rl.astype(np.float64) * O.soft_mask(N, 10.0, 6.0))
    assert np.abs(out - want).max() <= 2e-6 * np.abs(want).max()


def test_translate_image_is_a_shift(oracle):
    O = oracle
    rng = np.random.default_rng(4)
    N = 16
    rl = rng.standard_normal((N, N)).astype(np.float32)
    ft = sfft.rfft2(rl).astype(np.complex64)
    out = O.translate_image(ft, 3.0, -2.0)     # tx along columns (i), ty along rows (j)
    back = sfft.irfft2(out, s=(N, N))
    assert np.abs(back - np.roll(rl, (-2, 3), axis=(0, 1))).max() < 2e-5
    # radius form leaves pixels outside r untouched
    out_r = O.translate_image(ft, 3.0, -2.0, r=5.0)
    j = np.fft.fftfreq(N, 1.0 / N)[:, None]
    i = np.arange(N // 2 + 1)[None, :]
    inside = (i * i + j * j) < 25
    assert np.array_equal(out_r[~inside], ft[~inside]) and np.array_equal(out_r[inside], out[inside])


def test_translate_volume_is_a_shift(oracle):
    O = oracle
    rng = np.random.default_rng(5)
    P = 12
    rl = rng.standard_normal((P, P, P)).astype(np.float32)
    ft = sfft.rfftn(rl).astype(np.complex64)
    out = O.translate_volume(ft, 100.0, 1.0, 2.0, -3.0)
    back = sfft.irfftn(out, s=(P, P, P))
    assert np.abs(back - np.roll(rl, (-3, 2, 1), axis=(0, 1, 2))).max() < 5e-5
    out_r = O.translate_volume(ft, 4.0, 1.0, 2.0, -3.0)
    k = np.fft.fftfreq(P, 1.0 / P)[:, None, None]
    j = np.fft.fftfreq(P, 1.0 / P)[None, :, None]
    i = np.arange(P // 2 + 1)[None, None, :]
    inside = (i * i + j * j + k * k) < 16
    assert np.array_equal(out_r[~inside], ft[~inside]) and np.array_equal(out_r[inside], out[inside])


def test_disc_list_and_power_spectrum(oracle):
    O = oracle
    N, r = 32, 11
    dl = O.disc_list(N, r)
    brute = [(i, j) for j in range(-r, r) for i in range(0, r + 1) if i * i + j * j < r * r]
    assert dl["nPxl"] == len(brute)
    assert np.array_equal(np.stack([dl["iCol"], dl["iRow"]], 1), np.array(brute))
    assert np.any((dl["iCol"] == 0) & (dl["iRow"] < 0))        # unlike allocPreCalIdx, (0, j<0) is kept
    rng = np.random.default_rng(6)
    img = (rng.standard_normal((N, N // 2 + 1)) + 1j * rng.standard_normal((N, N // 2 + 1))).astype(np.complex64)
    ps = O.power_spectrum(img, r)
    keep = dl["iSig"] < r
    a2 = np.abs(img.reshape(-1)[dl["iPxl"]].astype(np.complex128)) ** 2
    want = np.bincount(dl["iSig"][keep], a2[keep], r) / np.bincount(dl["iSig"][keep], minlength=r)
    assert np.allclose(ps, want, rtol=2e-6)


def test_sigma_image_invariants(oracle):
    """noise-free image == CTF x top-pose slice x ramp  ->  residual spectrum 0 and model spectrum == data spectrum"""
    O = oracle
    rng = np.random.default_rng(7)
    N, P, projR, rSig = 32, 64, 13, 15
    _, vol, _ = make_case(O, N)
    im = full_images(O, vol, N, 2, rng, projR)
    l = 0
    dl = O.disc_list(N, projR)
    clean = np.zeros((N, N // 2 + 1), np.complex64)
    s = O.project(vol, P, 2, im["rot"][l], dl["iCol"], dl["iRow"])
    ramp = O.translate(im["tran"][l, 0], im["tran"][l, 1], N, dl["iCol"], dl["iRow"])
    c = O.ctf(im["pixelSize"], *im["attr"][l], N, dl["iCol"], dl["iRow"])
    clean.reshape(-1)[dl["iPxl"]] = (s * ramp) * c
    spec = O.sigma_image(vol, P, 2, N, projR, rSig, im["rot"][l], im["tran"][l], None, im["pixelSize"], im["attr"][l],
                         clean, clean)
    # (numpy's complex64 product rounds differently from the reference's Complex operator*: residual ~ 1 ulp^2)
    assert np.all(spec[2] <= 1e-9 * spec[1]) and np.all(spec[3] <= 1e-9 * spec[1])
    assert np.allclose(spec[0], spec[1], rtol=1e-5)
    # beyond projR the model is zero, so the residual spectrum is the data spectrum there
    spec = O.sigma_image(vol, P, 2, N, projR, rSig, im["rot"][l], im["tran"][l], im["offset"][l], im["pixelSize"],
                         im["attr"][l], im["img"][l], im["imgOri"][l])
    assert np.array_equal(spec[2][projR + 1:], spec[1][projR + 1:])
    assert np.all(spec[0][projR + 1:] == 0) and np.all(spec[2] > 0)
    # the noise in imgOri is 1.1x the noise of img and the signals cancel in both
    assert np.allclose(spec[3][2:projR - 1], 1.21 * spec[2][2:projR - 1], rtol=1e-3)


def test_norm_correction_oracle(oracle):
    """the oracle's restatement of Optimiser::normCorrection (src/Optimiser.cpp:6201-6394): a noise-free image has no residual;
    the norm over the ring is the sum of the residual shell spectra times their pixel counts; the median is GSL's; after the
    rescaling every image's norm is the median (to rounding)"""
    O = oracle
    rng = np.random.default_rng(8)
    N, P, projR, rL, rNorm = 32, 64, 13, 2.0, 10.0
    _, vol, _ = make_case(O, N)
    im = full_images(O, vol, N, 6, rng, projR)
    dl = O.disc_list(N, projR)
    clean = np.zeros((N, N // 2 + 1), np.complex64)
    s = O.project(vol, P, 2, im["rot"][0], dl["iCol"], dl["iRow"])
    clean.reshape(-1)[dl["iPxl"]] = (s * O.translate(im["tran"][0, 0], im["tran"][0, 1], N, dl["iCol"], dl["iRow"])) * \
        O.ctf(im["pixelSize"], *im["attr"][0], N, dl["iCol"], dl["iRow"])
    ref_power = O.norm_residual(vol, P, 2, N, projR, rL, rNorm, im["rot"][0], im["tran"][0], im["pixelSize"], im["attr"][0], 0 * clean)
    assert O.norm_residual(vol, P, 2, N, projR, rL, rNorm, im["rot"][0], im["tran"][0], im["pixelSize"], im["attr"][0], clean) <= 1e-9 * ref_power
    norm = np.array([O.norm_residual(vol, P, 2, N, projR, rL, rNorm, im["rot"][l], im["tran"][l], im["pixelSize"], im["attr"][l],
                                     im["img"][l]) for l in range(6)], np.float32)
    # direct numpy restatement on image 1: |img - model|^2 over rL^2 <= i^2 + j^2 < rNorm^2
    l = 1
    model = np.zeros((N, N // 2 + 1), np.complex64)
    sl = O.project(vol, P, 2, im["rot"][l], dl["iCol"], dl["iRow"])
    model.reshape(-1)[dl["iPxl"]] = (sl * O.translate(im["tran"][l, 0], im["tran"][l, 1], N, dl["iCol"], dl["iRow"])) * \
        O.ctf(im["pixelSize"], *im["attr"][l], N, dl["iCol"], dl["iRow"])
    jj = np.fft.fftfreq(N, 1.0 / N)[:, None]
    ii = np.arange(N // 2 + 1)[None, :]
    ring = (ii * ii + jj * jj >= rL * rL) & (ii * ii + jj * jj < rNorm * rNorm)
    assert np.isclose(norm[l], (np.abs(im["img"][l] - model) ** 2)[ring].sum(dtype=np.float64), rtol=1e-5)
    # the median: even and odd counts against numpy (linear interpolation at 0.5 (n - 1))
    for n in (6, 5, 1):
        assert np.isclose(O.median(norm[:n]), np.quantile(norm[:n].astype(np.float64), 0.5), rtol=1e-7)
    m = O.median(norm)
    img2, ori2 = O.norm_scale(im["img"], im["imgOri"], norm, m)
    assert np.allclose(np.abs(img2[3]) ** 2, np.abs(im["img"][3]) ** 2 * (m / norm[3]), rtol=1e-5)
    assert np.array_equal(img2[2] == 0, im["img"][2] == 0) and ori2.dtype == np.complex64


def test_sigma_accum_final(oracle):
    O = oracle
    rng = np.random.default_rng(8)
    nImg, rSig, nGroup = 11, 9, 3
    spec = rng.uniform(0.5, 2.0, size=(nImg, 4, rSig)).astype(np.float32)
    gid = rng.integers(1, nGroup + 1, nImg).astype(np.int32)
    gid[:nGroup] = np.arange(1, nGroup + 1)
    sigM, sigN, svd = O.sigma_accum(spec, gid, nGroup, True)
    for g in range(nGroup):
        sel = gid == g + 1
        assert np.allclose(sigM[g, :rSig], spec[sel, 2].sum(0) / 2, rtol=1e-6) and sigM[g, rSig] == sel.sum()
        assert np.allclose(svd[g, :rSig], np.sqrt(spec[sel, 0] / spec[sel, 1]).sum(0), rtol=1e-6)
    sig, rcp = O.sigma_final(sigM, sigN, svd, 100.0, 64, 1.32, True)
    alpha = np.sqrt(np.pi) * 100.0 / (64 * 1.32)
    m, n, s = (a[:, :rSig] / a[:, rSig:] for a in (sigM, sigN, svd))
    ratio = np.minimum(1, s)
    assert np.allclose(sig, ratio * m + (1 - ratio) * alpha * n, rtol=1e-5)
    assert np.allclose(rcp, -0.5 / sig, rtol=1e-6)
    # ungrouped: one pooled row copied to every group
    sigM, sigN, svd = O.sigma_accum(spec, gid, nGroup, False)
    sig, _ = O.sigma_final(sigM, sigN, svd, 100.0, 64, 1.32, False)
    assert np.array_equal(sig[0], sig[1]) and np.array_equal(sig[0], sig[2]) and sigM[0, rSig] == nImg


def test_golden_regression_next(oracle):
    """the oracle reproduces the committed vectors of tests/golden/oracle_next_n16.npz (see make_golden_next.py)"""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden_next as G
    want = np.load(os.path.join(here, "golden", "oracle_next_n16.npz"))
    got = G.compute()
    assert sorted(got.keys()) == sorted(want.files)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        assert a.shape == b.shape, k
        if a.dtype.kind in "iu":
            assert np.array_equal(a, b), k
        else:   # same machine arithmetic up to libm / FFT-library rounding
            assert np.allclose(a, b, rtol=2e-6, atol=1e-7 * max(1.0, float(np.abs(b).max()))), k
