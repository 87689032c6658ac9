"""GPU parity tests of the callers either side of the E/M loop: re-mask (rocFFT 2-D
batches), re-centring ramps on images / volumes, the sigma update -- HIP path through the C ABI vs the CPU oracle.

Bars: phase ramps use device sincosf (<= 2 ulp from glibc): |delta| <= 5e-7 * |src|; pixels outside the radius are
bit-identical (untouched).  FFT pipelines and shell sums have no fixed summation order: tolerances at each assert."""
import numpy as np
import pytest

from _next_util import full_images
from _util import make_case

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _images(rng, n, N):
    import scipy.fft as sfft
    rl = rng.standard_normal((n, N, N)).astype(np.float32)
    return sfft.rfft2(rl).astype(np.complex64)


@pytest.mark.parametrize("N,nImg", [(32, 5), (64, 3), (256, 4), (32, 1030)])
def test_remask(oracle, dev, N, nImg):
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(N + nImg)
    ft = _images(rng, nImg, N)
    rpx = np.float32(N * 0.35)
    want = O.remask(ft, rpx, 1.0, 6.0)
    d = T(ft, dev)
    ops.remask(d, float(rpx), 6.0)
    got = d.cpu().numpy()
    # single-precision FFT pair on each side (FFTW/pocketfft vs rocFFT): error relative to the image's largest term
    scale = np.abs(want).reshape(nImg, -1).max(1)[:, None, None]
    assert (np.abs(got - want) / scale).max() <= 5e-6
    # empty batch is a no-op
    ops.remask(torch.empty((0, N, N // 2 + 1), dtype=torch.complex64, device=dev), float(rpx))


def test_remask_host_entry(oracle, dev):
    import ctypes as C
    from thunder_amd import capi
    O = oracle
    rng = np.random.default_rng(11)
    N, nImg = 32, 7
    ft = _images(rng, nImg, N)
    want = O.remask(ft, 13.2, 1.32, 6.0)
    bufs = [ft[l].copy() for l in range(nImg)]          # separately allocated images, as vector<Image>
    ptrs = (C.c_void_p * nImg)(*[b.ctypes.data for b in bufs])
    capi.call("thx_ReMask_host", C.cast(ptrs, C.c_void_p), 13.2, 1.32, 6.0, N, nImg)
    got = np.stack(bufs)
    assert np.abs(got - want).max() <= 5e-6 * np.abs(want).max()


@pytest.mark.parametrize("N", [16, 64])
def test_translate_image(oracle, dev, N):
    from thunder_amd import capi, ops
    O = oracle
    rng = np.random.default_rng(12)
    nImg = 4
    ft = _images(rng, nImg, N)
    tr = rng.normal(0, 2.0, size=(nImg, 2))
    got = ops.translate_image(T(ft, dev), T(tr, dev)).cpu().numpy()
    want = np.stack([O.translate_image(ft[l], tr[l, 0], tr[l, 1]) for l in range(nImg)])
    assert np.all(np.abs(got - want) <= 5e-7 * np.abs(ft) + 1e-30)
    r = N / 2 - 3
    got = ops.translate_image(T(ft, dev), T(tr, dev), r=r).cpu().numpy()
    want = np.stack([O.translate_image(ft[l], tr[l, 0], tr[l, 1], r=r) for l in range(nImg)])
    assert np.all(np.abs(got - want) <= 5e-7 * np.abs(ft) + 1e-30)
    j = np.fft.fftfreq(N, 1.0 / N)[:, None]
    i = np.arange(N // 2 + 1)[None, :]
    outside = ~((i * i + j * j) < np.float32(r) ** 2)
    assert np.array_equal(got[:, outside], ft[:, outside])
    # in place + host entry (TranslateI2D)
    d = T(ft, dev)
    ops.translate_image(d, T(tr, dev), r=r, out=d)
    assert np.array_equal(d.cpu().numpy(), got)
    h = ft[0].copy()
    capi.call("thx_TranslateI2D_host", 0, h.ctypes.data, float(tr[0, 0]), float(tr[0, 1]), int(r), N)
    assert np.array_equal(h, got[0])


def test_translate_volume(oracle, dev):
    import scipy.fft as sfft
    from thunder_amd import capi, ops
    O = oracle
    rng = np.random.default_rng(13)
    P = 32
    ft = sfft.rfftn(rng.standard_normal((P, P, P)).astype(np.float32)).astype(np.complex64)
    r, t = 11, (0.7, -1.9, 2.3)
    want = O.translate_volume(ft, r, *t)
    got = ops.translate_volume(T(ft, dev), r, *t).cpu().numpy()
    assert np.all(np.abs(got - want) <= 5e-7 * np.abs(ft) + 1e-30)
    k = np.fft.fftfreq(P, 1.0 / P)[:, None, None]
    j = np.fft.fftfreq(P, 1.0 / P)[None, :, None]
    i = np.arange(P // 2 + 1)[None, None, :]
    outside = ~((i * i + j * j + k * k) < r * r)
    assert np.array_equal(got[outside], ft[outside]) and not np.array_equal(got[~outside], ft[~outside])
    h = ft.copy()
    capi.call("thx_TranslateI_host", 0, h.ctypes.data, t[0], t[1], t[2], r, P)
    assert np.array_equal(h, got)


@pytest.mark.parametrize("N,projR,rSig", [(32, 13, 15), (64, 28, 31)])
def test_sigma_update(oracle, dev, N, projR, rSig):
    from thunder_amd import ops
    O = oracle
    rng = np.random.default_rng(14)
    P, nImg, nGroup = 2 * N, 9, 3
    _, vol, _ = make_case(O, N)
    im = full_images(O, vol, N, nImg, rng, projR)
    want = np.stack([O.sigma_image(vol, P, 2, N, projR, rSig, im["rot"][l], im["tran"][l], im["offset"][l],
                                   im["pixelSize"], im["attr"][l], im["img"][l], im["imgOri"][l])
                     for l in range(nImg)])
    attr = ops.ctf_attr_tensor(im["attr"], dev)
    spec = ops.sigma_spectra(T(vol, dev), P, 2, projR, rSig, T(im["img"], dev), T(im["imgOri"], dev), attr,
                             im["pixelSize"], T(im["rot"], dev), T(im["tran"], dev), T(im["offset"], dev))
    got = spec.cpu().numpy()
    # shell sums of <= ~100 positive terms in a different order + 2-ulp ramps/CTF: 2e-5 relative
    assert np.all(np.abs(got - want) <= 2e-5 * np.abs(want) + 1e-12)
    # without the re-centring offset rows 2 and 3 use the same ramp
    spec0 = ops.sigma_spectra(T(vol, dev), P, 2, projR, rSig, T(im["img"], dev), T(im["img"], dev), attr,
                              im["pixelSize"], T(im["rot"], dev), T(im["tran"], dev), None).cpu().numpy()
    assert np.array_equal(spec0[:, 2], spec0[:, 3]) and np.array_equal(spec0[:, :3], got[:, :3])

    gid = rng.integers(1, nGroup + 1, nImg).astype(np.int32)
    gid[:nGroup] = np.arange(1, nGroup + 1)
    for group in (True, False):
        accW = O.sigma_accum(got, gid, nGroup, group)
        acc = tuple(torch.zeros((nGroup, rSig + 1), dtype=torch.float32, device=dev) for _ in range(3))
        ops.sigma_accum(acc, spec, gid, group)
        for a, w in zip(acc, accW):
            assert np.allclose(a.cpu().numpy(), w, rtol=2e-6, atol=0)
        # accumulate a second batch on top (READ-MODIFY-WRITE, as the per-rank loop does)
        ops.sigma_accum(acc, spec, gid, group)
        for a, w in zip(acc, accW):
            assert np.allclose(a.cpu().numpy(), 2 * w, rtol=2e-6, atol=0)
        sigW, rcpW = O.sigma_final(*[2 * w for w in accW], 100.0, N, im["pixelSize"], group)
        sig, rcp = ops.sigma_final(acc, 100.0, N, im["pixelSize"], group)
        assert np.allclose(sig.cpu().numpy(), sigW, rtol=5e-6) and np.allclose(rcp.cpu().numpy(), rcpW, rtol=5e-6)
