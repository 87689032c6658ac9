"""Model::compareTwoHemispheres on the device against the oracle: gold-standard FSC, its mask-corrected
form with random-phase substitution (the device's Philox phases are read back and fed to the oracle, which applies the
reference's arithmetic to them), core mask, low-resolution averaging of the two halves."""
import ctypes as C

import numpy as np
import pytest
import scipy.fft as sfft

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _half_maps(N, seed):
    from thunder_amd import synth
    rng = np.random.default_rng(seed)
    ref = synth.blob_map(N, nblob=12, seed=seed)
    k = np.fft.fftfreq(N) * N
    r = np.sqrt(k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :N // 2 + 1] ** 2)
    ft = sfft.rfftn(ref)
    amp = np.abs(ft).mean() * 6.0 * (1.0 + r / 3.0) ** 0.5          # noise growing with resolution: the FSC falls off
    out = []
    for _ in range(2):
        n = (rng.normal(size=ft.shape) + 1j * rng.normal(size=ft.shape)) * amp
        rl = sfft.irfftn(ft + n, s=(N, N, N)).astype(np.float32)      # a real map: Hermitian-consistent FT
        out.append(np.ascontiguousarray(sfft.rfftn(rl).astype(np.complex64)))
    return out


@pytest.mark.parametrize("N", [32, 64])
def test_compare_hemispheres_core_mask_fsc(oracle, dev, N):
    from thunder_amd import capi
    from thunder_amd.capi import ptr, stream_ptr
    O = oracle
    A_h, B_h = _half_maps(N, 40 + N)
    rU, coreR, ew, avgR, seed, call = N // 2 - 1, 0.3 * N, 6.0, 5, 777, 3
    # core mask parity
    mask = torch.empty((N, N, N), dtype=torch.float32, device=dev)
    capi.call("thx_core_mask_dev", ptr(mask), N, coreR, ew, stream_ptr())
    mask_h = np.zeros((N, N, N), np.float32)
    O.lib().orc_core_mask(mask_h.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(N), C.c_float(coreR), C.c_float(ew))
    assert np.abs(mask.cpu().numpy() - mask_h).max() <= 2e-7
    assert mask_h.max() == 1.0 and mask_h.min() == 0.0
    # the device's run
    A, B = T(A_h, dev), T(B_h, dev)
    fsc = np.zeros(rU, np.float32)
    thres = C.c_int(-99)
    capi.call("thx_compare_hemispheres_dev", ptr(A), ptr(B), N, rU, fsc.ctypes.data, None, coreR, ew, 1, avgR, seed, call,
              C.byref(thres), stream_ptr())
    # the phases it drew (same Philox streams), for the oracle
    ph = []
    for c, src in ((call, A_h), (call + 1, B_h)):
        p = torch.empty((N, N, N // 2 + 1), dtype=torch.float32, device=dev)
        d = torch.empty((N, N, N // 2 + 1), dtype=torch.complex64, device=dev)
        capi.call("thx_random_phase_dev", ptr(d), ptr(T(src, dev)), N, thres.value, seed, c, ptr(p), stream_ptr())
        ph.append(p.cpu().numpy())
        # randomPhase itself against the oracle's arithmetic on the same angles
        want = np.empty_like(src)
        O.lib().orc_random_phase(want.ctypes.data_as(C.POINTER(C.c_float)), src.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(N),
                                 C.c_int(thres.value), ph[-1].ctypes.data_as(C.POINTER(C.c_float)))
        got = d.cpu().numpy()
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
        k = np.fft.fftfreq(N) * N
        u = np.rint(np.sqrt(k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :N // 2 + 1] ** 2))
        assert np.array_equal(got[u <= thres.value], src[u <= thres.value]) and (ph[-1][u <= thres.value] == 0).all()
        pp = ph[-1][u > thres.value]
        assert 0 <= pp.min() and pp.max() < 2 * np.pi + 1e-6 and abs(pp.mean() - np.pi) < 0.05   # uniform on [0, 2 pi)
    want = O.compare_hemispheres(A_h, B_h, N, rU, ph[0], ph[1], coreR=coreR, ew=ew, avg_r=avgR)
    assert want["thres"] == thres.value and 1 <= thres.value < rU - 2
    # FSC curves: rocFFT vs pocketfft round trips of the masked maps, fp64 shell sums on the device vs float atomics order
    np.testing.assert_allclose(fsc, want["fsc"], atol=5e-5, rtol=0)
    assert fsc[0] > 0.99 and fsc[thres.value] > 0.75 and fsc[-1] < 0.5
    # averaging: identical arithmetic, identical result
    assert np.array_equal(A.cpu().numpy(), want["A"]) and np.array_equal(B.cpu().numpy(), want["B"])
    k = np.fft.fftfreq(N) * N
    r2 = k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :N // 2 + 1] ** 2
    assert np.array_equal(A.cpu().numpy()[r2 < avgR ** 2], B.cpu().numpy()[r2 < avgR ** 2])
    assert np.array_equal(A.cpu().numpy()[r2 >= avgR ** 2], A_h[r2 >= avgR ** 2])


def test_compare_hemispheres_plain_and_given_mask(oracle, dev):
    from thunder_amd import capi
    from thunder_amd.capi import ptr, stream_ptr
    O = oracle
    N = 32
    A_h, B_h = _half_maps(N, 5)
    rU = N // 2 - 1
    A, B = T(A_h, dev), T(B_h, dev)
    fsc = np.zeros(rU, np.float32)
    # no mask: the plain gold-standard FSC; average everywhere (the K > 1 branch, src/Model.cpp:688-696)
    capi.call("thx_compare_hemispheres_dev", ptr(A), ptr(B), N, rU, fsc.ctypes.data, None, 0.0, 6.0, 1, -1, 1, 1, None, stream_ptr())
    np.testing.assert_allclose(fsc, O.fsc(A_h, B_h, N, rU), atol=2e-6)
    want = O.compare_hemispheres(A_h, B_h, N, rU, avg_r=-1)
    assert np.array_equal(A.cpu().numpy(), want["A"]) and torch.equal(A, B)
    # a provided mask (_maskFSC): an off-centre soft blob
    ax = np.fft.fftfreq(N) * N
    d = np.sqrt((ax[:, None, None] - 2) ** 2 + (ax[None, :, None] + 1) ** 2 + ax[None, None, :] ** 2)
    mask_h = np.clip((0.35 * N - d) / 4.0, 0, 1).astype(np.float32)
    A, B = T(A_h, dev), T(B_h, dev)
    thres = C.c_int(0)
    capi.call("thx_compare_hemispheres_dev", ptr(A), ptr(B), N, rU, fsc.ctypes.data, ptr(T(mask_h, dev)), 0.0, 6.0, 0, 0, 9, 20,
              C.byref(thres), stream_ptr())
    ph = []
    for c, src in ((20, A_h), (21, B_h)):
        p = torch.empty((N, N, N // 2 + 1), dtype=torch.float32, device=dev)
        dd = torch.empty((N, N, N // 2 + 1), dtype=torch.complex64, device=dev)
        capi.call("thx_random_phase_dev", ptr(dd), ptr(T(src, dev)), N, thres.value, 9, c, ptr(p), stream_ptr())
        ph.append(p.cpu().numpy())
    want = O.compare_hemispheres(A_h, B_h, N, rU, ph[0], ph[1], mask=mask_h)
    assert want["thres"] == thres.value
    np.testing.assert_allclose(fsc, want["fsc"], atol=5e-5, rtol=0)
    assert np.array_equal(A.cpu().numpy(), A_h)      # avgFlag off: the halves are untouched
