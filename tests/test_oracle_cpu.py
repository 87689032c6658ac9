"""CPU tests of the oracle: reference-derived known answers, independent numpy cross-checks, invariants of the
algorithm, and the committed regression vectors (tests/golden/oracle_n16.npz -- produced by the oracle itself,
see tests/golden/make_golden.py: the oracle is PARITY-UNPINNED against the reference)."""
import os

import numpy as np
import pytest
import scipy.fft as sfft

from _util import edge_rotations, make_case, quat_to_mat

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_n16.npz")


def test_pixel_list_counts_match_reference_survey(oracle):
    """nPxl values measured with the compiled reference during the survey (SURVEY.md section 8 header and 8(d);
    BASELINE.md section 2): rU = N/2-2, rL = 0 -> 5941 / 24747 / 100941; rU = N/2-1 -> 6141 / 25135; r=24,rL=2 -> 866"""
    O = oracle
    assert [O.pixel_list(N, N // 2 - 2, 0)["nPxl"] for N in (128, 256, 512)] == [5941, 24747, 100941]
    assert [O.pixel_list(N, N // 2 - 1, 0)["nPxl"] for N in (128, 256)] == [6141, 25135]
    assert O.pixel_list(256, 24, 2)["nPxl"] == 866


def test_pixel_list_semantics(oracle):
    O = oracle
    N = 32
    pl = O.pixel_list(N, 12, 3)
    i, j = pl["iCol"], pl["iRow"]
    assert np.all(i >= 0) and not np.any((i == 0) & (j < 0))
    r = np.rint(np.hypot(i, j)).astype(int)
    assert np.all((r >= 3) & (r < 12)) and np.array_equal(r, pl["iSig"])
    assert np.array_equal(pl["iPxl"], np.where(j >= 0, j, j + N) * (N // 2 + 1) + i)
    assert np.array_equal(pl["iColPad"], 2 * i) and np.array_equal(pl["iRowPad"], 2 * j)
    # the product's host-side pixel list is the same list
    from thunder_amd.refine import pixel_list
    mine = pixel_list(N, 12, 3)
    for k in ("iCol", "iRow", "iPxl", "iSig", "iColPad", "iRowPad"):
        assert np.array_equal(mine[k], pl[k]), k
    assert O.pixel_list(N, 2, 5)["nPxl"] == 0  # empty list


def _np_interp(vol, P, x, y, z):
    """independent float32 numpy statement of the trilinear gather with Hermitian folding"""
    x, y, z = np.float32(x), np.float32(y), np.float32(z)
    conj = x < 0
    if conj:
        x, y, z = -x, -y, -z
    x0, y0, z0 = int(np.floor(x)), int(np.floor(y)), int(np.floor(z))
    xd, yd, zd = np.float32(x - np.float32(x0)), np.float32(y - np.float32(y0)), np.float32(z - np.float32(z0))
    acc = np.complex64(0)
    re, im = np.float32(0), np.float32(0)
    for k in range(2):
        for j in range(2):
            for i in range(2):
                w = np.float32(np.float32((xd if i else np.float32(1) - xd) * (yd if j else np.float32(1) - yd)) *
                               (zd if k else np.float32(1) - zd))
                v = vol[(z0 + k) % P, (y0 + j) % P, x0 + i]
                re = np.float32(re + np.float32(v.real * w))
                im = np.float32(im + np.float32(v.imag * w))
    return np.complex64(complex(re, -im if conj else im))


def test_interp_against_numpy(oracle):
    O = oracle
    rng = np.random.default_rng(0)
    P = 16
    vol = (rng.normal(size=(P, P, P // 2 + 1)) + 1j * rng.normal(size=(P, P, P // 2 + 1))).astype(np.complex64)
    pts = rng.uniform(-P / 2 + 1.5, P / 2 - 1.5, size=(200, 3))
    pts[:5] = [[0, 0, 0], [1, 2, 3], [-1, -2, -3], [0.5, -0.25, -0.75], [3.0, -1.0, -0.5]]
    for p in pts:
        got = O.interp_ft(vol, P, *p)
        want = _np_interp(vol, P, *p)
        assert got == want, (p, got, want)
    # grid points reproduce the stored voxel; x < 0 reproduces the conjugate of the mirrored voxel
    assert O.interp_ft(vol, P, 2, -3, 4) == vol[4, P - 3, 2]
    assert O.interp_ft(vol, P, -2, 3, -4) == np.conj(vol[4, P - 3, 2])


def test_project_is_linear_and_matches_rotation_of_plane(oracle):
    O = oracle
    N = 16
    ref, vol, pl = make_case(O, N)
    R = edge_rotations(np.random.default_rng(1), 3)[-1]
    a = O.project(vol, 2 * N, 2, R, pl["iCol"], pl["iRow"])
    b = O.project((2 * vol).astype(np.complex64), 2 * N, 2, R, pl["iCol"], pl["iRow"])
    assert np.array_equal(b, 2 * a)  # scaling by 2 is exact in binary floating point
    ident = np.eye(3).T.reshape(-1)
    s = O.project(vol, 2 * N, 2, ident, pl["iCol"], pl["iRow"])
    want = vol[0, np.where(pl["iRowPad"] >= 0, pl["iRowPad"], pl["iRowPad"] + 2 * N), pl["iColPad"]]
    assert np.array_equal(s, want)  # identity rotation reads grid points with weight exactly 1


def test_insert_is_adjoint_of_project(oracle):
    """<project(V), a> == <V, insert(a)> : trilinear gather and scatter use the same weights (to float rounding)"""
    O = oracle
    rng = np.random.default_rng(5)
    N, P = 16, 32
    pl = O.pixel_list(N, 6, 1)
    V = (rng.normal(size=(P, P, P // 2 + 1)) + 1j * rng.normal(size=(P, P, P // 2 + 1))).astype(np.complex64)
    a = (rng.normal(size=pl["nPxl"]) + 1j * rng.normal(size=pl["nPxl"])).astype(np.complex64)
    R = edge_rotations(rng, 3)[-2]
    s = O.project(V, P, 2, R, pl["iCol"], pl["iRow"])
    F = np.zeros_like(V)
    T = np.zeros(V.shape, np.float32)
    O.insertP(F, T, P, a, np.ones(pl["nPxl"], np.float32), R, 1.0, pl["iColPad"], pl["iRowPad"])
    # real inner product: the x<0 fold conjugates both sides consistently
    lhs = np.sum(s.real.astype(np.float64) * a.real + s.imag.astype(np.float64) * a.imag)
    rhs = np.sum(V.real.astype(np.float64) * F.real + V.imag.astype(np.float64) * F.imag)
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)
    assert abs(T.sum(dtype=np.float64) - pl["nPxl"]) <= 1e-3  # weights of each sample sum to 1


def test_ctf_and_translate_known_values(oracle):
    O = oracle
    N = 32
    pl = O.pixel_list(N, 10, 0)
    # zero shift -> ramp == 1; shift by N/2 along x -> (-1)^i
    r0 = O.translate(0.0, 0.0, N, pl["iCol"], pl["iRow"])
    assert np.array_equal(r0, np.ones(pl["nPxl"], np.complex64))
    r1 = O.translate(N / 2, 0.0, N, pl["iCol"], pl["iRow"])
    assert np.abs(r1.real - (-1.0) ** pl["iCol"]).max() < 1e-5 and np.abs(r1.imag).max() < 1e-5
    # CTF at the origin is the amplitude contrast; double-precision formula agrees to float rounding of chi
    attr = (3e5, 2.0e4, 1.9e4, 0.3, 2.7e7, 0.1, 0.0)
    c = O.ctf(1.32, *attr, N, pl["iCol"], pl["iRow"])
    k0 = np.where((pl["iCol"] == 0) & (pl["iRow"] == 0))[0][0]
    assert abs(c[k0] - 0.1) < 1e-7
    V, dU, dV, th, Cs, A, ph = attr
    lam = 12.2643247 / np.sqrt(V * (1 + V * 0.978466e-6))
    u = np.hypot(pl["iCol"] / (1.32 * N), pl["iRow"] / (1.32 * N))
    ang = np.arctan2(pl["iRow"], pl["iCol"]) - th
    df = -(dU + dV + (dU - dV) * np.cos(2 * ang)) / 2
    chi = np.pi * lam * df * u ** 2 + np.pi / 2 * Cs * lam ** 3 * u ** 4 - ph
    want = -np.sqrt(1 - A * A) * np.sin(chi) + A * np.cos(chi)
    assert np.abs(c - want).max() < 2e-4


def test_likelihood_forms_agree(oracle):
    O = oracle
    rng = np.random.default_rng(2)
    m, n = 500, 7
    dat = (rng.normal(size=(n, m)) + 1j * rng.normal(size=(n, m))).astype(np.complex64)
    pri = (rng.normal(size=m) + 1j * rng.normal(size=m)).astype(np.complex64)
    ctf = rng.uniform(-1, 1, size=(n, m)).astype(np.float32)
    sig = -rng.uniform(0.1, 1, size=(n, m)).astype(np.float32)
    res = np.zeros(n, np.float32)
    import ctypes as C
    from oracle.oracle import _p, c_f
    dat_pm, ctf_pm, sig_pm = (np.ascontiguousarray(a.T) for a in (dat, ctf, sig))
    O.lib().orc_logDataVSPrior_mn(_p(dat_pm, c_f), _p(pri, c_f), _p(ctf_pm, c_f), _p(sig_pm, c_f), C.c_int(n), C.c_int(m),
                                  _p(res, c_f))
    for j in range(n):
        a = O.logDataVSPrior(dat[j], pri, ctf[j], sig[j])
        assert a == res[j]  # same term order -> identical float sums (the reference's scalar 1xm and nxm forms)
        assert abs(a - O.logDataVSPrior_f64(dat[j], pri, ctf[j], sig[j])) < 1e-5 * abs(a)  # reference's own 1e-5 bar


def test_local_weights_equal_closed_form(oracle):
    """the running-baseline rescale (src/Optimiser.cpp:1383-1402) equals exp(L - max L) marginals"""
    O = oracle
    rng = np.random.default_rng(3)
    N = 16
    ref, vol, pl = make_case(O, N)
    from thunder_amd import synth
    q = synth.perturb_quats(synth.random_quats(1, rng), 7, 0.1, rng)[0]
    rot = np.stack([O.rotate3D(x) for x in q])
    dat = O.project(vol, 2 * N, 2, rot[3], pl["iCol"], pl["iRow"])
    ctf = np.ones(pl["nPxl"], np.float32)
    sig = np.full(pl["nPxl"], -0.5 / float(np.mean(np.abs(dat) ** 2)), np.float32)
    tran = rng.normal(0, 0.5, size=(4, 2))
    pR, pT = rng.uniform(0.5, 1.5, 7), rng.uniform(0.5, 1.5, 4)
    w = O.expect_local(vol, 2 * N, 2, N, pl["iCol"], pl["iRow"], dat, ctf, sig, rot, tran, pR=pR, pT=pT, pC=0.7)
    L = w["logW"][:, :, 0].astype(np.float64)
    s = np.exp(L - L.max())
    np.testing.assert_allclose(w["wR"], (s * (0.7 * pT)[None, :]).sum(1), rtol=1e-5)
    np.testing.assert_allclose(w["wT"], (s * (0.7 * pR)[:, None]).sum(0), rtol=1e-5)
    np.testing.assert_allclose(w["wC"][0], (s * pR[:, None] * pT[None, :]).sum(), rtol=1e-5)
    assert abs(w["baseLine"] - L.max()) < 1e-6 * abs(L.max()) and int(np.argmax(w["wR"] / 1)) >= 0


def test_roundtrip_fsc(oracle):
    """thunder_project -> thunder_reconstruct self-consistency: the survey measured FSC vs the input
    >= 0.998 where the map has signal with the compiled reference at N = 32 / 300 images; the oracle reproduces that"""
    O = oracle
    from thunder_amd import synth
    rng = np.random.default_rng(5)
    N, P = 32, 64
    ref, vol, pl = make_case(O, N)
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    T = np.zeros((P, P, P // 2 + 1), np.float32)
    one = np.ones(pl["nPxl"], np.float32)
    for q in synth.random_quats(300, rng):
        R = O.rotate3D(q)
        O.insertP(F, T, P, O.project(vol, P, 2, R, pl["iCol"], pl["iRow"]), one, R, 1.0, pl["iColPad"], pl["iRowPad"])
    O.normalise_TF(F, T, P)
    rec = O.reconstruct(F, T, P, N, 2, N // 2 - 2, MAP=False, gridCorr=True)
    f = O.fsc(sfft.rfftn(ref).astype(np.complex64), sfft.rfftn(rec).astype(np.complex64), N, N // 2)
    assert f[:8].min() >= 0.995, f
    assert 0.9 < float((ref * rec).sum() / (ref * ref).sum()) < 1.1


def test_symmetrize_makes_volume_symmetric(oracle):
    O = oracle
    from thunder_amd import synth
    N = 16
    ref, vol, pl = make_case(O, N)
    P = 2 * N
    sym = synth.cn_symmetry(4)
    Vs = O.symmetrize(vol, P, sym, 12.0)
    # a C4-symmetrised Fourier volume is invariant under the 90-degree rotation about z (to interpolation error: exact
    # here because 90-degree turns map grid points to grid points)
    R = sym[0]
    s1 = O.project(Vs, P, 2, np.eye(3).T.reshape(-1), pl["iCol"][:40], pl["iRow"][:40])
    s2 = O.project(Vs, P, 2, R, pl["iCol"][:40], pl["iRow"][:40])
    assert np.abs(s1 - s2).max() <= 1e-5 * np.abs(s1).max()
    assert np.array_equal(O.symmetrize(vol, P, np.zeros((0, 9)), 12.0), vol)


def test_golden_regression(oracle):
    O = oracle
    g = np.load(GOLD)
    N, pf = int(g["N"]), int(g["pf"])
    P = N * pf
    pl = O.pixel_list(N, N // 2 - 2, 0, pf)
    for k in ("iCol", "iRow", "iPxl", "iSig"):
        assert np.array_equal(pl[k], g[k])
    vol = O.set_projectee(g["ref"], pf)
    assert np.abs(vol - g["vol"]).max() <= 1e-5 * np.abs(g["vol"]).max()   # scipy FFT build may differ in the last bits
    vol = g["vol"]
    sl = np.stack([O.project(vol, P, pf, m, pl["iCol"], pl["iRow"]) for m in g["mats"]])
    assert np.array_equal(sl, g["slices"])
    ctf = np.stack([O.ctf(1.32, *a, N, pl["iCol"], pl["iRow"]) for a in g["attr"]])
    assert np.abs(ctf - g["ctf"]).max() <= 1e-6   # libm sinf/cosf
    ramps = np.stack([O.translate(np.float32(s[0]), np.float32(s[1]), N, pl["iCol"], pl["iRow"]) for s in g["shifts"]])
    assert np.abs(ramps - g["ramps"]).max() <= 1e-6
    ex = O.expect_local(vol, P, pf, N, pl["iCol"], pl["iRow"], g["dat"], g["ctf"][0], g["sig"], g["rot"], g["tran"],
                        pR=g["pR"], pT=g["pT"])
    np.testing.assert_allclose(ex["logW"], g["logW"], rtol=1e-6)
    np.testing.assert_allclose(ex["wR"], g["wR"], rtol=1e-4)
    np.testing.assert_allclose(ex["wT"], g["wT"], rtol=1e-4)
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    T = np.zeros((P, P, P // 2 + 1), np.float32)
    for k in range(len(g["mats"])):
        O.insertP(F, T, P, g["slices"][k], g["ctf"][k % 2], g["mats"][k], np.float32(0.25), pl["iColPad"], pl["iRowPad"])
    assert np.array_equal(F, g["F"]) and np.array_equal(T, g["T"])
    O.normalise_TF(F, T, P)
    Fs = O.symmetrize(F, P, g["sym"], (N // 2 - 2) * pf + 1)
    Ts = O.symmetrize(T, P, g["sym"], (N // 2 - 2) * pf + 1)
    assert np.array_equal(Fs, g["Fs"]) and np.array_equal(Ts, g["Ts"])
    rec = O.reconstruct(Fs, Ts, P, N, pf, N // 2 - 2, FSC=g["fscv"], joinHalf=True, MAP=True, gridCorr=True)
    assert np.abs(rec - g["rec"]).max() <= 1e-4 * np.abs(g["rec"]).max()
    tab = O.kernelRL_table()
    assert np.allclose(tab[:64], g["tab_head"], rtol=1e-6) and np.allclose(tab[-64:], g["tab_tail"], rtol=1e-6)
    assert abs(tab.sum(dtype=np.float64) - float(g["tab_sum"])) <= 1e-6 * float(g["tab_sum"])
    assert tab[0] == np.float32(g["nf"])  # MKB_RL_R2(0) == MKB_RL(0) == nf


def test_compare_hemispheres_oracle_properties(oracle):
    """mask-corrected FSC of Model::compareTwoHemispheres (oracle side): identical halves give FSC = 1 with or without the
    correction; for independent noise the phase-randomised, masked FSC is ~0 beyond the substitution shell, so the
    corrected curve stays close to the masked one; resP follows src/Functions/Spectrum.cpp:339-363"""
    import scipy.fft as sfft
    from thunder_amd import synth
    O = oracle
    N, rU = 32, 15
    rng = np.random.default_rng(3)
    ref = synth.blob_map(N, nblob=10)
    ft = np.ascontiguousarray(sfft.rfftn(ref).astype(np.complex64))
    ph = rng.uniform(0, 2 * np.pi, size=ft.shape).astype(np.float32)
    same = O.compare_hemispheres(ft, ft, N, rU, ph, ph, coreR=9.0)
    assert np.all(np.abs(same["fsc"] - 1) < 1e-4)
    assert O.res_p(np.array([1, .9, .85, .7, .9]), 0.8) == 2 and O.res_p(np.array([1, .9, .9]), 0.8) == 2
    assert O.res_p(np.array([1, .5, .9, .2]), 0.8, inverse=True) == 2
    noise = [np.ascontiguousarray(sfft.rfftn(rng.normal(size=(N, N, N)).astype(np.float32) * np.abs(ref).max() * 0.3).astype(np.complex64))
             for _ in range(2)]
    ph2 = rng.uniform(0, 2 * np.pi, size=ft.shape).astype(np.float32)
    out = O.compare_hemispheres(ft + noise[0], ft + noise[1], N, rU, ph, ph2, coreR=9.0, avg_r=4)
    plain = O.fsc(ft + noise[0], ft + noise[1], N, rU)
    assert out["thres"] is not None and out["fsc"][0] > 0.99
    assert np.all(out["fsc"][:out["thres"] + 2] >= plain[:out["thres"] + 2] - 0.02)   # masking removes noise outside the core
    k = np.fft.fftfreq(N) * N
    r2 = k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :N // 2 + 1] ** 2
    assert np.array_equal(out["A"][r2 < 16], out["B"][r2 < 16]) and not np.array_equal(out["A"][r2 >= 16], out["B"][r2 >= 16])
