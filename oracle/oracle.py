"""ctypes front-end of the CPU oracle (oracle/thunder_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by anything under thunder_amd/.

PARITY UNPINNED (see the header of thunder_oracle.c and DESIGN.md section 3): the reference is not
buildable in this image, and its tests hold no golden vectors for this path.

The 3-D FFT stages of Projector::setProjectee (src/Projector.cpp:123-148) and
Reconstructor::reconstruct (src/Reconstructor.cpp:1129-1831) use FFTW single precision in the
reference (src/FFT.cpp:176-232: unnormalised forward, 1/size on backward); here they go through
scipy.fft on float32 data (same convention, pocketfft single precision).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.fft as sfft

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f = C.POINTER(C.c_float)
c_d = C.POINTER(C.c_double)
c_i = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "libthunder_oracle.so")
    src = os.path.join(_HERE, "thunder_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libthunder_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_TIK_RL.restype = C.c_float
        _LIB.orc_TIK_RL.argtypes = [C.c_float]
        _LIB.orc_MKB_RL.restype = C.c_float
        _LIB.orc_MKB_RL.argtypes = [C.c_float] * 3
        _LIB.orc_logDataVSPrior.restype = C.c_float
        _LIB.orc_logDataVSPrior_f64.restype = C.c_double
        _LIB.orc_update_W_checkC.restype = C.c_float
        _LIB.orc_pixel_list.restype = C.c_int
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---------------------------------------------------------------------------------------------
def pixel_list(N, rU, rL, pf=2):
    """Optimiser::allocPreCalIdx (src/Optimiser.cpp:7991-8041) -> dict of int32 arrays."""
    cap = (N // 2 + 1) * N
    arrs = [np.zeros(cap, np.int32) for _ in range(6)]
    n = lib().orc_pixel_list(C.c_int(N), C.c_float(rU), C.c_float(rL), C.c_int(pf), *[_p(a, c_i) for a in arrs])
    names = ["iCol", "iRow", "iPxl", "iSig", "iColPad", "iRowPad"]
    out = {k: a[:n].copy() for k, a in zip(names, arrs)}
    out["nPxl"] = n
    return out


def ctf(pixelSize, voltage, defU, defV, theta, Cs, ampC, phaseShift, N, iCol, iRow):
    """CTF(RFLOAT* dst, ...) src/CTF.cpp:113-151."""
    iCol, iRow = i32(iCol), i32(iRow)
    dst = np.zeros(len(iCol), np.float32)
    lib().orc_ctf(_p(dst, c_f), C.c_float(pixelSize), C.c_float(voltage), C.c_float(defU), C.c_float(defV),
                  C.c_float(theta), C.c_float(Cs), C.c_float(ampC), C.c_float(phaseShift), C.c_int(N), C.c_int(N),
                  _p(iCol, c_i), _p(iRow, c_i), C.c_int(len(iCol)))
    return dst


def translate(tx, ty, N, iCol, iRow, src=None):
    """translate(Complex* dst, tx, ty, ...) src/Image/ImageFunctions.cpp:233-252 (:471-492 with src)."""
    iCol, iRow = i32(iCol), i32(iRow)
    dst = np.zeros(len(iCol), np.complex64)
    if src is None:
        lib().orc_translate(_p(dst, c_f), C.c_float(tx), C.c_float(ty), C.c_int(N), C.c_int(N), _p(iCol, c_i),
                            _p(iRow, c_i), C.c_int(len(iCol)))
    else:
        src = c64(src)
        lib().orc_translate_src(_p(dst, c_f), _p(src, c_f), C.c_float(tx), C.c_float(ty), C.c_int(N), C.c_int(N),
                                _p(iCol, c_i), _p(iRow, c_i), C.c_int(len(iCol)))
    return dst


def rotate3D(q):
    """rotate3D(dmat33&, dvec4) src/Geometry/Euler.cpp:181-189 -> 9 doubles column-major."""
    q = f64(q)
    out = np.zeros(9, np.float64)
    lib().orc_rotate3D(_p(out, c_d), _p(q, c_d))
    return out


def symmetry(name, cap=128):
    """Symmetry::init(sym) (src/Geometry/Symmetry.cpp:61-278, src/Geometry/SymmetryFunctions.cpp:13-164): the non-identity
    elements of the point group in the reference's order -> dict(n, R [n][9] column-major, quat [n][4])"""
    L = lib()
    L.orc_symmetry.restype = C.c_int
    R, q = np.zeros((cap, 9)), np.zeros((cap, 4))
    n = L.orc_symmetry(name.encode(), _p(R, c_d), _p(q, c_d), C.c_int(cap))
    if n < 0:
        raise ValueError("unknown point group %r" % name if n == -1 else "more than %d symmetry elements" % cap)
    return dict(n=n, R=np.ascontiguousarray(R[:n]), quat=np.ascontiguousarray(q[:n]), name=name)


def symmetrise(q, symQuat, anchor=None):
    """Particle::symmetrise(anchor) (src/Particle.cpp:2445-2470) on q [n][4] -> new array; anchor None = ANCHOR_POINT_2"""
    q = f64(q).copy().reshape(-1, 4)
    symQuat = f64(symQuat).reshape(-1, 4)
    a = None if anchor is None else f64(anchor)
    lib().orc_symmetrise(_p(q, c_d), C.c_int(len(q)), _p(symQuat, c_d), C.c_int(len(symQuat)), _p(a, c_d))
    return q


def project(vol, P, pf, mat, iCol, iRow):
    """Projector::project(Complex*, dmat33, iCol, iRow, nPxl) src/Projector.cpp:356-374."""
    vol, mat, iCol, iRow = c64(vol), f64(mat), i32(iCol), i32(iRow)
    dst = np.zeros(len(iCol), np.complex64)
    lib().orc_project(_p(dst, c_f), _p(vol, c_f), C.c_int(P), C.c_int(pf), _p(mat, c_d), _p(iCol, c_i),
                      _p(iRow, c_i), C.c_int(len(iCol)))
    return dst


def interp_ft(vol, P, x, y, z):
    vol = c64(vol)
    out = np.zeros(1, np.complex64)
    lib().orc_interp_ft(_p(vol, c_f), C.c_int(P), C.c_float(x), C.c_float(y), C.c_float(z), _p(out, c_f))
    return out[0]


def logDataVSPrior(dat, pri, ctf_, sigRcp):
    dat, pri, ctf_, sigRcp = c64(dat), c64(pri), f32(ctf_), f32(sigRcp)
    return float(lib().orc_logDataVSPrior(_p(dat, c_f), _p(pri, c_f), _p(ctf_, c_f), _p(sigRcp, c_f),
                                          C.c_int(len(ctf_))))


def logDataVSPrior_f64(dat, pri, ctf_, sigRcp):
    dat, pri, ctf_, sigRcp = c64(dat), c64(pri), f32(ctf_), f32(sigRcp)
    return float(lib().orc_logDataVSPrior_f64(_p(dat, c_f), _p(pri, c_f), _p(ctf_, c_f), _p(sigRcp, c_f),
                                              C.c_int(len(ctf_))))


def expect_local(vol, P, pf, N, iCol, iRow, dat, ctfP, sigRcp, rot, tran, nD=1, pC=1.0, pR=None, pT=None, pD=None,
                 cSearch=False):
    """One particle-filter phase of one image, src/Optimiser.cpp:1225-1406.
    rot [nR][9] column-major, tran [nT][2]; returns dict(wC, wR, wT, wD, baseLine, logW[nR][nT][nD])."""
    vol, iCol, iRow = c64(vol), i32(iCol), i32(iRow)
    dat, ctfP, sigRcp = c64(dat), f32(ctfP), f32(sigRcp)
    rot, tran = f64(rot).reshape(-1, 9), f64(tran).reshape(-1, 2)
    nR, nT, nPxl = len(rot), len(tran), len(iCol)
    pR = f64(np.ones(nR) if pR is None else pR)
    pT = f64(np.ones(nT) if pT is None else pT)
    pD = f64(np.ones(nD) if pD is None else pD)
    wC = np.zeros(1, np.float32)
    wR = np.zeros(nR, np.float32)
    wT = np.zeros(nT, np.float32)
    wD = np.zeros(nD, np.float32)
    base = np.zeros(1, np.float32)
    logW = np.zeros((nR, nT, nD), np.float32)
    lib().orc_expect_local(_p(vol, c_f), C.c_int(P), C.c_int(pf), C.c_int(N), _p(iCol, c_i), _p(iRow, c_i),
                           C.c_int(nPxl), _p(dat, c_f), _p(ctfP, c_f), C.c_int(1 if cSearch else 0), _p(sigRcp, c_f),
                           _p(rot, c_d), C.c_int(nR), _p(tran, c_d), C.c_int(nT), C.c_int(nD), C.c_double(pC),
                           _p(pR, c_d), _p(pT, c_d), _p(pD, c_d), _p(wC, c_f), _p(wR, c_f), _p(wT, c_f), _p(wD, c_f),
                           _p(base, c_f), _p(logW, c_f))
    return dict(wC=wC, wR=wR, wT=wT, wD=wD, baseLine=float(base[0]), logW=logW)


def expect_global(rotP, traP, datP, ctfP, sigRcpP, nK, kIdx, pR, pT, wC, wR, wT, baseLine):
    """Scanning phase for class kIdx, src/Optimiser.cpp:756-894 (in-place on wC/wR/wT/baseLine).
    rotP [nR][nPxl], traP [nT][nPxl], datP/ctfP/sigRcpP pixel-major [nPxl][nImg];
    wC [nImg][nK], wR [nK][nImg][nR], wT [nK][nImg][nT], baseLine [nImg] (NaN = unset)."""
    rotP, traP, datP = c64(rotP), c64(traP), c64(datP)
    ctfP, sigRcpP, pR, pT = f32(ctfP), f32(sigRcpP), f64(pR), f64(pT)
    nR, nPxl = rotP.shape
    nT = traP.shape[0]
    nImg = datP.shape[1]
    for a in (wC, wR, wT, baseLine):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().orc_expect_global(_p(rotP, c_f), _p(traP, c_f), _p(datP, c_f), _p(ctfP, c_f), _p(sigRcpP, c_f),
                            C.c_int(nImg), C.c_int(nPxl), C.c_int(nK), C.c_int(kIdx), C.c_int(nR), C.c_int(nT),
                            _p(pR, c_d), _p(pT, c_d), _p(wC, c_f), _p(wR, c_f), _p(wT, c_f), _p(baseLine, c_f))


def insertP(F, T, P, src, ctf_, rot, w, iColPad, iRowPad):
    """Reconstructor::insertP src/Reconstructor.cpp:782-863 (in place; F complex64, T float32)."""
    assert F.dtype == np.complex64 and T.dtype == np.float32 and F.flags.c_contiguous and T.flags.c_contiguous
    src, ctf_, rot, iColPad, iRowPad = c64(src), f32(ctf_), f64(rot), i32(iColPad), i32(iRowPad)
    lib().orc_insertP(_p(F, c_f), _p(T, c_f), C.c_int(P), _p(src, c_f), _p(ctf_, c_f), _p(rot, c_d), C.c_float(w),
                      _p(iColPad, c_i), _p(iRowPad, c_i), C.c_int(len(iColPad)))


def normalise_TF(F, T, P):
    lib().orc_normalise_TF(_p(F, c_f), _p(T, c_f), C.c_int(P))


def symmetrize(vol, P, symMat, r):
    """SYMMETRIZE_FT include/Geometry/Transformation.h:170-194; symMat [nSym][9] column-major."""
    symMat = f64(symMat).reshape(-1, 9)
    is_c = 1 if vol.dtype == np.complex64 else 0
    out = np.empty_like(vol)
    lib().orc_symmetrize(_p(out, c_f), _p(np.ascontiguousarray(vol), c_f), C.c_int(P), C.c_int(is_c), _p(symMat, c_d),
                         C.c_int(len(symMat)), C.c_double(r))
    return out


def kernelRL_table(a=1.9, alpha=15.0, n=100000):
    tab = np.zeros(n + 1, np.float32)
    lib().orc_kernelRL_table(_p(tab, c_f), C.c_int(n), C.c_float(a), C.c_float(alpha))
    return tab


def mkb_rl(r, a=1.9, alpha=15.0):
    """MKB_RL, src/Functions/Functions.cpp (nf = MKB_RL(0, a, alpha), src/Reconstructor.cpp:2600)"""
    return float(lib().orc_MKB_RL(C.c_float(r), C.c_float(a), C.c_float(alpha)))


def tik_rl(r):
    """TIK_RL, src/Functions/Functions.cpp:236-239, elementwise on a float32 array"""
    L = lib()
    r = f32(r).reshape(-1)
    return np.array([L.orc_TIK_RL(C.c_float(float(x))) for x in r], np.float32)


def fsc(A, B, P, nShell):
    """FSC(vec&, Volume, Volume) src/Functions/Spectrum.cpp:302-337 on two half-complex FTs."""
    A, B = c64(A), c64(B)
    out = np.zeros(nShell, np.float32)
    lib().orc_fsc(_p(out, c_f), C.c_int(nShell), _p(A, c_f), _p(B, c_f), C.c_int(P))
    return out


def stop_rule_init(transS, ctfRefineS=0.01):
    """state of the per-image stop rule of the local search, src/Optimiser.cpp:1168-1183 (not OPTIMISER_COMPRESS_CRITERIA)"""
    return dict(k1=1.0, k2=1.0, k3=1.0, s0=5.0 * transS, s1=5.0 * transS, d=5.0 * ctfRefineS, noDec=0)


def stop_rule(st, k1, k2, k3, s0, s1, d=0.0):
    """src/Optimiser.cpp:1510-1615, MODE_3D branch, evaluated after a phase with index >= MIN_N_PHASE_PER_ITER_LOCAL:
    returns True when the image's search ends (nPhaseWithNoVariDecrease == N_PHASE_WITH_NO_VARI_DECREASE = 1)"""
    f = 0.95   # PARTICLE_FILTER_DECREASE_FACTOR
    f2 = f * f   # gsl_pow_2(PARTICLE_FILTER_DECREASE_FACTOR): the product is formed first, as there
    if (k1 < st["k1"] * f2) or (k2 < st["k2"] * f2) or (k3 < st["k3"] * f2) or (s0 < st["s0"] * f) or \
       (s1 < st["s1"] * f) or (d < st["d"] * f):
        st["noDec"] = 0
    else:
        st["noDec"] += 1
    for key, v in (("k1", k1), ("k2", k2), ("k3", k3), ("s0", s0), ("s1", s1), ("d", d)):
        if v < st[key]:
            st[key] = v
    return st["noDec"] == 1


def res_p(fsc_, thres, pf=1, rL=1, inverse=False):
    """resP(fsc, thres, pf, rL, inverse), src/Functions/Spectrum.cpp:339-363"""
    n = len(fsc_)
    if inverse:
        result = n - 1
        while result >= rL and not (fsc_[result] > thres):
            result -= 1
    else:
        result = rL
        while result < n and not (fsc_[result] < thres):
            result += 1
        result -= 1
    return int(result / pf) if result >= 0 else -int(-result / pf)


def compare_hemispheres(A, B, N, rU, phasesA=None, phasesB=None, mask=None, coreR=0.0, ew=6.0, avg_r=None):
    """Model::compareTwoHemispheres, MODE_3D (src/Model.cpp:307-700): A, B complex64 half FTs [N][N][N/2+1] of the two
    half maps.  With a mask (given, _maskFSC, or the core mask of radius coreR, _coreFSC) the mask-corrected FSC of
    :424-563: fscUnmask -> randomPhaseThres = resP(fscUnmask, 0.8, 1, 1) -> FSC of the masked phase-randomised halves ->
    FSC of the masked halves -> (fscMask - fscRF) / (1 - fscRF) beyond randomPhaseThres + 2.  phasesA/B: the random
    phases per stored element (the reference draws them from GSL's global generator).  avg_r: None = no averaging,
    >= 0: average inside that radius (:663-674), < 0: everywhere.  Returns dict(fsc, thres, A, B)."""
    L = lib()
    A, B = c64(A).copy(), c64(B).copy()
    fsc_ = fsc(A, B, N, rU)
    thres = None
    if mask is not None or coreR > 0:
        if mask is None:
            mask = np.zeros((N, N, N), np.float32)
            L.orc_core_mask(_p(mask, c_f), C.c_int(N), C.c_float(coreR), C.c_float(ew))
        mask = f32(mask)
        thres = res_p(fsc_, 0.8, 1, 1, False)

        def masked_ft(ft):
            rl = np.ascontiguousarray(sfft.irfftn(ft, s=(N, N, N)).astype(np.float32))   # FFT::bw incl. 1/size
            out = np.empty_like(rl)
            L.orc_alpha_mask(_p(out, c_f), _p(rl, c_f), _p(mask, c_f), C.c_float(0.0), C.c_size_t(rl.size))
            return np.ascontiguousarray(sfft.rfftn(out).astype(np.complex64))
        rp = []
        for src, ph in ((A, phasesA), (B, phasesB)):
            d = np.empty_like(src)
            L.orc_random_phase(_p(d, c_f), _p(src, c_f), C.c_int(N), C.c_int(thres), _p(f32(ph), c_f))
            rp.append(masked_ft(d))
        fscRF = fsc(rp[0], rp[1], N, rU)
        fscMask = fsc(masked_ft(A), masked_ft(B), N, rU)
        out = np.empty(rU, np.float32)
        for i in range(rU):
            out[i] = fscMask[i] if i < thres + 2 else (fscMask[i] - fscRF[i]) / (np.float32(1) - fscRF[i])
        fsc_ = out
    if avg_r is not None:
        L.orc_average_halves(_p(A, c_f), _p(B, c_f), C.c_int(N), C.c_int(int(avg_r)))
    return dict(fsc=fsc_, thres=thres, A=A, B=B)


# ---------------------------------------------------------------------------------------------
def set_projectee(ref_rl, pf=2):
    """Projector::setProjectee(Volume) src/Projector.cpp:123-148 starting from the real-space map
    (the reference first does fft.bw on the FT it is handed): zero-pad x pf, divide by TIK_RL, r2c.
    ref_rl: float32 [N][N][N] in wrapped-index layout (index (k<0?k+N:k), ...: origin at [0,0,0]).
    Returns the complex64 padded FT [P][P][P/2+1]."""
    ref_rl = f32(ref_rl)
    N = ref_rl.shape[0]
    P = N * pf
    pad = np.zeros((P, P, P), np.float32)
    lib().orc_pad_gridcorr(_p(pad, c_f), _p(ref_rl, c_f), C.c_int(N), C.c_int(pf))
    return sfft.rfftn(pad).astype(np.complex64)


def reconstruct(F, T, P, N, pf, maxRadius, FSC=None, joinHalf=False, MAP=True, gridCorr=True, a=1.9, alpha=15.0,
                return_iters=False, max_rounds=30, T_inplace=False, force_rounds=None):
    """Reconstructor::reconstruct(Volume&) src/Reconstructor.cpp:1129-1831, MODE_3D.
    F complex64 [P][P][P/2+1], T float32 same grid (both AFTER prepareTF), P = PAD_SIZE = _pf * _size; returns float32 [N][N][N]
    (wrapped-index layout).  _size == _N (P == N * pf) is the grid at Nyquist; after Reconstructor::resizeSpace (:184-198; every
    iteration below Nyquist, src/Model.cpp:1113: _size = min(N, (rU + ceil(a)) * 2)) the four volumes and the gridding loop live on
    the SMALL grid P < N * pf -- the FFT plans are PAD_SIZE^3 (:121-124), convoluteC still divides QUAD_3 by (_N * _pf)^2
    (:2639-2645) -- and only the last step places F * W into an (_N * _pf)^3 padDst (:1677-1701) for the final c2r and
    VOL_EXTRACT_RL.  The reference changes _T3D in place (Wiener term :1242-1270, 1e-25 floor :1322-1324), so a
    second reconstruct() of the same iteration starts from the first one's T: T_inplace=True does the same to the caller's
    array (float32, contiguous); the default works on a copy.  force_rounds = k runs exactly k balancing rounds whatever the
    stop rule of :1530-1551 says (tests use it to compare two implementations after the SAME round when the rule -- which
    compares a max norm with 0.95 x its previous value on a loop that has not converged -- lets them stop in different ones)."""
    L = lib()
    F = c64(F).copy()
    if T_inplace:
        assert T.dtype == np.float32 and T.flags.c_contiguous
    else:
        T = f32(T).copy()
    n = T.size
    if MAP:
        FSC = f32(FSC)
        L.orc_wiener_T(_p(T, c_f), C.c_int(P), C.c_int(pf), C.c_int(maxRadius), _p(FSC, c_f), C.c_int(len(FSC)),
                       C.c_int(1 if joinHalf else 0))
    W = np.zeros(n, np.float32)
    L.orc_init_W_floor_T(_p(W, c_f), _p(T, c_f), C.c_int(P), C.c_int(pf), C.c_int(maxRadius))
    iters = 0
    diffs = []
    if gridCorr:
        tab = kernelRL_table(a, alpha)
        nf = float(L.orc_MKB_RL(C.c_float(0), C.c_float(a), C.c_float(alpha)))
        Cv = np.zeros((P, P, P // 2 + 1), np.complex64)
        diffC = diffCPrev = np.float32(np.finfo(np.float32).max)
        nNoDec = 0
        for m in range(max_rounds if force_rounds is None else force_rounds):  # MAX_N_ITER_BALANCE = 30
            L.orc_calc_C(_p(Cv, c_f), _p(T, c_f), _p(W, c_f), C.c_int(P))
            crl = sfft.irfftn(Cv, s=(P, P, P)).astype(np.float32)  # bwExecutePlan incl. 1/size
            crl = np.ascontiguousarray(crl)
            L.orc_convolute_rl(_p(crl, c_f), C.c_int(P), C.c_int(N * pf), _p(tab, c_f), C.c_int(100000), C.c_float(nf))
            Cv = np.ascontiguousarray(sfft.rfftn(crl).astype(np.complex64))
            diffCPrev = diffC
            diffC = np.float32(L.orc_update_W_checkC(_p(W, c_f), _p(Cv, c_f), C.c_int(P), C.c_int(pf),
                                                     C.c_int(maxRadius)))
            diffs.append(float(diffC))
            iters = m + 1
            if diffC > diffCPrev * np.float32(0.95):
                nNoDec += 1
            else:
                nNoDec = 0
            if force_rounds is None and ((diffC < 1e-2) or ((m >= 10) and (nNoDec == 2))):
                break
    else:
        L.orc_W_nogridcorr(_p(W, c_f), _p(T, c_f), C.c_int(P), C.c_int(pf), C.c_int(maxRadius))
    PN = N * pf
    assert P <= PN and P % 2 == 0
    pad = np.zeros((PN, PN, PN // 2 + 1), np.complex64)
    if P == PN:
        L.orc_FW(_p(pad, c_f), _p(F, c_f), _p(W, c_f), C.c_int(P), C.c_int(pf), C.c_int(maxRadius))
    else:
        L.orc_FW_pad(_p(pad, c_f), C.c_int(PN), _p(F, c_f), _p(W, c_f), C.c_int(P), C.c_int(pf), C.c_int(maxRadius))
    prl = np.ascontiguousarray(sfft.irfftn(pad, s=(PN, PN, PN)).astype(np.float32))
    dst = np.zeros((N, N, N), np.float32)
    L.orc_extract_tik(_p(dst, c_f), _p(prl, c_f), C.c_int(PN), C.c_int(N), C.c_int(pf), C.c_int(1))
    if return_iters:
        return dst, iters, diffs, W
    return dst


def baseline_block(vol, P, pf, N, pl, dat, ctf_, sigRcp, rot, tran, recoRot, recoTran, F, T, groups=1):
    """bench.py cpu_baseline: fixed-work E (nPhase x nR x nT) + M (mReco inserts) for a block.  groups > 1: F / T are
    [groups][P][P][P/2+1] -- one private pair per group of threads (the reference's MPI ranks); the caller sums them."""
    if groups > 1:
        assert F.shape[0] == groups and T.shape[0] == groups and F.flags.c_contiguous and T.flags.c_contiguous
    vol, dat, ctf_, sigRcp = c64(vol), c64(dat), f32(ctf_), f32(sigRcp)
    rot, tran, recoRot, recoTran = f64(rot), f64(tran), f64(recoRot), f64(recoTran)
    nImg, nPhase, nR = rot.shape[0], rot.shape[1], rot.shape[2]
    nT = tran.shape[2]
    mReco = recoRot.shape[1]
    wR = np.zeros((nImg, nR), np.float32)
    lib().orc_baseline_block_groups(_p(vol, c_f), C.c_int(P), C.c_int(pf), C.c_int(N), _p(i32(pl["iCol"]), c_i),
                             _p(i32(pl["iRow"]), c_i), _p(i32(pl["iColPad"]), c_i), _p(i32(pl["iRowPad"]), c_i),
                             C.c_int(pl["nPxl"]), C.c_int(nImg), _p(dat, c_f), _p(ctf_, c_f), _p(sigRcp, c_f),
                             _p(rot, c_d), _p(tran, c_d), C.c_int(nPhase), C.c_int(nR), C.c_int(nT), _p(recoRot, c_d),
                             _p(recoTran, c_d), C.c_int(mReco), _p(F, c_f), _p(T, c_f), _p(wR, c_f), C.c_int(groups))
    return wR


# ---------------------------------------------------------------------------------------------
# SURVEY 8 "next" rows f1 / f2
def soft_mask(N, r, ew):
    """softMask(Image& mask, r, ew) src/Functions/Mask.cpp:334-350 -> float32 [N][N] (wrapped index)."""
    m = np.zeros((N, N), np.float32)
    lib().orc_soft_mask(_p(m, c_f), C.c_int(N), C.c_float(r), C.c_float(ew))
    return m


def remask(imgs, maskRadius, pixelSize, ew=6.0):
    """Optimiser::reMaskImg src/Optimiser.cpp:6093-6149 (zeroMask): per image c2r (1/size), x mask, r2c.
    imgs complex64 [nImg][N][N/2+1]; returns a new array."""
    imgs = c64(imgs)
    N = imgs.shape[1]
    mask = soft_mask(N, np.float32(maskRadius) / np.float32(pixelSize), ew)
    out = np.empty_like(imgs)
    for l in range(imgs.shape[0]):
        rl = np.ascontiguousarray(sfft.irfft2(imgs[l], s=(N, N), norm="forward").astype(np.float32))
        lib().orc_scale_mul_rl(_p(rl, c_f), _p(mask, c_f), C.c_size_t(rl.size))
        out[l] = sfft.rfft2(rl).astype(np.complex64)
    return out


def translate_image(src, tx, ty, r=-1.0, dst=None):
    """translate(Image&, const Image&, [r,] tx, ty) src/Image/ImageFunctions.cpp:269-284 / :322-339."""
    src = c64(src)
    N = src.shape[0]
    out = src.copy() if dst is None else dst
    lib().orc_translate_image(_p(out, c_f), _p(src, c_f), C.c_int(N), C.c_float(r), C.c_float(tx), C.c_float(ty))
    return out


def translate_volume(src, r, tx, ty, tz):
    """translate(Volume&, const Volume&, r, tx, ty, tz) src/Image/ImageFunctions.cpp:363-384 (dst = copy of src)."""
    src = c64(src)
    P = src.shape[0]
    out = src.copy()
    lib().orc_translate_volume(_p(out, c_f), _p(src, c_f), C.c_int(P), C.c_float(r), C.c_float(tx), C.c_float(ty),
                               C.c_float(tz))
    return out


def disc_list(N, r):
    """Pixel set of Projector::project(Image&, mat) / powerSpectrum (keeps (0, j<0))."""
    lib().orc_disc_list.restype = C.c_int
    cap = (2 * r + 1) * (r + 1)
    arrs = [np.zeros(cap, np.int32) for _ in range(4)]
    n = lib().orc_disc_list(C.c_int(N), C.c_int(r), *[_p(a, c_i) for a in arrs])
    out = {k: a[:n].copy() for k, a in zip(["iCol", "iRow", "iPxl", "iSig"], arrs)}
    out["nPxl"] = n
    return out


def power_spectrum(img, r):
    """powerSpectrum(vec&, const Image&, r, 1) src/Functions/Spectrum.cpp:161-190."""
    img = c64(img)
    out = np.zeros(r, np.float32)
    lib().orc_power_spectrum(_p(out, c_f), _p(img, c_f), C.c_int(img.shape[0]), C.c_int(r))
    return out


def sigma_image(vol, P, pf, N, projR, rSig, rot, tran, offset, pixelSize, attr, img, imgOri):
    """Per-image part of Optimiser::allReduceSigma (src/Optimiser.cpp:6443-6565).
    attr = (voltage, defocusU, defocusV, theta, Cs, ampContrast, phaseShift); returns float32 [4][rSig]
    = sSVD, dSVD, vSigM, vSigN."""
    vol, img, imgOri = c64(vol), c64(img), c64(imgOri)
    rot, tran = f64(rot), f64(tran)
    offset = None if offset is None else f64(offset)
    out = np.zeros((4, rSig), np.float32)
    lib().orc_sigma_image(_p(vol, c_f), C.c_int(P), C.c_int(pf), C.c_int(N), C.c_int(projR), C.c_int(rSig),
                          _p(rot, c_d), _p(tran, c_d), _p(offset, c_d), C.c_float(pixelSize),
                          *[C.c_float(a) for a in attr], _p(img, c_f), _p(imgOri, c_f), _p(out[0], c_f),
                          _p(out[1], c_f), _p(out[2], c_f), _p(out[3], c_f))
    return out


def norm_residual(vol, P, pf, N, projR, rL, rNorm, rot, tran, pixelSize, attr, img):
    """Per-image part of Optimiser::normCorrection (src/Optimiser.cpp:6201-6358): norm(_ID[l]) of the masked image `img` against
    the top pose's CTF-modulated slice, summed over rL^2 <= i^2 + j^2 < rNorm^2."""
    vol, img = c64(vol), c64(img)
    rot, tran = f64(rot), f64(tran)
    lib().orc_norm_residual.restype = C.c_float
    return float(lib().orc_norm_residual(_p(vol, c_f), C.c_int(P), C.c_int(pf), C.c_int(N), C.c_int(projR), C.c_float(rL), C.c_float(rNorm),
                                         _p(rot, c_d), _p(tran, c_d), C.c_float(pixelSize), *[C.c_float(a) for a in attr], _p(img, c_f)))


def median(values):
    """median(vec, n), src/Functions/Functions.cpp:246-252"""
    v = f32(values)
    lib().orc_median.restype = C.c_float
    return float(lib().orc_median(_p(v, c_f), C.c_int(len(v))))


def norm_scale(img, imgOri, norm, m):
    """the closing loop of normCorrection (:6380-6392): both stacks of image l times sqrt(m / norm(l)) (RFLOAT)"""
    f = np.sqrt(np.float32(m) / f32(norm)).astype(np.float32)
    return (c64(img) * f[:, None, None]).astype(np.complex64), (c64(imgOri) * f[:, None, None]).astype(np.complex64)


def sigma_accum(spec, groupID, nGroup, group=True):
    """Group accumulation of allReduceSigma (src/Optimiser.cpp:6567-6597); groupID 1-based.
    Returns sigM, sigN, svd as float32 [nGroup][rSig+1]."""
    spec, groupID = f32(spec), i32(groupID)
    nImg, _, rSig = spec.shape
    acc = [np.zeros((nGroup, rSig + 1), np.float32) for _ in range(3)]
    lib().orc_sigma_accum(_p(acc[0], c_f), _p(acc[1], c_f), _p(acc[2], c_f), _p(spec, c_f), _p(groupID, c_i),
                          C.c_int(nImg), C.c_int(nGroup), C.c_int(rSig), C.c_int(1 if group else 0))
    return acc


def sigma_final(sigM, sigN, svd, maskRadius, size, pixelSize, group=True):
    """Closing arithmetic of allReduceSigma (src/Optimiser.cpp:6654-6707) -> sig, sigRcp [nGroup][rSig]."""
    sigM, sigN, svd = f32(sigM).copy(), f32(sigN).copy(), f32(svd).copy()
    nGroup, nc = sigM.shape
    rSig = nc - 1
    sig = np.zeros((nGroup, rSig), np.float32)
    rcp = np.zeros((nGroup, rSig), np.float32)
    lib().orc_sigma_final(_p(sig, c_f), _p(rcp, c_f), _p(sigM, c_f), _p(sigN, c_f), _p(svd, c_f), C.c_int(nGroup),
                          C.c_int(rSig), C.c_int(1 if group else 0), C.c_float(maskRadius), C.c_int(size),
                          C.c_float(pixelSize))
    return sig, rcp


def expect_precal(attr, N, pixelSize, iCol, iRow):
    """allocPreCal ctf=true branch (src/Optimiser.cpp:8124-8169): -> freq [nPxl], def [nImg][nPxl], K1, K2 [nImg]"""
    attr, iCol, iRow = f32(attr).reshape(-1, 7), i32(iCol), i32(iRow)
    nImg, nPxl = len(attr), len(iCol)
    freq = np.zeros(nPxl, np.float32)
    de = np.zeros((nImg, nPxl), np.float32)
    k1 = np.zeros(nImg, np.float32)
    k2 = np.zeros(nImg, np.float32)
    lib().orc_expect_precal(_p(freq, c_f), _p(de, c_f), _p(k1, c_f), _p(k2, c_f), _p(attr, c_f), C.c_int(nImg),
                            C.c_int(N), C.c_float(pixelSize), _p(iCol, c_i), _p(iRow, c_i), C.c_int(nPxl))
    return freq, de, k1, k2


def ctf_dsearch(freq, de, K1, K2, phaseShift, ampC, d):
    """defocus-search CTF rows of one image (src/Optimiser.cpp:1246-1272) -> [nD][nPxl]"""
    freq, de, d = f32(freq), f32(de), f64(d)
    out = np.zeros((len(d), len(freq)), np.float32)
    lib().orc_ctf_dsearch(_p(out, c_f), _p(freq, c_f), _p(de, c_f), C.c_float(K1), C.c_float(K2),
                          C.c_float(phaseShift), C.c_float(ampC), _p(d, c_d), C.c_int(len(d)), C.c_int(len(freq)))
    return out


def ctf_image(N, pixelSize, attr):
    """CTF(Image&, ...) src/CTF.cpp:31-66 -> complex64 [N][N/2+1]"""
    out = np.zeros((N, N // 2 + 1), np.complex64)
    lib().orc_ctf_image(_p(out, c_f), C.c_int(N), C.c_float(pixelSize), *[C.c_float(a) for a in attr])
    return out


# ---------------------------------------------------------------------------------------------
# MRC stacks and image ingestion
def mrc_read(path):
    """ImageFile::readMetaDataMRC / readImageMRC / readVolumeMRC (src/Image/ImageFile.cpp:209-300): returns float32
    [nz][ny][nx] in the reference's in-memory (wrapped-origin) layout: IMAGE_READ_CAST / VOLUME_READ_CAST move the file's
    centre-origin samples with MESH_*_INDEX (include/Image/ImageFile.h:383-388,418-435)."""
    with open(path, "rb") as f:
        head = f.read(1024)
        nx, ny, nz, mode = np.frombuffer(head, np.int32, 4)
        nsymbt = int(np.frombuffer(head, np.int32, 1, 92)[0])
        f.seek(1024 + nsymbt)
        dt = {0: np.int8, 1: np.int16, 2: np.float32}[int(mode)]
        raw = np.fromfile(f, dt, int(nx) * int(ny) * int(nz)).reshape(nz, ny, nx)
    return raw.astype(np.float32), (int(nx), int(ny), int(nz), int(mode), nsymbt)


def mrc_images(path):
    """every slice read as an Image: dst(i, j) = file[(j + nRow/2) % nRow][(i + nCol/2) % nCol]"""
    raw, _ = mrc_read(path)
    return np.ascontiguousarray(np.roll(raw, (-(raw.shape[1] // 2), -(raw.shape[2] // 2)), axis=(1, 2)))


def mrc_volume(path):
    raw, _ = mrc_read(path)
    return np.ascontiguousarray(np.roll(raw, tuple(-(s // 2) for s in raw.shape), axis=(0, 1, 2)))


def init_images(rl, r, ew=6.0):
    """Optimiser::initImg after reading (src/Optimiser.cpp:4700-4800) for one rank holding all N images:
    substractBgImg -> statImg -> maskImg (zeroMask) -> normaliseImg -> fwImg.  rl float32 [n][N][N] wrapped layout.
    Returns imgFT, imgOriFT (complex64 [n][N][N/2+1]) and dict(mean, stdN, stdD, stdS, stdStdN)."""
    lib().orc_stddev.restype = C.c_float
    rl = f32(rl).copy()
    n, N = rl.shape[0], rl.shape[1]
    st = np.zeros((n, 4), np.float64)
    for l in range(n):
        lib().orc_subtract_bg(_p(rl[l], c_f), C.c_int(N), C.c_float(r))
        lib().orc_stat_img(_p(st[l], c_d), _p(rl[l], c_f), C.c_int(N), C.c_float(r))
    # RFLOAT accumulators of the OpenMP reduction (:4815-4821), then /N (:4902-4912)
    acc = [np.float32(0)] * 4
    for l in range(n):
        for q in range(4):
            acc[q] = np.float32(acc[q] + np.float32(st[l, q]))
    mean, stdN, stdD, stdStdN = [np.float32(a / np.float32(n)) for a in acc]
    stdS = np.float32(stdD - stdN)
    stdStdN = np.float32(np.sqrt(np.float64(stdStdN) - np.float64(np.float32(np.float64(stdN) ** 2))))
    ori = rl.copy()
    msk = np.empty_like(rl)
    for l in range(n):
        lib().orc_soft_mask_bg(_p(msk[l], c_f), _p(rl[l], c_f), C.c_int(N), C.c_float(r), C.c_float(ew), C.c_float(0))
    scale = np.float32(1.0 / np.float64(stdN))
    msk *= scale
    ori *= scale
    return (sfft.rfft2(msk).astype(np.complex64), sfft.rfft2(ori).astype(np.complex64),
            dict(mean=float(mean), stdN=float(stdN), stdD=float(stdD), stdS=float(stdS), stdStdN=float(stdStdN)))


# ---------------------------------------------------------------------------------------------
# The particle filter of the local search (src/Particle.cpp) with its random draws as inputs
def _dp(a):
    return a.ctypes.data_as(c_d)


def _symq(symQuat):
    sq = np.zeros((0, 4)) if symQuat is None else f64(symQuat).reshape(-1, 4)
    return sq, len(sq)


def pf_perturb(q, t, k, s, pfR, pfT, transS, transQ, gR, gT, symQuat=None):
    """Particle::perturb(pf, PAR_R) + perturb(pf, PAR_T) of one image (src/Optimiser.cpp:1186-1208, src/Particle.cpp:
    1149-1272): q [nR][4], t [nT][2] support points, k (k1, k2, k3), s (s0, s1) of the last calVari; gR [nR][4], gT [nT][4]
    standard normals (the draws).  Returns new q, t and the balanced priors wR, wT (balanceWeight + normW).  symQuat [nSym][4]
    (Symmetry::quat): perturb(PAR_R) ends with symmetrise(&mean) (:1234)."""
    q, t, gR, gT = f64(q).copy(), f64(t).copy(), f64(gR), f64(gT)
    nR, nT = len(q), len(t)
    L = lib()
    sq, nSym = _symq(symQuat)
    L.orc_perturb_R_sym(_dp(q), C.c_int(nR), _dp(f64(k)), C.c_double(pfR), _dp(gR), _p(sq, c_d), C.c_int(nSym))
    wR = np.zeros(nR)
    L.orc_balance_weight_R(_dp(wR), _dp(q), C.c_int(nR))
    L.orc_perturb_T(_dp(t), C.c_int(nT), C.c_double(s[0]), C.c_double(s[1]), C.c_double(pfT), C.c_double(transS),
                    C.c_double(transQ), _dp(gT))
    wT = np.zeros(nT)
    L.orc_balance_weight_T(_dp(wT), _dp(t), C.c_int(nT))
    return q, t, wR, wT


def pf_resample(val, w, u, rank, u0, nOut=None):
    """Particle::resample(n, pt) for one parameter (src/Particle.cpp:1291-1430, PARTICLE_PRIOR_ONE): shuffle (rank[i] = new
    position of element i, the draw), _topX = element of the largest u, w *= u, systematic resampling with the draw u0 in
    [0, 1 / n).  Returns (values [n], weights [n], source index of every output in the UNSHUFFLED order, index of the top)."""
    n = len(w)
    nOut = n if nOut is None else int(nOut)       # resample(n', pt) with n' < n: the support points after a global scan
    inv = np.empty(n, np.int64)
    inv[np.asarray(rank)] = np.arange(n)          # shuffled[j] = original[inv[j]]
    ws, us = np.ascontiguousarray(f64(w)[inv]), np.ascontiguousarray(f64(u)[inv])
    top = int(inv[int(np.argmax(us))])            # d_value_max_index on the shuffled list: first maximum
    idx = np.zeros(nOut, np.int32)
    wo = np.zeros(nOut)
    lib().orc_resample(_p(idx, c_i), _dp(wo), _dp(ws), _dp(us), C.c_int(n), C.c_int(nOut), C.c_double(float(u0)))
    src = inv[idx]
    return val[src].copy(), wo, src, top


def pf_scan_support(gridR, gridT, uR, uT, peakFactorR, mLR, mLT, rankR, u0R, rankT, u0T, minK=0.0, minS=0.0, symQuat=None, iAnchor=0):
    """The filter of one image after a global scan, src/Optimiser.cpp:953-1008: the scanned grid with uniform priors and the scan
    weights uR [nRin] / uT [nTin] (RFLOAT) -> keepHalfHeightPeak(PAR_R) (OPTIMISER_PEAK_FACTOR_R; _T off), resample(mLR, PAR_R),
    resample(mLT, PAR_T), calVari(PAR_R), calVari(PAR_T), k = max(minK, k), s = max(minS, s).  u0R in [0, 1 / mLR), u0T in [0, 1 / mLT).  Returns dict(q, t, wR, wT, k, s,
    topR, topT, srcR, srcT, uRk)."""
    L = lib()
    gridR, gridT = f64(gridR), f64(gridT)
    nR, nT = len(gridR), len(gridT)
    u = f64(np.asarray(uR, np.float32).astype(np.float64)).copy()
    L.orc_keep_half_height_peak(_dp(u), C.c_int(nR), C.c_double(peakFactorR))
    ut = f64(np.asarray(uT, np.float32).astype(np.float64)).copy()
    q2, wR2, srcR, topR = pf_resample(gridR, np.full(nR, 1.0 / nR), u, rankR, u0R, nOut=mLR)
    t2, wT2, srcT, topT = pf_resample(gridT, np.full(nT, 1.0 / nT), ut, rankT, u0T, nOut=mLT)
    q2, t2 = np.ascontiguousarray(q2), np.ascontiguousarray(t2)
    k, mean, s = np.zeros(3), np.zeros(4), np.zeros(2)
    sq, nSym = _symq(symQuat)
    L.orc_cal_vari_R_sym(_dp(k), _dp(mean), _dp(q2), C.c_int(mLR), _p(sq, c_d), C.c_int(nSym), C.c_int(int(iAnchor)))   # (symmetrises, then) rotates q to the mean frame and back, in place
    L.orc_cal_vari_T(_dp(s), _dp(t2), C.c_int(mLT))
    k, s = np.maximum(k, minK), np.maximum(s, minS)                 # setK1..3 / setS0, S1 with the scan's minimum spread, :1032-1079
    return dict(q=q2, t=t2, wR=wR2, wT=wT2, k=k, s=s, topR=gridR[topR].copy(), topT=gridT[topT].copy(), srcR=srcR, srcT=srcT, uRk=u)


def pf_update(q, t, wR, wT, uR, uT, peakFactorR, rankR, u0R, rankT, u0T, symQuat=None, iAnchor=0):
    """The filter bookkeeping after the likelihoods of a phase, src/Optimiser.cpp:1408-1475: setUR, keepHalfHeightPeak(PAR_R)
    (OPTIMISER_PEAK_FACTOR_R; _T off), setUT, calRank1st, calVari(PAR_R / PAR_T), resample(mLR, PAR_R), resample(mLT, PAR_T).
    uR / uT are the E-step weights (RFLOAT).  Returns dict(q, t, wR, wT, k, s, topR, topT, srcR, srcT)."""
    L = lib()
    q, t = f64(q).copy(), f64(t).copy()
    nR, nT = len(q), len(t)
    u = f64(np.asarray(uR, np.float32).astype(np.float64)).copy()
    L.orc_keep_half_height_peak(_dp(u), C.c_int(nR), C.c_double(peakFactorR))
    ut = f64(np.asarray(uT, np.float32).astype(np.float64)).copy()
    k, mean, s = np.zeros(3), np.zeros(4), np.zeros(2)
    # with a point group calVari first replaces the support points by their counterparts next to a random anchor (iAnchor = the
    # draw); resample() then takes _topR again from those (src/Particle.cpp:1340-1345)
    sq, nSym = _symq(symQuat)
    L.orc_cal_vari_R_sym(_dp(k), _dp(mean), _dp(q), C.c_int(nR), _p(sq, c_d), C.c_int(nSym), C.c_int(int(iAnchor)))   # rotates q to the mean frame and back, in place
    L.orc_cal_vari_T(_dp(s), _dp(t), C.c_int(nT))
    q2, wR2, srcR, topR = pf_resample(q, wR, u, rankR, u0R)
    t2, wT2, srcT, topT = pf_resample(t, wT, ut, rankT, u0T)
    return dict(q=q2, t=t2, wR=wR2, wT=wT2, k=k, s=s, topR=q[topR].copy(), topT=t[topT].copy(), srcR=srcR, srcT=srcT,
                iTopR=topR, iTopT=topT, qPre=q, uRk=u, uTk=ut)


def pf_class_select(uC, wC, peakFactorC, rank, u0, pick):
    """the class an image continues with after the global scan, src/Optimiser.cpp:925-952: setUC, setPeakFactor(PAR_C)
    (PARTICLE_PEAK_FACTOR_C: 1 - 1e-2), keepHalfHeightPeak(PAR_C), resample(k, PAR_C) (src/Particle.cpp:1296-1338), rand(cls)
    (:2109-2120).  Draws as inputs: rank (shuffle), u0 in [0, 1 / k), pick in [0, k)."""
    k = len(uC)
    u = f64(np.asarray(uC, np.float32).astype(np.float64)).copy()
    lib().orc_keep_half_height_peak(_dp(u), C.c_int(k), C.c_double(peakFactorC))
    cls, _, src, _ = pf_resample(np.arange(k), f64(wC), u, rank, u0)
    return int(cls[int(pick)])


def pf_perturb_d(d, s, scale, init, g):
    """Particle::initD (init: d = 1 + N(0, scale^2), src/Particle.cpp:281-311) or perturb(scale, PAR_D) (d += N(0, s^2) scale,
    :1273-1287), then balanceWeight(PAR_D) + normW (:2405-2440).  g [nD] standard normals.  Returns d, wD."""
    g = f64(g)
    d = (1.0 + scale * g) if init else (f64(d) + (s * g) * scale)
    n = len(d)
    m = float(np.mean(d))
    sd = float(np.std(d, ddof=1)) if n > 1 else 0.0
    if sd == 0:
        w = np.ones(n)
    else:
        u = (d - m) / abs(sd)
        w = 1.0 / ((1.0 / (np.sqrt(2 * np.pi) * abs(sd))) * np.exp(-u * u / 2))      # gsl_ran_gaussian_pdf
    return d, w / w.sum()


def pf_update_d(d, wD, uD, rank, u0):
    """setUD, calRank1st(PAR_D), calVari(PAR_D), resample(mLD, PAR_D) (src/Optimiser.cpp:1424-1470; OPTIMISER_PEAK_FACTOR_D off)"""
    d = f64(d)
    u = np.asarray(uD, np.float32).astype(np.float64)
    top = float(d[int(np.argmax(u))])
    s = float(np.std(d, ddof=1)) if len(d) > 1 else 0.0
    d2, w2, src, _ = pf_resample(d, wD, u, rank, u0)
    return d2, w2, s, top, src


def cal_vari(q, t, symQuat=None, iAnchor=0, return_q=False):
    """Particle::calVari(PAR_R) + (PAR_T) -> (k [3], s [2]) (q is rotated to the mean frame and back: a copy here; with a point
    group the copy is symmetrised about q[iAnchor] first and return_q hands it back)"""
    q = f64(q).copy()
    k, mean, s = np.zeros(3), np.zeros(4), np.zeros(2)
    sq, nSym = _symq(symQuat)
    lib().orc_cal_vari_R_sym(_dp(k), _dp(mean), _dp(q), C.c_int(len(q)), _p(sq, c_d), C.c_int(nSym), C.c_int(int(iAnchor)))
    lib().orc_cal_vari_T(_dp(s), _dp(f64(t)), C.c_int(len(t)))
    return (k, s, q) if return_q else (k, s)


def soft_mask_volume(vol, r, ew, bg=0.0):
    """softMask(Volume& dst, const Volume& src, r, ew, bg) src/Functions/Mask.cpp:499-521 (returns a new array)"""
    out = f32(vol).copy()
    lib().orc_soft_mask_volume(_p(out, c_f), C.c_int(out.shape[0]), C.c_float(r), C.c_float(ew), C.c_float(bg))
    return out


# ---------------------------------------------------------------------------------------------
CALLS_PER_ITER, SLOT_CLASS, SLOT_SUPPORT, SLOT_RESET, SLOT_PHASE0, SLOT_DRAWS, SLOT_BALANCE = 1024, 1, 2, 3, 8, 1000, 1001


class Iteration:
    """One EM iteration, chained exactly as the reference runs it (MODE_3D; K >= 1 classes; local or global search; any point
    group; no CTF search):

      Optimiser::expectation   src/Optimiser.cpp:631-1140    global search only: scan of every image against K classes x nR rotations
                                                             x nT shifts at r = rScan (expect_global), class of the image
                                                             (pf_class_select), support points (pf_scan_support)
                               :1141-1660                    allocPreCal rows, then per image and phase: perturb -> slices x
                                                             ramps -> logDataVSPrior -> weights -> setU / keepHalfHeightPeak /
                                                             calRank1st / calVari / resample (phase index from 1 and
                                                             perturbFactorSGlobal throughout after a scan, :1185-1212)
      Optimiser::maximization  :3405-3530                    normCorrection (:6201-6394; not in the first iteration, not after a
                                                             global search), allReduceSigma (:6395-6710), reconstructRef
                                                             (:6711-7766): mReco Particle::rand draws -> translate -> insertP into the
                                                             image's class; prepareTF (normalise, symmetrizeT, symmetrizeF,
                                                             src/Reconstructor.cpp:1056-1091); per class reconstruct (MAP off) ->
                                                             [balanceClass] -> compareTwoHemispheres(fsc) -> reconstruct (MAP on,
                                                             joinHalf, the FSC Model::resetReco set at the END OF THE PREVIOUS
                                                             iteration) -> [balanceClass] -> compareTwoHemispheres(avg)
      Optimiser::run           :3800-4073                    reCentreImg, reMaskImg (not after a global search), solventFlatten,
                                                             Model::refreshProj, Model::resetReco

    Random draws are inputs: `ph` offers draw_n4 / draw_u4 / shuffle_ranks(seed, image, call, purpose, index) (tests hand in
    their numpy replica of the device's Philox streams, tests/_philox.py).  Numbering as thx_refine_iterate's: image = the
    image's index, call = iteration * 1024 + slot (1 class selection, 2 support points, 8 + 2 p / 9 + 2 p perturb / update of
    phase index p, 1000 insertion draws, 1001 balanceClass; 3 = the calVari of reset).

    cfg: dict(N, pf, nHalfA, mLR, mLT, nPhase, mReco, batch, rL, nGroup, groupSig, pixelSize, maskRadiusPx, sigma2Init,
    transS, transQ, pfL, pfS, peakFactorR, seed, coreFSC, goldenAverage, solventFlatten) and, optional: normCorrection, nK,
    rScan, pfSGlobal, peakFactorC, scanMinK, scanMinS, balanceClass, sym (= symmetry(name) or None).  ref: [N]^3 or [K][N]^3.
    `resolve` (optional): callback(phase, image, own) -> own, lets a test adopt the device's choice where a discrete decision
    (resampled indices, top support point) hinges on rounding -- after checking its tie rule; if it has a method
    after_perturb(phase, image, q_in, q, t, wR, wT) -> (q, t, wR, wT) it is also shown every perturbed cloud (the mean frame
    of Particle::perturb is numerically undetermined when the resampled cloud has collapsed onto a few points); after_scan(image,
    own) -> own is shown the class and the support points a global scan gave an image."""

    def __init__(self, cfg, imgOri, attr, gid, quat0, tran0, ref, ph, grid=None, cls0=None):
        self.c = dict(cfg)
        c = self.c
        for k_, v_ in (("normCorrection", 0), ("nK", 1), ("rScan", 0), ("pfSGlobal", 0.5), ("peakFactorC", 1.0 - 1e-2), ("scanMinK", 0.0),
                       ("scanMinS", 0.0), ("balanceClass", 0), ("sym", None)):
            c.setdefault(k_, v_)
        self.ph = ph
        N, pf = c["N"], c["pf"]
        self.N, self.pf, self.P = N, pf, N * pf
        self.rU, self.rSig = N // 2 - 2, N // 2 - 1
        self.K = max(1, int(c["nK"]))
        self.imgOri = c64(imgOri).copy()
        self.n = self.imgOri.shape[0]
        self.attr = f32(attr).reshape(self.n, 7)
        self.gid = i32(gid)
        self.q0, self.t0 = f64(quat0), f64(tran0)
        self.ref = f32(ref).reshape(self.K, N, N, N)
        self.symQ = None if not c["sym"] or c["sym"]["n"] == 0 else c["sym"]["quat"]
        self.symR = None if self.symQ is None else c["sym"]["R"]
        nA = c["nHalfA"]
        self.ranges = [(0, nA), (nA, self.n)]
        self.mask2d = soft_mask(N, np.float32(c["maskRadiusPx"]), 6.0)
        self.have_grid = grid is not None
        if grid is not None:   # Particle::reset(k, nR, nT, 1): the drawn rotations, symmetrise()d next to ANCHOR_POINT_2
            gq, gt = grid
            self.gridR = f64(gq) if self.symQ is None else symmetrise(gq, self.symQ, None)
            self.gridT = f64(gt)
        self.cls0 = np.zeros(self.n, np.int32) if cls0 is None else i32(cls0)
        self.mLD = int(c.get("mLD", 0))
        self.iterCount = 0
        self.set_cutoff(self.rU, self.rU)                      # Nyquist: r = rU = N / 2 - 2, _size = _N
        self.reset()

    def set_cutoff(self, r, rU, a=1.9):
        """The frequency cut-offs of the NEXT iteration, as Model::updateR / updateRU leave them before it starts (the schedule itself is
        the caller's): r = Optimiser::_r -- the expectation's pixel list allocPreCalIdx(_r, _rL) (src/Optimiser.cpp:631,1693; a global
        search scans on it too), Projector::_maxRadius (Model::refreshProj, src/Model.cpp:1042: what allReduceSigma's and
        normCorrection's slices are cut at) and normCorrection's rNorm = min(_r, .) (:6203) --; rU = Model::_rU -- the
        reconstruction's pixel list allocPreCalIdx(rU, 0) (:6722-6741), Reconstructor::_maxRadius and its grid
        _size = min(N, (rU + ceil(a)) * 2) (Model::resetReco, src/Model.cpp:1100-1125, Reconstructor::resizeSpace
        src/Reconstructor.cpp:184-198), the shells of compareTwoHemispheres' FSC.  thx_refine_set_cutoff's twin."""
        c, N, pf = self.c, self.N, self.pf
        assert c["rL"] < r <= N // 2 - 1 and 0 < rU <= N // 2 - 1
        self.rE, self.rU = int(r), int(rU)
        self.size = min(N, (self.rU + int(np.ceil(a))) * 2)
        self.PF = pf * self.size
        self.pl = pixel_list(N, self.rE, c["rL"], pf)          # expectation: allocPreCalIdx(_r, _rL), :631
        self.plM = pixel_list(N, self.rU, 0, pf)               # reconstruction: allocPreCalIdx(rU, 0), :6722
        self.ctfM = np.stack([ctf(c["pixelSize"], *self.attr[l], N, self.plM["iCol"], self.plM["iRow"]) for l in range(self.n)])
        self.ctfP = np.stack([ctf(c["pixelSize"], *self.attr[l], N, self.pl["iCol"], self.pl["iRow"]) for l in range(self.n)])
        self.plS = None
        if self.have_grid:      # the scan runs on the expectation's list: the radius of the scan = min(cfg rScan, r)
            self.rS = min(int(c["rScan"]), self.rE)
            self.plS = pixel_list(N, self.rS, c["rL"], pf)
            self.ctfS = np.stack([ctf(c["pixelSize"], *self.attr[l], N, self.plS["iCol"], self.plS["iRow"]) for l in range(self.n)])
        if self.mLD:   # allocPreCal(.., ctf = true): the pre-calculated rows of the defocus search, :8124-8169
            self.freqD, self.defD, self.K1, self.K2 = expect_precal(self.attr, N, c["pixelSize"], self.pl["iCol"], self.pl["iRow"])
        if self.iterCount == 0:                                # Model::initProjReco: setFSC(vec::Constant(_rU, 1)), src/Model.cpp:1086
            self.fscReco = np.ones((self.K, self.rU), np.float32)

    # -- helpers ------------------------------------------------------------------------------
    def _remask(self, imgs):
        out = np.empty_like(imgs)
        N = self.N
        for l in range(imgs.shape[0]):
            rl = np.ascontiguousarray(sfft.irfft2(imgs[l], s=(N, N), norm="forward").astype(np.float32))
            lib().orc_scale_mul_rl(_p(rl, c_f), _p(self.mask2d, c_f), C.c_size_t(rl.size))
            out[l] = sfft.rfft2(rl).astype(np.complex64)
        return out

    def call(self, slot):
        return self.iterCount * CALLS_PER_ITER + slot

    def attr_d(self, l, d):
        """_ctfAttr[l] with defocusU * d, defocusV * d (double products narrowed to RFLOAT, src/Optimiser.cpp:6534-6545,7183-7197)"""
        a = self.attr[l].copy()
        a[1] = np.float32(np.float64(a[1]) * d)
        a[2] = np.float32(np.float64(a[2]) * d)
        return a

    def _anchor(self, l, call, n):
        """anch = _r.row(gsl_rng_uniform_int(engine, _nR)) of calVari (src/Particle.cpp:1030): Philox purpose 13"""
        return min(int(self.ph.draw_u4(self.c["seed"], l, call, 13, 0)[0] * n), n - 1)

    def reset(self):
        """the state before the first iteration (thx_refine_reset): Optimiser::initImg has masked the images, flat noise
        model, support points as loaded, weights uniform, Particle::load -> calVari"""
        c = self.c
        self.iterCount = 0
        self.vols = [[set_projectee(self.ref[k], self.pf) for k in range(self.K)] for _ in range(2)]
        self.offset = np.zeros((self.n, 2))
        self.img = self._remask(self.imgOri)
        self.sig = np.full((2, c["nGroup"], self.rSig), np.float32(c["sigma2Init"]), np.float32)
        self.sigRcp = np.full((2, c["nGroup"], self.rSig), np.float32(-0.5) / np.float32(c["sigma2Init"]), np.float32)
        self.q, self.t = self.q0.copy(), self.t0.copy()
        self.k = np.zeros((self.n, 3))
        self.s = np.zeros((self.n, 2))
        for l in range(self.n):
            self.k[l], self.s[l], self.q[l] = cal_vari(self.q[l], self.t[l], self.symQ, self._anchor(l, SLOT_RESET, c["mLR"]), return_q=True)
            if self.symQ is None:
                self.q[l] = self.q0[l]      # (the round trip through the mean frame stays inside calVari's copy on the device)
        self.topR, self.topT = self.q[:, 0].copy(), self.t[:, 0].copy()
        self.cls = self.cls0.copy()
        self.d = np.ones((self.n, max(1, self.mLD)))
        self.wD = np.full((self.n, max(1, self.mLD)), 1.0 / max(1, self.mLD))
        self.sD, self.topD = np.zeros(self.n), np.ones(self.n)
        self.fscReco = np.ones((self.K, self.rU), np.float32)            # Model::initProjReco, src/Model.cpp:1086

    def fsc_of_maps(self, mapA, mapB, iterCount, k=0):
        """Model::compareTwoHemispheres(true, false) on the two MAP-off half maps (src/Model.cpp:307-612): FSC over rU shells,
        mask-corrected with the core mask when coreFSC (random phases: the Philox streams of iteration `iterCount`, class k)"""
        c, ph, N = self.c, self.ph, self.N
        A, B = sfft.rfftn(f32(mapA)).astype(np.complex64), sfft.rfftn(f32(mapB)).astype(np.complex64)
        coreR = float(int(np.rint(np.float32(c["maskRadiusPx"])))) if c["coreFSC"] else 0.0
        phA = phB = None
        if c["coreFSC"]:
            ne = N * N * (N // 2 + 1)
            call = 0x40000000 + 2 * (iterCount * 16 + k)
            e = np.arange(ne, dtype=np.uint64)
            lo32, hi32 = (e & np.uint64(0xFFFFFFFF)).astype(np.uint32), (e >> np.uint64(32)).astype(np.uint32)
            pi = 3.14159265358979323846   # TSGSL_ran_flat(engine, 0, 2 * M_PI) narrowed to RFLOAT; device: (float)(u * 2 * pi)
            phA = (ph.draw_u4(c["seed"], lo32, call, 9, hi32)[0] * 2 * pi).astype(np.float32)
            phB = (ph.draw_u4(c["seed"], lo32, call + 1, 9, hi32)[0] * 2 * pi).astype(np.float32)
        return compare_hemispheres(A, B, N, self.rU, phA, phB, coreR=coreR, ew=6.0)["fsc"]

    def balance_map(self, distr):
        """Optimiser::determineBalanceClass (src/Optimiser.cpp:5518-5584) -> bm [K] (-1 = the class keeps its own reference)"""
        K, c = self.K, self.c
        thres = 0.05 / K                                      # CLASS_BALANCE_FACTOR / _para.k
        cum = np.where(distr < thres, 0.0, distr - thres)
        tot = cum.sum()
        bm = np.full(K, -1, np.int64)
        if not tot > 0:
            return bm
        cum = np.cumsum(cum / tot)
        for t in range(K):
            if distr[t] < thres:
                ind = np.float32(self.ph.draw_u4(c["seed"], 0, self.call(SLOT_BALANCE), 14, t)[0])
                j = 0
                while j < K - 1 and cum[j] < ind:
                    j += 1
                bm[t] = j
        return bm

    # -- the stages -----------------------------------------------------------------------------
    def _scan(self, vi, lo, hi, resolve, out):
        """global search of the images [lo, hi) of half vi: scan over the K classes -> class -> support points"""
        c, ph, N, P, pf, K = self.c, self.ph, self.N, self.P, self.pf, self.K
        plS, seed, mLR, mLT = self.plS, c["seed"], c["mLR"], c["mLT"]
        nR, nT = len(self.gridR), len(self.gridT)
        n = hi - lo
        datS = np.ascontiguousarray(self.img[lo:hi].reshape(n, -1)[:, plS["iPxl"]])
        sigS = np.ascontiguousarray(self.sigRcp[vi][self.gid[lo:hi] - 1][:, plS["iSig"]])
        ctfS = self.ctfS[lo:hi]
        mats = np.stack([rotate3D(q) for q in self.gridR])
        traP = np.stack([translate(np.float32(s[0]), np.float32(s[1]), N, plS["iCol"], plS["iRow"]) for s in self.gridT])
        dat_pm, ctf_pm, sig_pm = (np.ascontiguousarray(a.T) for a in (datS, ctfS, sigS))
        wC, wR, wT = np.zeros((n, K), np.float32), np.zeros((K, n, nR), np.float32), np.zeros((K, n, nT), np.float32)
        base = np.full(n, np.nan, np.float32)                    # "unset", :737-745
        pR, pT = np.full((n, nR), 1.0 / nR), np.full((n, nT), 1.0 / nT)
        for k in range(K):
            rotP = np.stack([project(self.vols[vi][k], P, pf, m, plS["iCol"], plS["iRow"]) for m in mats])
            expect_global(rotP, traP, dat_pm, ctf_pm, sig_pm, K, k, pR, pT, wC, wR, wT, base)
        out["scanUC"][lo:hi], out["scanBase"][lo:hi] = wC, base
        out["scanUR"][lo:hi], out["scanUT"][lo:hi] = wR.transpose(1, 0, 2), wT.transpose(1, 0, 2)
        callC, callS = self.call(SLOT_CLASS), self.call(SLOT_SUPPORT)
        for l in range(lo, hi):
            li = l - lo
            own = dict(uC=wC[li], uR=wR[:, li], uT=wT[:, li], base=float(base[li]))
            if resolve is not None and hasattr(resolve, "scan_weights"):
                own = resolve.scan_weights(l, own)          # (checks the weights; hands back the device's, which the filter continues from)
            cls = pf_class_select(own["uC"], np.full(K, 1.0 / K), c["peakFactorC"], ph.shuffle_ranks(seed, l, callC, 6, K),
                                  ph.draw_u4(seed, l, callC, 7, 0)[0] / K, min(int(ph.draw_u4(seed, l, callC, 8, 0)[0] * K), K - 1))
            rankR, rankT = ph.shuffle_ranks(seed, l, callS, 2, nR), ph.shuffle_ranks(seed, l, callS, 4, nT)
            ws = pf_scan_support(self.gridR, self.gridT, own["uR"][cls], own["uT"][cls], c["peakFactorR"], mLR, mLT, rankR,
                                 ph.draw_u4(seed, l, callS, 3, 0)[0] / mLR, rankT, ph.draw_u4(seed, l, callS, 5, 0)[0] / mLT,
                                 c["scanMinK"], c["scanMinS"], symQuat=self.symQ, iAnchor=self._anchor(l, callS, mLR))
            ws.update(cls=cls, rankR=rankR, rankT=rankT)
            if resolve is not None and hasattr(resolve, "after_scan"):
                ws = resolve.after_scan(l, ws)
            self.cls[l] = ws["cls"]
            self.q[l], self.t[l], self.k[l], self.s[l] = ws["q"], ws["t"], ws["k"], ws["s"]
            self.topR[l], self.topT[l] = ws["topR"], ws["topT"]
            self.wR0[l], self.wT0[l] = ws["wR"], ws["wT"]

    def _expect(self, vi, lo, hi, glob, resolve, out, ctfs=False):
        c, ph, N, P, pf = self.c, self.ph, self.N, self.P, self.pf
        mLD = self.mLD
        pl, seed, mLR, mLT = self.pl, c["seed"], c["mLR"], c["mLT"]
        # allocPreCal(mask = true, pixelMajor = false, ctf = false), :8043-8171
        datP = np.ascontiguousarray(self.img[lo:hi].reshape(hi - lo, -1)[:, pl["iPxl"]])
        sigRcpP = np.ascontiguousarray(self.sigRcp[vi][self.gid[lo:hi] - 1][:, pl["iSig"]])
        p0 = 1 if glob else 0
        # the per-image stop rule (:1510-1615) when maxPhase > nPhase: from phase index nPhase (MIN_N_PHASE_PER_ITER_LOCAL / _GLOBAL) on
        # the image's variances are compared with the smallest seen so far after every phase; the driver's form of the reference's
        # `for phase < MAX_N_PHASE_PER_ITER ... break` (phases outermost, a mask over the images)
        rule = c.get("maxPhase", 0) > c["nPhase"]
        pEnd = c["maxPhase"] if rule else p0 + c["nPhase"]
        stop = {l: stop_rule_init(c["transS"], c.get("ctfRefineS", 0.01) if ctfs else 0.01) for l in range(lo, hi)}
        active = {l: True for l in range(lo, hi)}
        for p in range(p0, pEnd):
            pi = p - p0
            if not any(active.values()):
                break
            callP = self.call(SLOT_PHASE0 + 2 * p)
            callU = callP + 1
            f = c["pfL"] if p == 0 else (c["pfSGlobal"] if glob else c["pfS"])
            for l in range(lo, hi):
                if not active[l]:
                    continue
                gR = np.stack(ph.draw_n4(seed, l, callP, 0, np.arange(mLR)), axis=1)
                gT = np.stack(ph.draw_n4(seed, l, callP, 1, np.arange(mLT)), axis=1)
                q, t, wR, wT = pf_perturb(self.q[l], self.t[l], self.k[l], self.s[l], f, f, c["transS"], c["transQ"], gR, gT, symQuat=self.symQ)
                if resolve is not None and hasattr(resolve, "after_perturb"):
                    if hasattr(resolve, "k_in"):
                        resolve.k_in[l] = (self.k[l].copy(), self.s[l].copy())
                    if self.symQ is not None and hasattr(resolve, "q_pure"):
                        # (for the checker: the same perturbation before symmetrise(&mean) replaced the points by their mates)
                        resolve.q_pure = pf_perturb(self.q[l], self.t[l], self.k[l], self.s[l], f, f, c["transS"], c["transQ"], gR, gT)[0]
                    q, t, wR, wT = resolve.after_perturb(pi, l, self.q[l], q, t, wR, wT)
                rot = np.stack([rotate3D(x) for x in q])
                if ctfs:
                    # phase 0: Particle::initD(mLD, ctfRefineS); later: perturb(perturbFactorSCTF, PAR_D) (:1196-1209); then the CTF rows
                    # of every defocus factor (:1246-1272)
                    gD = ph.draw_n4(seed, l, callP, 10, np.arange(mLD))[0]
                    dD, wDD = pf_perturb_d(self.d[l], self.sD[l], c["ctfRefineS"] if p == 0 else c["pfSCTF"], p == 0, gD)
                    ctfRows = ctf_dsearch(self.freqD, self.defD[l], self.K1[l], self.K2[l], self.attr[l][6], self.attr[l][5], dD)
                    e = expect_local(self.vols[vi][self.cls[l]], P, pf, N, pl["iCol"], pl["iRow"], datP[l - lo], ctfRows,
                                     sigRcpP[l - lo], rot, t, nD=mLD, pC=1.0, pR=wR, pT=wT, pD=wDD, cSearch=True)
                else:
                    e = expect_local(self.vols[vi][self.cls[l]], P, pf, N, pl["iCol"], pl["iRow"], datP[l - lo], self.ctfP[l],
                                     sigRcpP[l - lo], rot, t, nD=1, pC=1.0, pR=wR, pT=wT)
                iA = self._anchor(l, callU, mLR)
                own = pf_update(q, t, wR, wT, e["wR"], e["wT"], c["peakFactorR"],
                                ph.shuffle_ranks(seed, l, callU, 2, mLR), ph.draw_u4(seed, l, callU, 3, 0)[0] / mLR,
                                ph.shuffle_ranks(seed, l, callU, 4, mLT), ph.draw_u4(seed, l, callU, 5, 0)[0] / mLT,
                                symQuat=self.symQ, iAnchor=iA)
                own.update(uR=e["wR"], uT=e["wT"], tPre=t, qIn=q, wRIn=wR, wTIn=wT, li=l, callU=callU, iAnchor=iA,
                           scaleL=float(np.abs(e["logW"]).max()))
                if ctfs:   # setUD, calRank1st(PAR_D), calVari(PAR_D), resample(mLD, PAR_D), :1424-1470
                    d2, wD2, sD, topD, srcD = pf_update_d(dD, wDD, e["wD"], ph.shuffle_ranks(seed, l, callU, 11, mLD),
                                                          ph.draw_u4(seed, l, callU, 12, 0)[0] / mLD)
                    own.update(uD=e["wD"], dIn=dD, wDIn=wDD, d=d2, wD=wD2, sD=sD, topD=topD, srcD=srcD)
                if resolve is not None:
                    own = resolve(pi, l, own)
                if ctfs:
                    self.d[l], self.wD[l], self.sD[l], self.topD[l] = own["d"], own["wD"], own["sD"], own["topD"]
                    out["uD"][pi, l], out["dP"][pi, l] = e["wD"], dD
                self.q[l], self.t[l], self.k[l], self.s[l] = own["q"], own["t"], own["k"], own["s"]
                self.topR[l], self.topT[l] = own["topR"], own["topT"]
                out["uR"][pi, l], out["uT"][pi, l] = e["wR"], e["wT"]
                out["srcR"][pi, l], out["srcT"][pi, l] = own["srcR"], own["srcT"]
                out["k"][pi, l], out["s"][pi, l] = own["k"], own["s"]
                out["phases"][l] = pi + 1
                kS, sS = own.get("k_stop", own["k"]), own.get("s_stop", own["s"])      # (a checker may hand in the variances to decide on)
                if rule and p >= c["nPhase"] and stop_rule(stop[l], kS[0], kS[1], kS[2], sS[0], sS[1], float(self.sD[l]) if ctfs else 0.0):
                    active[l] = False
                    out["nP"][l] = p

    def _norm_correction(self, out):
        """Optimiser::normCorrection, src/Optimiser.cpp:6201-6394: residual power of every masked image against its top pose's
        CTF-modulated slice over rL <= r < rNorm, median over ALL particles, both stacks rescaled, rows cut again"""
        c, N, P, pf = self.c, self.N, self.P, self.pf
        res = max(res_p(self.fscReco[k], 0.75, 1, 1, False) for k in range(self.K))   # Model::resolutionP(0.75, false), src/Model.cpp:984-994
        rNorm = float(min(self.rE, res))                        # TSGSL_MIN_RFLOAT(_r, _model.resolutionP(0.75, false)), :6203
        norm = np.zeros(self.n, np.float32)
        for vi, (lo, hi) in enumerate(self.ranges):
            for l in range(lo, hi):
                norm[l] = norm_residual(self.vols[vi][self.cls[l]], P, pf, N, self.rE, float(c["rL"]), rNorm, rotate3D(self.topR[l]),
                                        self.topT[l], c["pixelSize"], self.attr[l], self.img[l])
        m = median(norm)
        self.img, self.imgOri = norm_scale(self.img, self.imgOri, norm, m)
        out.update(norm=norm, normMedian=m, rNorm=rNorm)

    def _reconstruct_all(self, F, T, force, bm):
        """reconstructRef after prepareTF (src/Optimiser.cpp:7326-7747): per class reconstruct with MAP off -> balanceClass ->
        compareTwoHemispheres(fsc) -> reconstruct with MAP on (Reconstructor::_FSC of the previous iteration, joinHalf) ->
        balanceClass -> compareTwoHemispheres(avg) -> solventFlatten.  T [2][K] is changed in place as the reference changes
        _T3D.  force: None = the reference's stop rule, else the round counts to run [MAP off / on][half][class]."""
        c, N, P, pf, K = self.c, self.N, self.PF, self.pf, self.K          # (P here: PAD_SIZE = _pf * _size of the reconstructors)
        empty = [[not (T[vi][k].flat[0] > 0) for k in range(K)] for vi in range(2)]
        maps = [[np.zeros((N, N, N), np.float32) for _ in range(K)] for _ in range(2)]
        rounds = np.zeros((2, 2, K), np.int64)
        for vi in range(2):
            for k in range(K):
                if empty[vi][k]:
                    continue
                m, it, diffs, _ = reconstruct(F[vi][k], T[vi][k], P, N, pf, self.rU, MAP=False, joinHalf=True, gridCorr=True, return_iters=True,
                                              T_inplace=True, force_rounds=None if force is None else int(force[0][vi][k]))
                maps[vi][k], rounds[0, vi, k] = m, it

        def balance(mm):
            for vi in range(2):
                for t in range(K):
                    if bm[t] >= 0 and bm[t] != t:
                        mm[vi][t] = mm[vi][bm[t]].copy()
        balance(maps)
        res = dict(mapsFsc=[[m.copy() for m in h] for h in maps],
                   fsc=np.stack([self.fsc_of_maps(maps[0][k], maps[1][k], self.iterCount, k) for k in range(K)]))
        mapsX = [[np.zeros((N, N, N), np.float32) for _ in range(K)] for _ in range(2)]
        for vi in range(2):
            for k in range(K):
                if empty[vi][k]:
                    continue
                m, it, diffs, _ = reconstruct(F[vi][k], T[vi][k], P, N, pf, self.rU, FSC=self.fscReco[k], joinHalf=True, MAP=True, gridCorr=True,
                                              return_iters=True, T_inplace=True, force_rounds=None if force is None else int(force[1][vi][k]))
                mapsX[vi][k], rounds[1, vi, k] = m, it
        balance(mapsX)
        res["mapsMAP"] = [[m.copy() for m in h] for h in mapsX]
        if c["goldenAverage"]:   # compareTwoHemispheres(false, true, AVERAGE_TWO_HEMISPHERE_THRES), :7747
            # one class under the gold standard: inside r = Model::resolutionP(0.95, false) of the FSC just computed
            # (MODEL_RESOLUTION_BASE_AVERAGE, include/Config.h:129-131, src/Model.cpp:616-674); several classes: everywhere (:688-696)
            avgR = res_p(res["fsc"][0], 0.95, 1, 1, False) if (K == 1 and c["goldenAverage"] == 1) else -1
            for k in range(K):
                A, B = sfft.rfftn(mapsX[0][k]).astype(np.complex64), sfft.rfftn(mapsX[1][k]).astype(np.complex64)
                cm = compare_hemispheres(A, B, N, self.rU, avg_r=avgR)
                mapsX[0][k], mapsX[1][k] = [np.ascontiguousarray(sfft.irfftn(x, s=(N, N, N)).astype(np.float32)) for x in (cm["A"], cm["B"])]
            res["avgR"] = avgR
        keep = [[empty[vi][k] and not (bm[k] >= 0 and bm[k] != k and not empty[vi][bm[k]]) for k in range(K)] for vi in range(2)]
        if c["solventFlatten"]:   # Optimiser::solventFlatten(false), :7958-7975
            mapsX = [[m if keep[vi][k] else soft_mask_volume(m, np.float32(c["maskRadiusPx"]), 6.0, 0.0) for k, m in enumerate(h)]
                     for vi, h in enumerate(mapsX)]
        res["maps"], res["rounds"], res["keep"] = mapsX, rounds, keep
        return res

    def iterate(self, resolve=None, force_rounds=None, search="local", device_FT=None):
        c, ph, N, P, pf, K = self.c, self.ph, self.N, self.P, self.pf, self.K
        plM = self.plM
        seed, mLR, mLT = c["seed"], c["mLR"], c["mLT"]
        glob, ctfs = search == "global", search == "ctf"
        nPh = max(c["nPhase"], c.get("maxPhase", 0))
        out = dict(uR=np.zeros((nPh, self.n, mLR), np.float32), uT=np.zeros((nPh, self.n, mLT), np.float32),
                   uD=np.zeros((nPh, self.n, max(1, self.mLD)), np.float32), dP=np.zeros((nPh, self.n, max(1, self.mLD))),
                   srcR=np.zeros((nPh, self.n, mLR), np.int64), srcT=np.zeros((nPh, self.n, mLT), np.int64),
                   k=np.zeros((nPh, self.n, 3)), s=np.zeros((nPh, self.n, 2)),
                   phases=np.zeros(self.n, np.int64), nP=np.zeros(self.n, np.int64))   # phases an image ran; phase index it stopped in
        if glob:
            nR, nT = len(self.gridR), len(self.gridT)
            out.update(scanUC=np.zeros((self.n, K), np.float32), scanUR=np.zeros((self.n, K, nR), np.float32),
                       scanUT=np.zeros((self.n, K, nT), np.float32), scanBase=np.zeros(self.n, np.float32))
            self.wR0, self.wT0 = np.zeros((self.n, mLR)), np.zeros((self.n, mLT))
        # ---- expectation of both halves (the M-step of a half does not feed the other half's E-step, and the draws are numbered
        # by iteration and phase: the order E0 M0 E1 M1 of a driver without normCorrection gives the same numbers) ----
        for vi, (lo, hi) in enumerate(self.ranges):
            if glob:
                self._scan(vi, lo, hi, resolve, out)
                out["r0"], out["t0"], out["k0"], out["s0"] = self.q.copy(), self.t.copy(), self.k.copy(), self.s.copy()
            self._expect(vi, lo, hi, glob, resolve, out, ctfs)
        out["cls"] = self.cls.copy()
        if c["normCorrection"] and self.iterCount != 0 and not glob:
            self._norm_correction(out)
        datM = np.ascontiguousarray(self.imgOri.reshape(self.n, -1)[:, plM["iPxl"]])
        PF = self.PF                                          # the reconstructors' grid: _pf * _size (Reconstructor::allocSpace after resizeSpace)
        F = [[np.zeros((PF, PF, PF // 2 + 1), np.complex64) for _ in range(K)] for _ in range(2)]
        T = [[np.zeros((PF, PF, PF // 2 + 1), np.float32) for _ in range(K)] for _ in range(2)]
        w = np.float32(np.float32(1.0) / np.float32(c["mReco"]))
        callD = self.call(SLOT_DRAWS)
        for vi, (lo, hi) in enumerate(self.ranges):
            # allReduceSigma (OPTIMISER_SIGMA_RANK1ST, OPTIMISER_SIGMA_WHOLE_FREQUENCY), :6395-6710
            spec = np.stack([sigma_image(self.vols[vi][self.cls[l]], P, pf, N, self.rE, self.rSig, rotate3D(self.topR[l]), self.topT[l],
                                         self.offset[l], c["pixelSize"], self.attr_d(l, self.topD[l]) if ctfs else self.attr[l], self.img[l],
                                         self.imgOri[l])
                             for l in range(lo, hi)])
            acc = sigma_accum(spec, self.gid[lo:hi], c["nGroup"], bool(c["groupSig"]))
            sig, rcp = sigma_final(*acc, np.float32(c["maskRadiusPx"]) * np.float32(c["pixelSize"]), N, c["pixelSize"],
                                   bool(c["groupSig"]))
            # HOT LOOP C, :7038-7241: Particle::rand = uniform picks among the (resampled) support points
            for l in range(lo, hi):
                u = ph.draw_u4(seed, l, callD, 7, np.arange(c["mReco"]))
                iR = np.minimum((u[0] * mLR).astype(np.int64), mLR - 1)
                iT = np.minimum((u[1] * mLT).astype(np.int64), mLT - 1)
                iD = np.minimum((u[2] * max(1, self.mLD)).astype(np.int64), max(1, self.mLD) - 1)
                kc = self.cls[l]
                for m in range(c["mReco"]):
                    tt = self.t[l, iT[m]] - self.offset[l]
                    src = translate(np.float32(-tt[0]), np.float32(-tt[1]), N, plM["iCol"], plM["iRow"], src=datM[l])
                    cm = self.ctfM[l]
                    if ctfs:   # CTF(ctf, .., defocusU * d, defocusV * d, ..) of the draw's defocus factor, :7183-7202
                        cm = ctf(c["pixelSize"], *self.attr_d(l, self.d[l, iD[m]]), N, plM["iCol"], plM["iRow"])
                    insertP(F[vi][kc], T[vi][kc], PF, src, cm, rotate3D(self.q[l, iR[m]]), w, plM["iColPad"], plM["iRowPad"])
            self.sig[vi], self.sigRcp[vi] = sig, rcp
        out["F_raw"], out["T_raw"] = [[x.copy() for x in h] for h in F], [[x.copy() for x in h] for h in T]
        # prepareTF (one rank per half: the all-reduces are the identity), :7268 -> src/Reconstructor.cpp:1056-1091: normalise
        # (allReduceT's tail, :2455-2476), symmetrizeT, symmetrizeF
        symr = self.rU * pf + 1
        for vi in range(2):
            for k in range(K):
                if not T[vi][k].flat[0] > 0:
                    continue
                normalise_TF(F[vi][k], T[vi][k], PF)
                if self.symR is not None:
                    T[vi][k] = symmetrize(T[vi][k], PF, self.symR, symr)
                    F[vi][k] = symmetrize(F[vi][k], PF, self.symR, symr)
        out["F_sym"], out["T_sym"] = [[x.copy() for x in h] for h in F], [[x.copy() for x in h] for h in T]
        Tn = [[t.copy() for t in h] for h in T]
        # class distribution (refreshClassDistr, :5484-5516) and balanceClass after a global search
        distr = np.bincount(self.cls, minlength=K).astype(np.float64)
        distr /= distr.sum()
        bm = self.balance_map(distr) if (glob and K > 1 and c["balanceClass"]) else np.full(K, -1, np.int64)
        out["bm"], out["distr"] = bm, distr
        rec = self._reconstruct_all(F, T, None, bm)
        out["F"], out["T"] = F, T
        if force_rounds is not None and not np.array_equal(np.asarray(force_rounds), rec["rounds"]):
            out["forced"] = self._reconstruct_all(F, Tn, np.asarray(force_rounds), bm)
        if device_FT is not None:
            # for the checker: the reconstruction stage on the DEVICE's F / T after prepareTF (every stage compared on identical
            # inputs), for the device's round counts -- and once more with 1e-6 relative noise on those inputs: how far this
            # volume's balancing loop carries the rounding level of the insertion (thin coverage makes it amplify by 1e4 and more)
            fr = None if force_rounds is None else np.asarray(force_rounds)
            rng = np.random.default_rng(20240)
            Fd = [[np.where(Tn[vi][k].flat[0] > 0, device_FT[0][vi][k], 0).astype(np.complex64) for k in range(K)] for vi in range(2)]
            Td = [[np.where(Tn[vi][k].flat[0] > 0, device_FT[1][vi][k], 0).astype(np.float32) for k in range(K)] for vi in range(2)]
            noisy = lambda X, dt: [[(x * (1 + 1e-6 * rng.standard_normal(x.shape))).astype(dt) for x in h] for h in X]
            Fn_, Tn_ = noisy(Fd, np.complex64), noisy(Td, np.float32)
            out["onDevice"] = self._reconstruct_all(Fd, Td, fr, bm)
            state = (self.fscReco.copy(), self.iterCount)

            def with_noise():   # (on demand: the checker asks only where a comparison needs the volume's conditioning)
                keep = (self.fscReco, self.iterCount)
                self.fscReco, self.iterCount = state
                try:
                    return self._reconstruct_all(Fn_, Tn_, fr, bm)
                finally:
                    self.fscReco, self.iterCount = keep
            out["onDeviceNoise"] = with_noise
        out["mapsFsc"], out["mapsMAP"], fsc_, mapsX, rounds = rec["mapsFsc"], rec["mapsMAP"], rec["fsc"], rec["maps"], rec["rounds"]
        if "avgR" in rec:
            out["avgR"] = rec["avgR"]
        out["keep"] = rec["keep"]
        for vi in range(2):
            for k in range(K):
                if not rec["keep"][vi][k]:
                    self.vols[vi][k] = set_projectee(mapsX[vi][k], pf)           # Model::refreshProj
        self.fscReco = fsc_.astype(np.float32).copy()                            # Model::resetReco, src/Model.cpp:1122
        # reCentreImg + reMaskImg, :6065-6149 -- not after a global search (:3790-3810)
        if not glob:
            for l in range(self.n):
                tr = self.topT[l].copy()
                self.offset[l] -= tr
                self.t[l] -= tr
                self.topT[l] -= tr
                self.img[l] = translate_image(self.imgOri[l], self.offset[l, 0], self.offset[l, 1])
            self.img = self._remask(self.img)
        self.iterCount += 1
        out.update(fsc=fsc_, maps=mapsX, rounds=rounds, sig=self.sig.copy(), offset=self.offset.copy(), topR=self.topR.copy(),
                   q=self.q.copy(), t=self.t.copy(), vols=[[v for v in h] for h in self.vols], img=self.img, d=self.d.copy(), topD=self.topD.copy())
        return out


class ClassifyStages:
    """The first three stages of one K-class classification iteration with a global search, chained as the reference runs them
    (MODE_3D, C1, no CTF search), stage boundaries of the native driver's global search (thunder_amd/csrc/thx_refine.hip:scan_and_select):

      scan      src/Optimiser.cpp:756-894    for every class k: slices of reference k at the nR scanned rotations (Projector::project at
                                             r = rScan), likelihood of every image at every (rotation, shift), weights and the running
                                             baseline carried from class to class (expect_global above)
      classes   :925-952                     setUC / keepHalfHeightPeak(PAR_C) / resample(k, PAR_C) / rand(cls)  (pf_class_select)
      support   :953-1079                    keepHalfHeightPeak(PAR_R), resample(mLR, PAR_R), resample(mLT, PAR_T), calVari, minimum
                                             spread of the scanning phase  (pf_scan_support), with the weights of the image's class

    Random draws are inputs: `ph` offers draw_u4 / shuffle_ranks(seed, image, call, purpose, index) (tests hand in their numpy replica
    of the device's Philox streams, tests/_philox.py); the driver's numbering: call 1 = class selection (purposes 6 / 7 / 8), call 2 =
    support points (2 / 3 rotations, 4 / 5 shifts), image = index in the rank's shard."""

    def __init__(self, N, K, vols, quat, shifts, rScan, rL, ph, pf=2):
        self.N, self.K, self.pf, self.P = N, K, pf, N * pf
        self.vols, self.quat, self.shifts, self.ph = vols, f64(quat), f64(shifts), ph
        self.plS = pixel_list(N, rScan, rL, pf)

    def scan(self, datS, ctfS, sigS):
        """datS / ctfS / sigS: image-major rows [nImg][nPxlS] on the scan's pixel list -> wC [nImg][K], wR [K][nImg][nR],
        wT [K][nImg][nT], base [nImg] (uniform priors: Particle::reset(nR, nT))"""
        N, K, P, pf, plS = self.N, self.K, self.P, self.pf, self.plS
        nImg, nR, nT = datS.shape[0], len(self.quat), len(self.shifts)
        mats = np.stack([rotate3D(q) for q in self.quat])
        traP = np.stack([translate(np.float32(s[0]), np.float32(s[1]), N, plS["iCol"], plS["iRow"]) for s in self.shifts])
        dat_pm, ctf_pm, sig_pm = (np.ascontiguousarray(a.T) for a in (datS, ctfS, sigS))
        wC, wR, wT = np.zeros((nImg, K), np.float32), np.zeros((K, nImg, nR), np.float32), np.zeros((K, nImg, nT), np.float32)
        base = np.full(nImg, np.nan, np.float32)           # "unset", :737-745
        pR, pT = np.full((nImg, nR), 1.0 / nR), np.full((nImg, nT), 1.0 / nT)
        for k in range(K):
            rotP = np.stack([project(self.vols[k], P, pf, m, plS["iCol"], plS["iRow"]) for m in mats])
            expect_global(rotP, traP, dat_pm, ctf_pm, sig_pm, K, k, pR, pT, wC, wR, wT, base)
        return wC, wR, wT, base

    def classes(self, uC, seed, peakFactorC, call=1):
        ph, (nImg, K) = self.ph, uC.shape
        return np.asarray([pf_class_select(uC[l], np.full(K, 1.0 / K), peakFactorC, ph.shuffle_ranks(seed, l, call, 6, K),
                                           ph.draw_u4(seed, l, call, 7, 0)[0] / K, min(int(ph.draw_u4(seed, l, call, 8, 0)[0] * K), K - 1))
                           for l in range(nImg)], np.int32)

    def support(self, uR, uT, cls, l, seed, peakFactorR, mLR, mLT, minK, minS, call=2):
        """support points of image l from the scan weights of its class -> (pf_scan_support's dict, the rotation shuffle's ranks)"""
        ph, nR, nT = self.ph, len(self.quat), len(self.shifts)
        rankR, rankT = ph.shuffle_ranks(seed, l, call, 2, nR), ph.shuffle_ranks(seed, l, call, 4, nT)
        ws = pf_scan_support(self.quat, self.shifts, uR[cls[l], l], uT[cls[l], l], peakFactorR, mLR, mLT, rankR,
                             ph.draw_u4(seed, l, call, 3, 0)[0] / mLR, rankT, ph.draw_u4(seed, l, call, 5, 0)[0] / mLT, minK, minS)
        return ws, rankR
