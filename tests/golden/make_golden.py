"""Generates tests/golden/oracle_n16.npz -- REGRESSION vectors produced by this repo's own oracle
(oracle/thunder_oracle.c), NOT by the reference: the reference cannot be built or run in this image (DESIGN.md
section 3), and its own tests hold no vectors for this path.  They freeze the oracle's behaviour so that an
accidental change to it (or to the HIP path, which is compared against the same file on the GPU) is caught.
The only reference-derived known answers available are the pixel-list sizes recorded in SURVEY.md 8 / BASELINE.md 2
(measured with the compiled reference during the survey); they are asserted in tests/test_oracle_cpu.py.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from thunder_amd import synth  # noqa: E402
from _util import edge_rotations  # noqa: E402


def main():
    rng = np.random.default_rng(20240601)
    N, pf = 16, 2
    P = N * pf
    ref = synth.blob_map(N, seed=3, nblob=6)
    vol = O.set_projectee(ref, pf)
    pl = O.pixel_list(N, N // 2 - 2, 0, pf)
    mats = edge_rotations(rng, n_random=4)
    slices = np.stack([O.project(vol, P, pf, m, pl["iCol"], pl["iRow"]) for m in mats])
    attr = synth.ctf_params(2, rng)
    ctf = np.stack([O.ctf(1.32, *a, N, pl["iCol"], pl["iRow"]) for a in attr])
    shifts = np.array([[0.7, -1.3], [2.25, 0.5]])
    ramps = np.stack([O.translate(np.float32(s[0]), np.float32(s[1]), N, pl["iCol"], pl["iRow"]) for s in shifts])
    quat = synth.random_quats(1, rng)
    qs = synth.perturb_quats(quat, 6, 0.05, rng)[0]
    rot = np.stack([O.rotate3D(q) for q in qs])
    dat = (slices[6] * ctf[0] * ramps[0]).astype(np.complex64)
    dat = (dat + 0.05 * (rng.normal(size=dat.shape) + 1j * rng.normal(size=dat.shape))).astype(np.complex64)
    sig = np.full(pl["nPxl"], -0.5 / 0.05 ** 2 / 2, np.float32)
    tran = np.array([[0.7, -1.3], [0.5, -1.0], [1.0, -1.5]])
    pR = rng.uniform(0.5, 1.5, 6)
    pT = rng.uniform(0.5, 1.5, 3)
    ex = O.expect_local(vol, P, pf, N, pl["iCol"], pl["iRow"], dat, ctf[0], sig, rot, tran, pR=pR, pT=pT)
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    T = np.zeros((P, P, P // 2 + 1), np.float32)
    for k in range(len(mats)):
        O.insertP(F, T, P, slices[k], ctf[k % 2], mats[k], np.float32(0.25), pl["iColPad"], pl["iRowPad"])
    Fn, Tn = F.copy(), T.copy()
    O.normalise_TF(Fn, Tn, P)
    sym = synth.cn_symmetry(3)
    Fs = O.symmetrize(Fn, P, sym, (N // 2 - 2) * pf + 1)
    Ts = O.symmetrize(Tn, P, sym, (N // 2 - 2) * pf + 1)
    fscv = np.linspace(1, 0.2, N // 2).astype(np.float32)
    rec = O.reconstruct(Fs, Ts, P, N, pf, N // 2 - 2, FSC=fscv, joinHalf=True, MAP=True, gridCorr=True)
    tab = O.kernelRL_table()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_n16.npz")
    np.savez_compressed(out, N=N, pf=pf, ref=ref, vol=vol, iCol=pl["iCol"], iRow=pl["iRow"], iPxl=pl["iPxl"], iSig=pl["iSig"],
                        mats=mats, slices=slices, attr=attr, ctf=ctf, shifts=shifts, ramps=ramps, rot=rot, dat=dat,
                        sig=sig, tran=tran, pR=pR, pT=pT, wR=ex["wR"], wT=ex["wT"], wC=ex["wC"], wD=ex["wD"],
                        logW=ex["logW"], baseLine=np.float32(ex["baseLine"]), F=F, T=T, Fs=Fs, Ts=Ts, sym=sym,
                        fscv=fscv, rec=rec, tab_head=tab[:64], tab_tail=tab[-64:], tab_sum=np.float64(tab.sum(dtype=np.float64)),
                        nf=np.float32(O.lib().orc_MKB_RL(0.0, 1.9, 15.0)))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
