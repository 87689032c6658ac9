"""Generates tests/golden/classify_n16.npz -- REGRESSION vectors of the first three stages of a K-class classification iteration
(scan weights, class of every image, support points of the local search) produced by this repo's own oracle
(oracle/thunder_oracle.c through tests/_classify_util.py), NOT by the reference.  The CPU suite re-runs the
oracle against them (tests/test_pf_cpu.py); on the GPU the same stages of the native driver (thx_refine_iterate with THX_SEARCH_GLOBAL,
thunder_amd/csrc/thx_refine.hip:scan_and_select) are held against the LIVE oracle chain (tests/test_iteration_gpu.py::
test_classification_matches_oracle_chain), not against this file.

Run from the repo root:  python tests/golden/make_golden_classify.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CFG = dict(N=16, K=2, nImg=12, nR=40, nT=3, rScan=5, rL=1, mLR=8, mLT=3, seed=31337, peakFactorR=1e-3, peakFactorC=1.0 - 1e-2,
           minK=(40 ** (-1.0 / 3) / 0.5) ** 2, minS=0.3, case_seed=77)


def compute(O):
    import _classify_util as U
    c = CFG
    cs = U.make_case(O, c["N"], c["K"], c["nImg"], c["nR"], c["nT"], c["case_seed"], noise=0.3)
    st = U.stages(O, c["N"], c["K"], cs["vols"], cs["quat"], cs["shifts"], c["rScan"], c["rL"])
    s2m = U.sub_rows(cs["plM"], st.plS)
    wC, wR, wT, base = st.scan(cs["datM"][:, s2m], cs["ctfM"][:, s2m], cs["sigM"][:, s2m])
    cls = st.classes(wC, c["seed"], c["peakFactorC"])
    r0 = np.zeros((c["nImg"], c["mLR"], 4)); t0 = np.zeros((c["nImg"], c["mLT"], 2))
    srcR = np.zeros((c["nImg"], c["mLR"]), np.int64); k123 = np.zeros((c["nImg"], 3)); s01 = np.zeros((c["nImg"], 2))
    for l in range(c["nImg"]):
        ws, _ = st.support(wR, wT, cls, l, c["seed"], c["peakFactorR"], c["mLR"], c["mLT"], c["minK"], c["minS"])
        r0[l], t0[l], srcR[l], k123[l], s01[l] = ws["q"], ws["t"], ws["srcR"], ws["k"], ws["s"]
    return dict(refs=cs["refs"], quat=cs["quat"], shifts=cs["shifts"], datM=cs["datM"], ctfM=cs["ctfM"], sigM=cs["sigM"], cls_true=cs["cls_true"],
                wC=wC, wR=wR, wT=wT, base=base, cls=cls, r0=r0, t0=t0, srcR=srcR, k123=k123, s01=s01)


def main():
    from oracle import oracle as O
    out = compute(O)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "classify_n16.npz"), **out)
    print({k: v.shape for k, v in out.items()}, "classes recovered", float((out["cls"] == out["cls_true"]).mean()))


if __name__ == "__main__":
    main()
