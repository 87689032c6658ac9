"""Generates tests/golden/iteration_n32.npz -- the oracle's chain of two EM iterations (oracle.Iteration) on seeded synthetic
particles at N = 32: REGRESSION vectors of this repo's own oracle (the reference cannot be built or run in this image,
DESIGN.md section 3).  The inputs are not stored: tests/_iter_util.make_inputs regenerates them from the seed (numpy's PCG64 and
the oracle's own arithmetic) and the fixture carries their SHA-256, so a drift of the inputs shows up as such.

Run from the repo root:  python tests/golden/make_golden_iteration.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
import _iter_util as U  # noqa: E402

PARAMS = dict(N=32, n=120, seed=7321, mReco=20, batch=32, snr=2.0)


def input_hash(inp):
    h = hashlib.sha256()
    for k in ("imgOri", "attr", "gid", "quat0", "tran0", "ref"):
        h.update(np.ascontiguousarray(inp[k]).tobytes())
    return h.hexdigest()


def compute():
    p = PARAMS
    inp = U.make_inputs(O, p["N"], p["n"], seed=p["seed"], mReco=p["mReco"], batch=p["batch"], snr=p["snr"])
    it = U.oracle_chain(O, inp)
    out = {"input_sha256": np.frombuffer(input_hash(inp).encode(), np.uint8)}
    for i in (1, 2):
        r = it.iterate()
        out.update({"it%d_uR" % i: r["uR"], "it%d_uT" % i: r["uT"], "it%d_srcR" % i: r["srcR"].astype(np.int16),
                    "it%d_srcT" % i: r["srcT"].astype(np.int16), "it%d_k" % i: r["k"], "it%d_s" % i: r["s"],
                    "it%d_sig" % i: r["sig"], "it%d_fsc" % i: r["fsc"][0], "it%d_rounds" % i: np.asarray(r["rounds"], np.int32).reshape(-1),
                    "it%d_maps" % i: np.stack([h[0] for h in r["maps"]]), "it%d_mapsFsc" % i: np.stack([h[0] for h in r["mapsFsc"]]),
                    "it%d_topR" % i: r["topR"], "it%d_offset" % i: r["offset"], "it%d_avgR" % i: np.asarray([r["avgR"]], np.int32),
                    "it%d_Tsum" % i: np.asarray([float(h[0].sum(dtype=np.float64)) for h in r["T_raw"]]),
                    "it%d_Fabs" % i: np.asarray([float(np.abs(h[0]).sum(dtype=np.float64)) for h in r["F_raw"]])})
    return inp, out


if __name__ == "__main__":
    inp, out = compute()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "iteration_n32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
