"""SURVEY 8(d)'s cross-check of the CPU baseline: the oracle port's per-op rates on 8 threads in the build container against
SURVEY 3.5's rates of the compiled reference on the same 8 cores ("must agree within 15 %; ours may be faster; report both").
Writes profiles/cpu_port_vs_survey.json.  CPU only (numpy + the oracle); run from the repo root:

    python tools/cpu_port_vs_survey.py
"""
import json
import os
import sys
import time
import ctypes as C

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from thunder_amd import synth  # noqa: E402

THREADS = 8
SURVEY = {128: {"nPxl": 6141, "project_slices_per_s": 23.9e3, "logDataVSPrior_Gpix_per_s": 3.3, "insertP_per_s": 3.78e3},
          256: {"nPxl": 25135, "project_slices_per_s": 2.39e3, "logDataVSPrior_Gpix_per_s": 3.6, "insertP_per_s": 0.70e3}}


def pixel_list_of_size(N, want):
    for rU in range(N // 2 + 2, N // 2 - 6, -1):
        for rL in (0, 1, 2):
            pl = O.pixel_list(N, rU, rL, 2)
            if pl["nPxl"] == want:
                return pl, rU, rL
    raise SystemExit("no pixel list of %d pixels at N = %d" % (want, N))


def main():
    out = {"threads": THREADS, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip(),
           "note": "oracle port (oracle/thunder_oracle.c, gcc -O2 -ffp-contract=off) against SURVEY 3.5's compiled-reference rates, "
                   "both on 8 threads of the build container; project / insertP one call per (thread, rotation) as the reference's "
                   "OpenMP loops call them", "boxes": {}}
    rng = np.random.default_rng(1)
    for N, ref in SURVEY.items():
        P = 2 * N
        pl, rU, rL = pixel_list_of_size(N, ref["nPxl"])
        vol = O.set_projectee(synth.blob_map(N, nblob=20), 2)
        L = O.lib()
        for f in ("orc_bench_project", "orc_bench_insertP", "orc_bench_logDataVSPrior"):
            getattr(L, f).restype = C.c_double
        fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
        nrot = 512 if N == 256 else 4096
        mats = np.ascontiguousarray(np.stack([O.rotate3D(q) for q in synth.random_quats(nrot, rng)]))
        iCol, iRow, iColP, iRowP = (np.ascontiguousarray(pl[k], np.int32) for k in ("iCol", "iRow", "iColPad", "iRowPad"))
        vol = np.ascontiguousarray(vol, np.complex64)
        args_p = (vol.ctypes.data_as(fp), P, 2, mats.ctypes.data_as(dp), nrot, iCol.ctypes.data_as(ip), iRow.ctypes.data_as(ip), pl["nPxl"], THREADS)
        L.orc_bench_project(*args_p)                                   # first touch
        r_proj = nrot / L.orc_bench_project(*args_p)
        sl = O.project(vol, P, 2, mats[0], pl["iCol"], pl["iRow"])
        ctf = np.ones(pl["nPxl"], np.float32)
        sig = np.full(pl["nPxl"], -0.5, np.float32)
        nl, sink = 200000 if N == 128 else 50000, C.c_double(0)
        t_like = L.orc_bench_logDataVSPrior(sl.ctypes.data_as(fp), sl.ctypes.data_as(fp), ctf.ctypes.data_as(fp), sig.ctypes.data_as(fp), pl["nPxl"], nl,
                                            THREADS, C.byref(sink))
        r_like = nl * pl["nPxl"] / t_like / 1e9
        F = np.zeros((P, P, P // 2 + 1), np.complex64)
        T = np.zeros((P, P, P // 2 + 1), np.float32)
        nins = 256 if N == 256 else 2048
        args_i = (F.ctypes.data_as(fp), T.ctypes.data_as(fp), P, sl.ctypes.data_as(fp), ctf.ctypes.data_as(fp), mats.ctypes.data_as(dp), nins,
                  C.c_float(1.0), iColP.ctypes.data_as(ip), iRowP.ctypes.data_as(ip), pl["nPxl"], THREADS)
        L.orc_bench_insertP(*args_i)
        r_ins = nins / L.orc_bench_insertP(*args_i)
        box = {"pixel_list": {"nPxl": pl["nPxl"], "rU": rU, "rL": rL},
               "project_slices_per_s": {"port": r_proj, "survey_reference": ref["project_slices_per_s"], "port_over_reference": r_proj / ref["project_slices_per_s"]},
               "logDataVSPrior_Gpix_per_s": {"port": r_like, "survey_reference": ref["logDataVSPrior_Gpix_per_s"], "port_over_reference": r_like / ref["logDataVSPrior_Gpix_per_s"],
                                            "note": "scalar restatement of logDataVSPrior_m_huabin; the reference's figure is its AVX form (ENABLE_SIMD_256)"},
               "insertP_per_s": {"port": r_ins, "survey_reference": ref["insertP_per_s"], "port_over_reference": r_ins / ref["insertP_per_s"],
                                 "note": "the CPU baseline's insertP: 8 threads on ONE F / T with `omp atomic` float adds, as the reference's"}}
        out["boxes"][str(N)] = box
        print(N, json.dumps(box, indent=1))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "cpu_port_vs_survey.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
