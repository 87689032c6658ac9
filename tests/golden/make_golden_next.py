"""Generates tests/golden/oracle_next_n16.npz -- REGRESSION vectors of this repo's own oracle for the callers either side
of the E/M loop: re-mask, re-centring ramps, sigma update, defocus-search rows, image
ingestion, particle-filter statistics.  Like oracle_n16.npz they freeze the oracle's behaviour (the reference cannot be
built or run in this image).

Run from the repo root:  python tests/golden/make_golden_next.py
"""
import ctypes as C
import os
import sys

import numpy as np
import scipy.fft as sfft

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from thunder_amd import synth  # noqa: E402
from _next_util import full_images  # noqa: E402


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def compute():
    rng = np.random.default_rng(20240602)
    N, pf = 16, 2
    P = N * pf
    ref = synth.blob_map(N, seed=3, nblob=6)
    vol = O.set_projectee(ref, pf)
    out = {}
    # f2: re-mask, re-centring ramps
    rl = rng.standard_normal((2, N, N)).astype(np.float32)
    ft = sfft.rfft2(rl).astype(np.complex64)
    out["remask_in"], out["remask_out"] = ft, O.remask(ft, 6.6, 1.32, 6.0)
    out["translate_image"] = O.translate_image(ft[0], 1.25, -0.75)
    out["translate_image_r"] = O.translate_image(ft[0], 1.25, -0.75, r=5.0)
    out["translate_volume"] = O.translate_volume(vol, 9.0, 0.5, -1.5, 2.25)
    # f1: sigma update
    projR, rSig = N // 2 - 2, N // 2 - 1
    im = full_images(O, vol, N, 4, rng, projR)
    spec = np.stack([O.sigma_image(vol, P, pf, N, projR, rSig, im["rot"][l], im["tran"][l], im["offset"][l], im["pixelSize"],
                                   im["attr"][l], im["img"][l], im["imgOri"][l]) for l in range(4)])
    gid = np.array([1, 2, 2, 1], np.int32)
    acc = O.sigma_accum(spec, gid, 2, True)
    sig, rcp = O.sigma_final(*acc, 9.0, N, im["pixelSize"], True)
    out.update(sig_img=im["img"], sig_imgOri=im["imgOri"], sig_rot=im["rot"], sig_tran=im["tran"], sig_offset=im["offset"],
               sig_attr=im["attr"], sig_spec=spec, sig_gid=gid, sigM=acc[0], sigN=acc[1], svd=acc[2], sig=sig, sigRcp=rcp)
    # a2: defocus-search rows, whole-image CTF
    pl = O.pixel_list(N, N // 2 - 2, 2, pf)
    attr = synth.ctf_params(2, rng)
    fq, de, k1, k2 = O.expect_precal(attr, N, 1.32, pl["iCol"], pl["iRow"])
    dpara = np.array([0.97, 1.0, 1.04])
    out.update(pre_attr=attr, pre_freq=fq, pre_def=de, pre_k1=k1, pre_k2=k2, pre_d=dpara,
               pre_rows=O.ctf_dsearch(fq, de[0], k1[0], k2[0], attr[0, 6], attr[0, 5], dpara), ctf_image=O.ctf_image(N, 1.32, attr[1]))
    # f4: ingestion
    raw = (3.0 + rng.standard_normal((5, N, N))).astype(np.float32) * 11.0
    iF, oF, st = O.init_images(raw, 5.5)
    out.update(ing_raw=raw, ing_img=iF, ing_ori=oF, ing_stats=np.array([st[k] for k in ("mean", "stdN", "stdD", "stdS", "stdStdN")]))
    # f4: particle-filter statistics
    q = synth.perturb_quats(synth.random_quats(1, rng), 40, 0.04, rng)[0]
    A = np.zeros(16)
    O.lib().orc_infer_acg.restype = C.c_int
    rounds = O.lib().orc_infer_acg(_dp(A), _dp(np.ascontiguousarray(q)), 40)
    k, mean, qq = np.zeros(3), np.zeros(4), np.ascontiguousarray(q.copy())
    O.lib().orc_cal_vari_R(_dp(k), _dp(mean), _dp(qq), 40)
    wb = np.zeros(40)
    O.lib().orc_balance_weight_R(_dp(wb), _dp(np.ascontiguousarray(q)), 40)
    t = rng.normal(0, 1.0, size=(9, 2))
    s01, wt = np.zeros(2), np.zeros(9)
    O.lib().orc_cal_vari_T(_dp(s01), _dp(np.ascontiguousarray(t)), 9)
    O.lib().orc_balance_weight_T(_dp(wt), _dp(np.ascontiguousarray(t)), 9)
    u = rng.uniform(0, 1, 40) ** 3
    idx, wo = np.zeros(40, np.int32), np.zeros(40)
    O.lib().orc_resample(idx.ctypes.data_as(C.POINTER(C.c_int)), _dp(wo), _dp(wb), _dp(u), 40, 40, C.c_double(0.3 / 40))
    out.update(pf_q=q, pf_A=A, pf_rounds=np.int32(rounds), pf_k=k, pf_mean=mean, pf_wbal=wb, pf_t=t, pf_s01=s01, pf_wt=wt,
               pf_u=u, pf_idx=idx, pf_wres=wo)
    return out


def main():
    out = compute()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_next_n16.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
