"""where do the window kernel and the oracle differ at N = 256?  (debugging aid: the inputs of
tests/test_fullsize_gpu.py::_insert_vs_oracle with individual ingredients switched off)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from oracle import oracle as O
from thunder_amd import capi, ops, synth
from thunder_amd.refine import pixel_list
from test_fullsize_gpu import _filter_draws, _oracle_insert, _noisy_rows, T
dev = torch.device("cuda:0")
N = int(os.environ.get("DBG_N", "256")); P = 2 * N
pl = pixel_list(N, N // 2 - 2, 0)
plan = ops.RecoPlan(N, N, 2)
vol_h = plan.set_projectee(T(synth.blob_map(N, nblob=8), dev)).cpu().numpy()
plan.close()


def run(name, dat, ctf, quat, tran, offS, w, detail=False):
    nImg, mReco = quat.shape[0], quat.shape[1]
    Fw, Tw = _oracle_insert(O, P, N, pl, dat, ctf, quat, tran, offS, w)
    rot = ops.rotmat(T(quat.reshape(-1, 4), dev)).reshape(nImg, mReco, 9)
    for plain in ("0", "1"):
        os.environ["THX_INSERT_PLAIN"] = plain
        capi.call("thx_knobs_reload")
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        ops.insert(F, Tt, P, T(dat, dev), T(ctf, dev), T(w, dev), rot, T(tran, dev), T(pl["iCol"], dev), T(pl["iRow"], dev), 2, N,
                   offS=T(offS, dev))
        Fg, Tg = F.cpu().numpy(), Tt.cpu().numpy()
        dT = Tg - Tw
        eF, eT = np.abs(Fg - Fw).max() / np.abs(Fw).max(), np.abs(dT).max() / np.abs(Tw).max()
        print("%-34s plain=%s  eF %.2e  eT %.2e   sumT got %.6f want %.6f  max|T| %.4f" % (name, plain, eF, eT, Tg.sum(dtype=np.float64),
                                                                                     Tw.sum(dtype=np.float64), np.abs(Tw).max()), flush=True)
        if plain == "0" and eT > 1e-5 and detail:
            idx = np.argsort(-np.abs(dT).ravel())[:10]
            for e in idx:
                k, j, i = np.unravel_index(e, dT.shape)
                ks, js = (k if k < P // 2 else k - P), (j if j < P // 2 else j - P)
                print("   voxel k %4d j %4d i %4d  r %.1f  T got %.6f want %.6f  d %.3e" % (ks, js, i, np.sqrt(ks * ks + js * js + i * i),
                                                                                     Tg[k, j, i], Tw[k, j, i], dT[k, j, i]))
            print("   #voxels |dT| > 1e-5 max: %d ; sum dT %.3e ; sum |dT| %.3e" % (int((np.abs(dT) > 1e-5 * np.abs(Tw).max()).sum()), dT.sum(dtype=np.float64), np.abs(dT).sum(dtype=np.float64)))


def case(seed, nImg, mReco, spread):
    rng = np.random.default_rng(seed)
    quat0 = synth.random_quats(nImg, rng)
    shift0 = rng.normal(0, 2.0, size=(nImg, 2))
    attr = synth.ctf_params(nImg, rng)
    dat, ctf = _noisy_rows(O, vol_h, P, N, pl, quat0, shift0, attr, rng)
    quat, tran = _filter_draws(rng, synth, quat0, shift0, nImg, 125, 9, mReco, spread)
    offS = rng.normal(0, 0.8, size=(nImg, 2))
    w = (rng.uniform(0.5, 1.0, size=nImg) / mReco).astype(np.float32)
    return dat, ctf, quat, tran, offS, w


dat, ctf, quat, tran, offS, w = case(2560, 3, 100, 0.0175)
run("test case as is (3 img)", dat, ctf, quat, tran, offS, w, detail=True)
for l in range(3):
    run("  image %d alone" % l, dat[l:l + 1], ctf[l:l + 1], quat[l:l + 1], tran[l:l + 1], offS[l:l + 1], w[l:l + 1], detail=(l == 0))
l = 0
one = np.ones_like(dat[l:l + 1]); onef = np.ones_like(ctf[l:l + 1])
run("  img0: dat=1", one, ctf[l:l + 1], quat[l:l + 1], tran[l:l + 1], offS[l:l + 1], w[l:l + 1])
run("  img0: ctf=1", dat[l:l + 1], onef, quat[l:l + 1], tran[l:l + 1], offS[l:l + 1], w[l:l + 1])
run("  img0: w=0.01", dat[l:l + 1], ctf[l:l + 1], quat[l:l + 1], tran[l:l + 1], offS[l:l + 1], np.full(1, 0.01, np.float32))
run("  img0: tran=offS=0", dat[l:l + 1], ctf[l:l + 1], quat[l:l + 1], 0 * tran[l:l + 1], 0 * offS[l:l + 1], w[l:l + 1])
run("  img0: first 10 draws", dat[l:l + 1], ctf[l:l + 1], quat[l:l + 1, :10], tran[l:l + 1, :10], offS[l:l + 1], w[l:l + 1], detail=True)
