"""One rank of an N-rank run of the native iteration driver on ONE GPU (tests/test_multirank_gpu.py starts N of these).

Every rank generates the SAME one-rank job (RefineShard(world=1), seeded), keeps its own part (refine.take_shard: the j-th contiguous
part of half rank mod 2, with the image numbering of the one-rank layout), builds the native communicators over the TEST-ONLY
shared-memory transport (THX_COMM_TRANSPORT=shm, thx_comm.hip: RCCL refuses two ranks on one device) -- the 128-byte ids travel
through files in --dir, as the reference broadcasts them over MPI (gpu/src/cuthunder.cu:4192-4206) -- and runs the UNCHANGED
thx_refine_iterate: the world / hemi branches of thx_refine.hip (norm gather, sigma tables, class histogram, the reduce of the 64-bit
accumulators towards the reconstructing rank, the map broadcasts).  What it leaves in <dir>/rank<r>.npz is compared with the
one-rank run (the same script with --world 1).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # one class, local search, normCorrection on (the configuration bench.py times), 3 iterations
    "k1": dict(N=32, n=240, K=1, sym=None, scan=None, iters=("local", "local", "local"), norm=True, kw=dict(mLR=40, mLT=5, mReco=16)),
    # K = 4 classes: global search with balanceClass, then a local search in the assigned classes.  Class 3 has no particles
    # (K_used = 3), so that determineBalanceClass has to hand it another class's reference -- the same one on every rank
    "k4": dict(N=32, n=384, K=4, sym=None, scan=dict(nR=120, nT=4, rScan=8), iters=("global", "local"), norm=False,
               kw=dict(mLR=40, mLT=4, mReco=16, K_used=3)),
    # point group C4, one class
    "c4": dict(N=32, n=200, K=1, sym="C4", scan=None, iters=("local", "local"), norm=False, kw=dict(mLR=40, mLT=5, mReco=16)),
    # round 6: frequency cut-offs handed in before every iteration (thx_refine_set_cutoff on EVERY rank): the E-step list, the M-step list
    # and the reconstructors' grid (44^3, 56^3, 64^3) change from iteration to iteration; the half-set reduce runs on the resized grid
    "k1cut": dict(N=32, n=240, K=1, sym=None, scan=None, iters=("local", "local", "local"), norm=True, kw=dict(mLR=40, mLT=5, mReco=16),
                  cut=((8, 9), (10, 12), (14, 14))),
}


def file_share(d, rank):
    ctr = [0]

    def share_from(root, uid):
        ctr[0] += 1
        path = os.path.join(d, "id_%d.bin" % ctr[0])
        if rank == root:
            with open(path + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(path + ".tmp", path)
            return uid
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 300:
                raise RuntimeError("rank %d: no id from rank %d" % (rank, root))
            time.sleep(0.01)
        return open(path, "rb").read()
    return share_from


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--case", default="k1")
    ap.add_argument("--gpu-per-rank", action="store_true", help="rank r on device r and RCCL as the transport (needs `world` GPUs)")
    a = ap.parse_args()
    import torch
    from thunder_amd import capi
    from thunder_amd.native import NativeRefine, make_comms
    from thunder_amd.refine import RefineShard, take_shard
    capi.load()
    dev = torch.device("cuda", a.rank if a.gpu_per_rank else 0)
    torch.cuda.set_device(dev)
    c = CASES[a.case]
    N, n, K = c["N"], c["n"], c["K"]
    full = RefineShard(N, n, dev, snr=0.2, K=K, sym=c["sym"], scan=c["scan"], search=c["iters"][0], allocate=False, nblob=16, batch=64, **c["kw"])
    full.balanceClass = 1 if K > 1 else 0
    if c["kw"].get("K_used") is not None:
        # the class without particles must not attract any in the scan either (a handful of misassigned images would keep it above
        # CLASS_BALANCE_FACTOR / K): its initial reference is far too strong to explain any image
        for k in range(c["kw"]["K_used"], K):
            full.refs[k] *= 50.0
    full.release_generation_state()
    sh = take_shard(full, a.rank, a.world) if a.world > 1 else full
    hemi = wcomm = None
    if a.world > 1:
        hemi, wcomm = make_comms(a.rank, a.world, file_share(a.dir, a.rank))
        tp = capi.load().thx_comm_transport(wcomm.handle)
        assert tp == (b"rccl" if a.gpu_per_rank else b"shm"), tp
    nat = NativeRefine(sh, hemi, wcomm, norm_correction=c["norm"])
    cap = nat.capture(maps=True)
    nat.reset()
    out = {}
    nV = 2 if a.world == 1 else 1
    for it, search in enumerate(c["iters"]):
        nat.set_search(search)
        if c.get("cut"):
            nat.set_cutoff(*c["cut"][it])
        fsc = nat.iterate()
        torch.cuda.synchronize()
        st = nat.stats()
        v = nat.view()
        m = sh.nImg
        out["fsc%d" % it] = np.atleast_2d(fsc)
        # (the accumulators are volumes of the reconstructors' CURRENT grid, contiguous from the start of the capture buffers)
        PF = v.fdim
        volF = PF * PF * (PF // 2 + 1)
        out["Fraw%d" % it] = cap["Fraw"].cpu().numpy().reshape(-1)[:nV * K * volF].reshape(nV, K, PF, PF, PF // 2 + 1).copy()
        out["Traw%d" % it] = cap["Traw"].cpu().numpy().reshape(-1)[:nV * K * volF].reshape(nV, K, PF, PF, PF // 2 + 1).copy()
        out["mapsFsc%d" % it] = cap["mapsFsc"].cpu().numpy()
        out["maps%d" % it] = np.stack([[nat.map(h, k).cpu().numpy() for k in range(K)] for h in (0, 1)])
        out["sig%d" % it] = nat.fetch(v.sig, np.float32, (nV, sh.nGroup, N // 2 - 1))
        out["cls%d" % it] = nat.fetch(v.cls, np.int32, (m,))
        out["topR%d" % it] = nat.fetch(v.topR, np.float64, (m, 4))
        out["topT%d" % it] = nat.fetch(v.topT, np.float64, (m, 2))
        out["offset%d" % it] = nat.fetch(v.offset, np.float64, (m, 2))
        out["r%d" % it] = nat.fetch(v.r, np.float64, (m, sh.mLR, 4))
        out["rounds%d" % it] = nat.rounds()
        out["balanced%d" % it] = np.asarray(list(st.balanced), np.int32)
        out["classCount%d" % it] = np.asarray(list(st.classCount), np.int32)
        out["normMedian%d" % it] = np.float32(st.normMedian)
        if c["norm"] and st.normMedian > 0:
            out["norm%d" % it] = nat.fetch(v.norm, np.float32, (m,))
    out["lo_hi"] = np.asarray([getattr(sh, "img_base", 0), getattr(sh, "img_base", 0) + sh.nImg])
    print("rank %d of %d, case %s: images per class after iteration 1 %s, balanced %s" % (a.rank, a.world, a.case, out["classCount0"][:K], out["balanced0"][:K]))
    np.savez(os.path.join(a.dir, "rank%d.npz" % a.rank), **out)
    nat.close()
    if hemi is not None:
        hemi.close()
    if wcomm is not None:
        wcomm.close()


if __name__ == "__main__":
    main()
