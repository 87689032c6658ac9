"""Probe: do k_bin (HBM-write bound) and k_acc (LDS-atomic bound) of the brick-sorted insertion overlap when two insertion calls
run on two streams (two host threads: the call waits on its stream)?  Times two batches back to back on one stream against
the same two batches concurrently on two streams.  usage: python tools/insert_overlap_probe.py [nParticles]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd import ops
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sh = RefineShard(256, n, dev, batch=n // 2)
sh.run(2)
sh.refresh_rows(0)
wR, wT = sh.expectation(0)
rot, tran = sh.draw_reco(0, wR, wT)
lo, hi = sh.ranges[sh.halves[0]]
m = hi - lo
h = m // 2
F = [torch.zeros((sh.P, sh.P, sh.P // 2 + 1), dtype=torch.complex64, device=dev) for _ in range(2)]
T = [torch.zeros((sh.P, sh.P, sh.P // 2 + 1), dtype=torch.float32, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def ins(k, stream):
    a, b = lo + k * h, lo + (k + 1) * h
    with torch.cuda.stream(stream):
        ops.insert(F[k], T[k], sh.P, sh.datM[a:b], sh.ctfM[a:b], sh.w[a:b], rot[k * h:(k + 1) * h], tran[k * h:(k + 1) * h], sh.iColM, sh.iRowM,
                   sh.pf, sh.N, offS=sh.offset[a:b])


for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ins(0, streams[0]); ins(1, streams[0])
    torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=ins, args=(k, streams[k])) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); t_par = time.perf_counter() - t0
    print("2 x %d images: one stream %.1f ms, two streams %.1f ms (%.2f x)" % (h, t_seq * 1e3, t_par * 1e3, t_par / t_seq), flush=True)
