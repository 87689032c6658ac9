"""timing aid: one defocus-search phase (mLR 125 x mLT 9 x mLD 9) over particle-filter-like clouds at the 256^3 box,
fused kernel vs one sweep per defocus factor (THX_EXPECT_ND=sweep)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd import ops
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sh = RefineShard(256, n, dev, batch=n)
sh.expectation(0)                       # brings the filter clouds to their steady state
lo, hi = sh.ranges[0]
st = sh.pf_state
rot = ops.rotmat(st["r"][lo:hi].reshape(-1, 4)).reshape(hi - lo, sh.mLR, 9)
tran = st["t"][lo:hi].contiguous()
nD = 9
ctfD = (sh.ctfP[lo:hi, None, :] * torch.linspace(0.96, 1.04, nD, device=dev)[None, :, None]).contiguous()
cells = sh.cells[0:1]
for mode in ("sweep", "fused"):
    if mode == "sweep":
        os.environ["THX_EXPECT_ND"] = "sweep"
    else:
        os.environ.pop("THX_EXPECT_ND", None)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ops.expect_local(cells, sh.P, sh.pf, sh.N, sh.iCol, sh.iRow, sh.datP[lo:hi], ctfD, sh.sigRcpP[lo:hi], rot, tran, nD=nD,
                             packed=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d images, nD = %d, %s: %.1f ms (%.1f us per image)" % (hi - lo, nD, mode, dt * 1e3, dt / (hi - lo) * 1e6))
os.environ.pop("THX_EXPECT_ND", None)
for nd in (1, 2, 4):
    c = ctfD[:, :nd].contiguous()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ops.expect_local(cells, sh.P, sh.pf, sh.N, sh.iCol, sh.iRow, sh.datP[lo:hi], c, sh.sigRcpP[lo:hi], rot, tran, nD=nd,
                             packed=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d images, nD = %d (default path): %.1f ms (%.1f us per image)" % (hi - lo, nd, dt * 1e3, dt / (hi - lo) * 1e6))
