"""profiling aid: E-step kernel time per phase, normal vs cache-resident sampling (THX_EXPECT_DEBUG=1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sh = RefineShard(256, n, dev, batch=2048)
for dbg in ("0", "1"):
    os.environ["THX_EXPECT_DEBUG"] = dbg
    sh.expectation(0); sh.expectation(1); torch.cuda.synchronize()
    sh.expect_ms.clear()
    sh.expectation(0, timed=True); sh.expectation(1, timed=True); torch.cuda.synchronize()
    print("THX_EXPECT_DEBUG=%s per-phase ms:" % dbg, ["%.1f" % a.elapsed_time(b) for a, b, _ in sh.expect_ms])
