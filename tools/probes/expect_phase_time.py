"""profiling aid: E-step launch time per phase for different occupancy caps (THX_EXPECT_WG_PER_CU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
sh = RefineShard(256, 10000, dev)
for cap in ("2", "3", "0"):
    os.environ["THX_EXPECT_WG_PER_CU"] = cap
    sh.reset_reference()
    sh.iteration()
    sh.expect_ms.clear()
    sh.iteration(timed=True)
    torch.cuda.synchronize()
    print("cap", cap, ["%.0f" % a.elapsed_time(b) for a, b, _ in sh.expect_ms])
