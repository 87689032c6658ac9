"""workload for tools/pmc_traffic.sh: one E-step phase + one insertion over 2048 synthetic 256^3 particles, preceded by
a calibration copy of known size (1 GiB read + 1 GiB written by a float4 elementwise kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(os.environ.get("THX_PROBE_PARTICLES", "2048"))
a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
torch.cuda.synchronize()
b.copy_(a)            # calibration dispatch: 1 GiB in, 1 GiB out
torch.cuda.synchronize()
sh = RefineShard(256, n, dev, batch=2048, nPhase=1)
for vi in (0, 1):
    wR, wT = sh.expectation(vi)
    rot, tran = sh.draw_reco(vi, wR, wT)
    sh.insertion(vi, rot, tran)
torch.cuda.synchronize()
print("probe done", n)
