import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd import capi, ops
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
sh = RefineShard(64, 700, dev, mReco=20, batch=256)
sh.refresh_rows(0)
wR, wT = sh.expectation(0)
rot, tran = sh.draw_reco(0, wR, wT)
res = {}
for plain in ("1", "0"):
    os.environ["THX_INSERT_PLAIN"] = plain
    capi.call("thx_knobs_reload")
    sh.insertion(0, rot, tran)
    F, T = sh.F[0].clone(), sh.T[0].clone()
    res[plain] = (F, T)
    print("plain", plain, "F absmax %.4g T max %.4g T[0] %.4g sumT %.6g nan %s" % (F.abs().max().item(), T.max().item(), T.flatten()[0].item(), T.sum().item(), bool(torch.isnan(F.abs()).any())))
print("eF %.3e eT %.3e" % (((res["0"][0] - res["1"][0]).abs().max() / res["1"][0].abs().max()).item(), ((res["0"][1] - res["1"][1]).abs().max() / res["1"][1].abs().max()).item()))
dT = (res["0"][1] - res["1"][1])
i = int(dT.abs().argmax()); P = sh.P; nc = P // 2 + 1
print("worst T voxel", i // (P * nc), (i // nc) % P, i % nc, res["0"][1].flatten()[i].item(), res["1"][1].flatten()[i].item())
print("offset rms", sh.offset.pow(2).mean().sqrt().item(), "datM absmax", sh.datM.abs().max().item(), "median", sh.datM.abs().median().item())
