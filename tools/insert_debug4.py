import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd import capi, ops
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
for plain in ("1", "0"):
    os.environ["THX_INSERT_PLAIN"] = plain
    capi.call("thx_knobs_reload")
    sh = RefineShard(64, 700, dev, mReco=20, batch=256)
    fsc = sh.run(1)
    print("plain", plain, "fsc", np.round(fsc[:12], 3), "rounds", sh.reco_rounds, flush=True)
    for vi in (0, 1):
        F, T = sh.F[vi], sh.T[vi]
        print("   half", vi, "F absmax %.4g T max %.4g min %.4g nanF %s map absmax %.4g" % (F.abs().max().item(), T.max().item(), T.min().item(), bool(torch.isnan(F.abs()).any()), sh.last["maps"][vi].abs().max().item()))
