"""A/B timing of the two insertion forms on the filter's steady-state draws: THX_INSERT=win (per-image window kernel) against the
default brick-sorted form.  usage: python tools/insert_ab.py [nParticles] [box]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd import capi
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
box = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sh = RefineShard(box, n, dev, batch=2048)
sh.run(int(os.environ.get("ITERS", "2")))   # the filter's steady-state clouds
sh.refresh_rows(0)
wR, wT = sh.expectation(0)
rot, tran = sh.draw_reco(0, wR, wT)
m = rot.shape[0]
r = rot.reshape(m, sh.mReco, 9)
distinct = np.mean([len(np.unique(r[i].cpu().numpy(), axis=0)) for i in range(0, m, max(1, m // 64))])
print("avg distinct rotations per image among %d draws: %.1f" % (sh.mReco, distinct))
for rep in range(3):
    sh.insertion(0, rot, tran); torch.cuda.synchronize()
    t0 = time.perf_counter(); sh.insertion(0, rot, tran); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("THX_INSERT=%s  %.1f ms for %d images = %.2f us/particle" % (os.environ.get("THX_INSERT", "sort"), dt * 1e3, m, dt / m * 1e6), flush=True)
