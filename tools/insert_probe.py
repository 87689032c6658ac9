"""profiling aid: time the window insertion kernel under the THX_INSERT_DEBUG ablations (needs a library built with
THX_EXTRA_FLAGS=-DTHX_PROFILING): 0 = production, 1 = no LDS adds, 2 = no flush, 3 = neither, 4 = no group walk at all"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd import capi
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sh = RefineShard(256, n, dev, batch=2048)
sh.run(1)                                  # the filter's steady-state clouds
sh.refresh_rows(0)
wR, wT = sh.expectation(0)
rot, tran = sh.draw_reco(0, wR, wT)
m = rot.shape[0]
r = rot.reshape(m, sh.mReco, 9)
distinct = np.mean([len(np.unique(r[i].cpu().numpy(), axis=0)) for i in range(0, m, max(1, m // 64))])
print("avg distinct rotations per image among %d draws: %.1f" % (sh.mReco, distinct))
for dbg in [int(x) for x in os.environ.get("DBGS", "0,1,2,3,4").split(",")]:
    os.environ["THX_INSERT_DEBUG"] = str(dbg)
    capi.call("thx_knobs_reload")
    sh.insertion(0, rot, tran); torch.cuda.synchronize()
    t0 = time.perf_counter(); sh.insertion(0, rot, tran); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("debug=%d  %.1f ms for %d images = %.1f us/particle" % (dbg, dt * 1e3, m, dt / m * 1e6), flush=True)
