"""profiling aid: time the tiled insert kernel under the THX_INSERT_DEBUG ablations"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd import capi
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sh = RefineShard(256, n, dev, batch=2048)
wR, wT = sh.expectation(0)
rot, tran = sh.draw_reco(0, wR, wT)
n = rot.shape[0]
# distinct rotations per image among the draws
r = rot.reshape(n, sh.mReco, 9)
distinct = np.mean([len(np.unique(r[i].cpu().numpy(), axis=0)) for i in range(0, n, max(1, n // 64))])
print("avg distinct rotations per image among %d draws: %.1f" % (sh.mReco, distinct))
for dbg in [int(x) for x in os.environ.get("DBGS", "8,0,1,2,3,4").split(",")]:
    os.environ["THX_INSERT_DEBUG"] = str(dbg)
    sh.insertion(0, rot, tran); torch.cuda.synchronize()
    t0 = time.perf_counter(); sh.insertion(0, rot, tran); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    msg = "debug=%d  %.1f ms for %d images = %.1f us/particle" % (dbg, dt * 1e3, n, dt / n * 1e6)
    if dbg == 8:
        out = (ctypes.c_ulonglong * 2)()
        capi.load().thx_debug_insert_stats(out)
        msg += "  in-brick adds %d fallback adds %d (%.2f%%)" % (out[0], out[1], 100.0 * out[1] / max(1, out[0] + out[1]))
    print(msg)
os.environ["THX_INSERT_DEBUG"] = "0"
os.environ["THX_INSERT_PLAIN"] = "1"  # set before the library is loaded (read once)
