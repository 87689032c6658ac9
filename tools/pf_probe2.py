import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from thunder_amd.refine import RefineShard
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
for use_pf in (False, True):
    sh = RefineShard(N, n, dev, particle_filter=use_pf)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sh.refresh_rows(0)
        wR, wT = sh.expectation(0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        rot, tran = sh.draw_reco(0, wR, wT)
        sh.insertion(0, rot, tran)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        lo, hi = sh.ranges[0]
        print('pf' if use_pf else 'fixed', 'iter', it, 'E-step %.1f ms  insertion %.1f ms for %d images' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, hi - lo))
        # spread of the draws about their first draw, in voxels at the edge of the sphere
        R = rot.reshape(hi - lo, sh.mReco, 3, 3)   # column-major 3x3: element [c][r]
        z = R[:, :, 2, :]                           # third column = projection direction
        ang = torch.rad2deg(torch.arccos((z * z[:, :1]).sum(-1).clamp(-1, 1)))
        q = torch.quantile(ang.flatten(), torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev, dtype=ang.dtype))
        print('   view-direction angle of the draws to draw 0 (deg): p50 %.2f p90 %.2f p99 %.2f p99.9 %.2f max %.2f' % (*q.tolist(), float(ang.max())))
        if use_pf:
            st = sh.pf_state
            print('   k med', np.median(st['k'][lo:hi].cpu().numpy(), 0), 'unique rotations per image med', float(torch.tensor([len(torch.unique(st['r'][i], dim=0)) for i in range(lo, lo + 50)], dtype=torch.float32).median()))
    del sh
    torch.cuda.empty_cache()
