import torch, numpy as np, time, sys
sys.path.insert(0, '.')
from thunder_amd.refine import RefineShard
dev = torch.device('cuda:0')
sh = RefineShard(64, 1200, dev)
for it in range(3):
    fsc = sh.iteration(timed=True)
    torch.cuda.synchronize()
    print('it', it, 'fsc', np.round(fsc[:12], 3), 'sig[0,0,:6]', sh.sig[0,0,:6].cpu().numpy(), 'sigma2', sh.sigma2,
          'offset rms', float(sh.offset.pow(2).mean().sqrt()), 'tran rms', float(sh.tranP[-1][:,0].pow(2).mean().sqrt()))
print({k: round(sum(a.elapsed_time(b) for a, b in v)/3, 2) for k, v in sh.stage_ms.items()})
print('svd-ish check: sig last cols', sh.sig[0,0,-6:].cpu().numpy())
