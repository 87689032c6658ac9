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
if sh.use_pf:
    st = sh.pf_state
    print('pf k', st['k'][:3].cpu().numpy(), 's', st['s'][:3].cpu().numpy())
    import thunder_amd.synth as sy
    d = (st['topR'].cpu().numpy() * sh.quat).sum(1)
    print('angular error of top pose (deg) median', np.degrees(2*np.arccos(np.clip(np.abs(d),0,1))).__array__().__class__ and float(np.median(np.degrees(2*np.arccos(np.clip(np.abs(d),0,1))))))
    print('shift error median', float(np.median(np.linalg.norm(st['topT'].cpu().numpy() - (sh.shift + sh.offset.cpu().numpy()), axis=1))))
