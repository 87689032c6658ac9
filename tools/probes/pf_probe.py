import sys, numpy as np, torch
sys.path.insert(0, '.')
from thunder_amd.refine import RefineShard
from thunder_amd import ops
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
snr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
sh = RefineShard(N, 400, dev, nPhase=1, snr=snr)
st = sh.pf_state
def err():
    d = (st['topR'].cpu().numpy() * sh.quat).sum(1)
    return float(np.median(np.degrees(2*np.arccos(np.clip(np.abs(d),0,1)))))
def cloud_err():
    d = np.abs((st['r'].cpu().numpy() * sh.quat[:, None, :]).sum(2))
    return float(np.median(np.degrees(2*np.arccos(np.clip(d,0,1)))))
print('init k med', np.median(st['k'].cpu().numpy(), 0), 's med', np.median(st['s'].cpu().numpy(), 0), 'cloud err deg', cloud_err())
lo, hi = 0, 200
for ph in range(6):
    f = sh.pfL if ph == 0 else sh.pfS
    sl = slice(lo, hi)
    sh.pf_call += 1
    ops.pf_perturb(st['r'][sl], st['t'][sl], st['wR'][sl], st['wT'][sl], st['k'][sl], st['s'][sl], f, f, sh.transS, sh.transQ, sh.pf_seed, sh.pf_call)
    print('phase', ph, 'after perturb: cloud err', cloud_err(), 'wR min/max', float(st['wR'][sl].min()), float(st['wR'][sl].max()))
    rot = ops.rotmat(st['r'][sl].reshape(-1, 4)).reshape(hi - lo, sh.mLR, 9)
    r = ops.expect_local(sh.vols[0:1], sh.P, sh.pf, sh.N, sh.iCol, sh.iRow, sh.datP[sl], sh.ctfP[sl], sh.sigRcpP[sl], rot, st['t'][sl], pR=st['wR'][sl], pT=st['wT'][sl], want_logW=True)
    uR = r.wR
    # is the most likely rotation the closest to the truth?
    d = np.abs((st['r'][sl].cpu().numpy() * sh.quat[lo:hi, None, :]).sum(2))
    ang = np.degrees(2*np.arccos(np.clip(d, 0, 1)))
    best = uR.argmax(1).cpu().numpy()
    print('   angle of best-likelihood point: med', np.median(ang[np.arange(hi-lo), best]), ' min angle in cloud med', np.median(ang.min(1)), ' n eff', float((uR.sum(1)**2/(uR**2).sum(1)).median()))
    sh.pf_call += 1
    ops.pf_update(st['r'][sl], st['t'][sl], st['wR'][sl], st['wT'][sl], r.wR, r.wT, st['k'][sl], st['s'][sl], st['topR'][sl], st['topT'][sl], sh.peakFactorR, sh.pf_seed, sh.pf_call)
    print('   after update: k med', np.median(st['k'][sl].cpu().numpy(), 0), 's med', np.median(st['s'][sl].cpu().numpy(), 0), 'cloud err', cloud_err(), 'top err', err())
