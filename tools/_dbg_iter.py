import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import oracle as O
import _iter_util as U, _philox as PH
from test_iteration_gpu import native_from_inputs
from thunder_amd import ops
dev=torch.device("cuda:0")
N=int(sys.argv[1]); n=int(sys.argv[2]); snr=float(sys.argv[3])
inp = U.make_inputs(O, N, n, seed=100+N, mReco=20, batch=int(sys.argv[4]), snr=snr)
c=inp["cfg"]; P=2*N; rU=N//2-2
it = U.oracle_chain(O, inp)
nat, shim = native_from_inputs(inp, dev)
cap = nat.capture()
nat.reset(); torch.cuda.synchronize()
nat.iterate(); torch.cuda.synchronize()
capn = {k: v.cpu().numpy() for k, v in cap.items() if v is not None}
fol = U.Follower(O, capn, c)
out = it.iterate(fol)
T_=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
plan = ops.RecoPlan(N, N, 2)
def dev_reco(F,T):
    F=T_(F.copy()); T=T_(T.copy()); ops.normalise_TF(F,T,P)
    m = plan.reconstruct(F,T,rU,joinHalf=True,MAP=False,gridCorr=True); return m.cpu().numpy(), plan.last_iters
def or_reco(F,T):
    F=F.copy(); T=T.copy(); O.normalise_TF(F,T,P)
    m,itn,d,_=O.reconstruct(F,T,P,N,2,rU,MAP=False,joinHalf=True,gridCorr=True,return_iters=True); return m,itn
Fo,To = out["F_raw"][0], out["T_raw"][0]
Fd,Td = capn["Fraw"][0], capn["Traw"][0]
m_oo,i1 = or_reco(Fo,To); m_do,i2 = dev_reco(Fo,To); m_dd,i3 = dev_reco(Fd,Td); m_od,i4 = or_reco(Fd,Td)
rel=lambda a,b: np.abs(a-b).max()/np.abs(b).max()
print("rounds", i1,i2,i3,i4)
print("dev reco(oracle FT) vs oracle reco(oracle FT): %.2e" % rel(m_do,m_oo))
print("dev reco(dev FT)    vs oracle reco(dev FT):    %.2e" % rel(m_dd,m_od))
print("oracle reco(dev FT) vs oracle reco(oracle FT): %.2e" % rel(m_od,m_oo))
print("dev reco(dev FT)    vs dev reco(oracle FT):    %.2e" % rel(m_dd,m_do))
# where do F/T differ relative to the voxel's own T
nz = To>0
relT = np.abs(Td-To)[nz]/To[nz]
print("T rel err: median %.2e 99%% %.2e max %.2e ; at voxels with T<1e-3 max: median %.2e max %.2e" % (np.median(relT), np.percentile(relT,99), relT.max(), np.median(relT[To[nz]<1e-3*To.max()]), relT[To[nz]<1e-3*To.max()].max()))
print("voxels T_or>0 & T_dev==0:", int(((To>0)&(Td==0)).sum()), " T_or==0 & T_dev>0:", int(((To==0)&(Td>0)).sum()), "of", int(nz.sum()))
z = (To>0)&(Td==0)
if z.any(): print("  largest oracle T where device has 0: %.3g of max; F there %.3g of max" % (To[z].max()/To.max(), np.abs(Fo[z]).max()/np.abs(Fo).max()))
