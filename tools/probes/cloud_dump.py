"""dump the particle filter's support-point clouds the E-step kernel actually sees (quaternions after Particle::perturb, per
phase) for a few images in the second iteration of the bench workload, plus the draws of the insertion -> gpurun_out/clouds.npz
(input of tools/brick_model.py: which fraction of the samples an LDS-staged sub-volume of a given margin would serve)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n, keep = int(os.environ.get("THX_PROBE_PARTICLES", "1024")), 128
sh = RefineShard(256, n, dev)
sh.run(1)
ops, st = sh.ops, sh.pf_state
out = {}
for vi in (0,):
    sh.refresh_rows(vi)
    lo, hi = sh.ranges[sh.halves[vi]]
    if sh.use_packed and sh.cells is None:
        sh.cells = ops.pack_projector(sh.vols, sh.P)
    sl = slice(lo, hi)
    for p in range(sh.nPhase):
        sh.pf_call += 1
        f = sh.pfL if p == 0 else sh.pfS
        ops.pf_perturb(st["r"][sl], st["t"][sl], st["wR"][sl], st["wT"][sl], st["k"][sl], st["s"][sl], f, f, sh.transS, sh.transQ,
                       sh.pf_seed, sh.pf_call)
        out["quat_phase%d" % p] = st["r"][lo:lo + keep].cpu().numpy()
        out["tran_phase%d" % p] = st["t"][lo:lo + keep].cpu().numpy()
        rotB = ops.rotmat(st["r"][sl].reshape(-1, 4)).reshape(hi - lo, sh.mLR, 9)
        r = ops.expect_local(sh.cells[vi:vi + 1], sh.P, sh.pf, sh.N, sh.iCol, sh.iRow, sh.datP[sl], sh.ctfP[sl], sh.sigRcpP[sl], rotB,
                             st["t"][sl], nD=1, pR=st["wR"][sl], pT=st["wT"][sl], packed=True)
        sh.pf_call += 1
        ops.pf_update(st["r"][sl], st["t"][sl], st["wR"][sl], st["wT"][sl], r.wR, r.wT, st["k"][sl], st["s"][sl], st["topR"][sl],
                      st["topT"][sl], sh.peakFactorR, sh.pf_seed, sh.pf_call)
    rot, tran = sh.draw_reco(vi, r.wR, r.wT)
    out["reco_rot"] = rot[:keep].cpu().numpy()
    out["true_quat"] = sh.quat[lo:lo + keep]
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/clouds.npz", **out)
q = out["quat_phase0"]
print("dumped", {k: v.shape for k, v in out.items()})
