"""Probe: what would cross-image sharing of the E-step's gathers buy AT BEST?  The local-search kernel is timed on one batch of
view-ordered images with (A) the filter's own clouds, (B) every image using image 0's cloud (all resident workgroups touch the same
cells at the same time: perfect coherence, the ceiling of any co-scheduling scheme), (C) groups of 8 consecutive images sharing a
cloud, (D) image 0's cloud turned by a random angle of up to `deg` degrees per image.  usage: python tools/coherence_probe.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from thunder_amd import ops
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sh = RefineShard(256, 2 * n, dev, batch=n)
sh.run(2)
sh.refresh_rows(0)
lo, hi = sh.ranges[sh.halves[0]]
m = hi - lo
st = sh.pf_state
if sh.cells is None:
    sh.cells = ops.pack_projector(sh.vols, sh.P)
# the clouds as a phase sees them: resampled by the previous phase, then perturbed (Particle::perturb, factor 0.5 in phases 2 / 3,
# 2 in phase 1) -- 125 distinct rotations per image
f = float(os.environ.get("PERTURB", "0.5"))
r0, t0, wR, wT = st["r"][lo:hi].clone(), st["t"][lo:hi].clone(), st["wR"][lo:hi].clone(), st["wT"][lo:hi].clone()
ops.pf_perturb(r0, t0, wR, wT, st["k"][lo:hi].clone(), st["s"][lo:hi].clone(), f, f, sh.transS, sh.transQ, 12345, 1)
print("perturbation factor %.1f; distinct rotations per image: %.1f" % (f, np.mean([len(np.unique(r0[i].cpu().numpy(), axis=0)) for i in range(0, m, max(1, m // 32))])))


def qmul(a, b):
    w1, x1, y1, z1 = a.unbind(-1); w2, x2, y2, z2 = b.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)


def run(name, r, perm=None):
    rot = ops.rotmat(r.reshape(-1, 4).contiguous()).reshape(m, sh.mLR, 9)
    dat, ctf, sig, tt, pR_, pT_ = sh.datP[lo:hi], sh.ctfP[lo:hi], sh.sigRcpP[lo:hi], t0, wR, wT
    if perm is not None:
        rot, dat, ctf, sig, tt, pR_, pT_ = [x[perm].contiguous() for x in (rot, dat, ctf, sig, tt, pR_, pT_)]
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t_ = time.perf_counter()
        ops.expect_local(sh.cells[0:1], sh.P, sh.pf, sh.N, sh.iCol, sh.iRow, dat, ctf, sig, rot, tt,
                         pR=pR_, pT=pT_, workspace=sh.ws[0], packed=True, wg_per_cu=int(os.environ.get("WG", "2")))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t_)
    print("%-58s %7.1f ms for %d images = %6.2f us per image-phase" % (name, best * 1e3, m, best / m * 1e6), flush=True)


run("A  the filter's own clouds (view-ordered images)", r0)
if os.environ.get("E13") == "1":     # round 6, Appendix A E13: only A and B (the kernel's non-memory floor), for the counter passes
    run("B  every image on image 0's cloud", r0[:1].expand(m, -1, -1).contiguous())
    sys.exit(0)
# E: the same images and clouds as A, stored so that the workgroups of ONE XCD (workgroup i -> XCD i mod 8, 64 resident per XCD at 2
# per CU) are 64 view-neighbours: position 8 (64 q + r) + x holds image (8 q + x) 64 + r
for blk in (64, 16):
    if m % (8 * blk) == 0:
        pos = torch.arange(m, device=dev)
        x, j = pos % 8, pos // 8
        src = ((j // blk) * 8 + x) * blk + (j % blk)
        run("E  A's images, %d view-neighbours per XCD at a time" % blk, r0, perm=src)
run("B  every image on image 0's cloud", r0[:1].expand(m, -1, -1).contiguous())
run("C  groups of 8 consecutive images share a cloud", r0[::8].repeat_interleave(8, dim=0)[:m].contiguous())
g = torch.Generator(device=dev); g.manual_seed(3)
for deg in (0.25, 1.0, 3.0):
    ax = torch.randn((m, 3), generator=g, device=dev, dtype=torch.float64); ax = ax / ax.norm(dim=1, keepdim=True)
    ang = torch.rand((m,), generator=g, device=dev, dtype=torch.float64) * np.radians(deg)
    dq = torch.cat([torch.cos(ang / 2)[:, None], ax * torch.sin(ang / 2)[:, None]], 1)          # [m][4]
    run("D  image 0's cloud turned by up to %.2f degrees per image" % deg, qmul(dq[:, None, :], r0[:1].expand(m, -1, -1)).contiguous())
