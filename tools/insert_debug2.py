import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd import capi, ops, synth
from thunder_amd.refine import pixel_list
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for N, nImg, mReco, scale in ((64, 300, 20, 1.0), (64, 300, 20, 1e4), (128, 500, 100, 3e3), (64, 6, 8, 1.0)):
    P = 2 * N
    rng = np.random.default_rng(1)
    pl = pixel_list(N, N // 2 - 2, 0)
    q0 = synth.random_quats(nImg, rng)
    quat = synth.perturb_quats(q0, mReco, 0.03, rng)
    tran = rng.normal(0, 1, size=(nImg, mReco, 2))
    dat = ((rng.normal(size=(nImg, pl["nPxl"])) + 1j * rng.normal(size=(nImg, pl["nPxl"]))) * scale).astype(np.complex64)
    dat[:, 0] *= 50     # a dominant DC pixel, as in real image FTs
    ctf = rng.uniform(-1, 1, size=(nImg, pl["nPxl"])).astype(np.float32)
    w = np.full(nImg, 1.0 / mReco, np.float32)
    rot = ops.rotmat(T(quat.reshape(-1, 4))).reshape(nImg, mReco, 9)
    out = {}
    for plain in ("1", "0"):
        os.environ["THX_INSERT_PLAIN"] = plain
        capi.call("thx_knobs_reload")
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        for b0 in range(0, nImg, 175):
            b1 = min(nImg, b0 + 175)
            ops.insert(F, Tt, P, T(dat[b0:b1]), T(ctf[b0:b1]), T(w[b0:b1]), rot[b0:b1].contiguous(), T(tran[b0:b1]), T(pl["iCol"]), T(pl["iRow"]), 2, N)
        out[plain] = (F, Tt)
    eF = ((out["0"][0] - out["1"][0]).abs().max() / out["1"][0].abs().max()).item()
    eT = ((out["0"][1] - out["1"][1]).abs().max() / out["1"][1].abs().max()).item()
    print("N %d nImg %d mReco %d scale %g : eF %.2e eT %.2e  sumT win %.4f plain %.4f" % (N, nImg, mReco, scale, eF, eT, out["0"][1].sum().item(), out["1"][1].sum().item()), flush=True)
