"""CTF rows, device vs oracle: where do the differences come from? (chi, the cos(2 angle) term, or sinf/cosf of chi)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from thunder_amd import ops, synth
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for N in (64, 256):
    rng = np.random.default_rng(3)
    pl = O.pixel_list(N, N // 2 - 2, 0)
    attr = synth.ctf_params(8, rng)
    got = ops.ctf(T(attr), 1.32, T(pl["iCol"]), T(pl["iRow"]), N).cpu().numpy()
    want = np.stack([O.ctf(1.32, *a, N, pl["iCol"], pl["iRow"]) for a in attr])
    d = np.abs(got - want)
    # float64 evaluation of the same formula from the same float inputs
    i, j = pl["iCol"].astype(np.float64), pl["iRow"].astype(np.float64)
    ex = []
    for a in attr.astype(np.float64):
        V, dU, dV, th, Cs, A, ph = a
        lam = 12.2643247 / np.sqrt(V * (1 + V * 0.978466e-6))
        u = np.hypot(i / (np.float32(1.32) * N), j / (np.float32(1.32) * N))
        ang = np.arctan2(j, i) - th
        de = -(dU + dV + (dU - dV) * np.cos(2 * ang)) / 2
        chi = np.pi * lam * de * u * u + np.pi / 2 * Cs * lam ** 3 * u ** 4 - ph
        ex.append(-np.sqrt(1 - A * A) * np.sin(chi) + A * np.cos(chi))
    ex = np.stack(ex)
    print("N %d: |dev - oracle| max %.2e  p99 %.2e  mean %.2e ; |dev - f64| max %.2e mean %.2e ; |oracle - f64| max %.2e mean %.2e ; chi max %.0f rad (ulp %.1e)" % (
        N, d.max(), np.percentile(d, 99), d.mean(), np.abs(got - ex).max(), np.abs(got - ex).mean(), np.abs(want - ex).max(), np.abs(want - ex).mean(),
        np.abs(chi).max(), np.spacing(np.float32(np.abs(chi).max()))))
