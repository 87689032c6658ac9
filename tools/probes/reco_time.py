"""timing aid: one grid-corrected reconstruction (gridding-weight iteration + final transform) on analytic inputs,
hand-written FFT passes vs THX_FFT=rocfft, for a given box size"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from thunder_amd import ops, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
P = 2 * N
plan = ops.RecoPlan(N, N, 2)
vol = plan.set_projectee(torch.from_numpy(synth.blob_map(N, nblob=12)).to(dev))
ax = torch.fft.fftfreq(P, d=1.0 / P, device=dev)
r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :P // 2 + 1] ** 2)
Tt = (1.0 / (1.0 + r / 8.0)).to(torch.float32).contiguous()
Tt[r >= (N // 2 - 2) * 2 + 1] = 0
F = (vol * Tt).contiguous()
del r
from thunder_amd import capi
modes = ("rocfft", "hand_natural", "hand", "hand_waves8", "hand_waves4") + (("hand_x2", "hand_x2_natural") if N == 512 else ())
ref = None
for mode in modes:
    for k, v in (("THX_FFT", "rocfft" if mode == "rocfft" else None), ("THX_RECO_WT", "natural" if mode in ("hand_natural", "hand_x2_natural") else None),
                 ("THX_FFTZ_WAVES", {"hand_waves8": "8", "hand_waves4": "4", "hand_x2": "16", "hand_x2_natural": "16"}.get(mode))):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    capi.call("thx_knobs_reload")
    plan.reconstruct(F.clone(), Tt.clone(), N // 2 - 2, MAP=False, gridCorr=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = plan.reconstruct(F.clone(), Tt.clone(), N // 2 - 2, MAP=False, gridCorr=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if mode == "hand":
        ref = m
    elif mode.startswith("hand") and ref is not None:
        print("    %s == hand bit for bit: %s" % (mode, bool(torch.equal(m, ref))))
    print("N = %d  %s: %.1f ms, %d rounds -> %.2f ms per round" % (N, mode, dt * 1e3, plan.last_iters, dt * 1e3 / plan.last_iters))
