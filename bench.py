#!/usr/bin/env python
"""bench.py -- particles/sec of one refinement iteration of the E/M hot path on N MI355X GPUs.

One "step" = one full iteration over the rank's HBM-resident shard of synthetic particles:
  row gathers from the masked image stack, nPhase particle-filter phases (mLR rotations x mLT shifts each), sigma
  update, mReco insertions per particle, half-set reduce (RCCL when N > 2), prepareTF, reconstruct (MAP off) -> FSC ->
  reconstruct (MAP on), projector refresh, re-centring and re-masking (rocFFT 2-D) of the particle images.
Workload = BASELINE.json configs[1] ("10k synthetic 256^3 particles, 3D refinement, 1xMI355X") per GPU, weak scaling.
Launch: `python bench.py` (N=1) or
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INSERT_BYTES_PER_PIXEL_SAMPLE = 204  # 12 B in + F 8x8 Bx(R+W) + T 8x4 Bx(R+W)
EXPECT_BYTES_PER_PIXEL_SAMPLE = 64   # 8 neighbours x 8 B


def cpu_baseline(args, shard):
    """oracle (`kind: port`) on a bounded sample of the SAME workload, all host cores (OpenMP over images)."""
    from oracle import oracle as O
    import torch
    cores = os.cpu_count() or 1
    n = min(shard.nImg, max(4, int(args.cpu_particles) if args.cpu_particles else cores))
    pl = O.pixel_list(shard.N, shard.rU, 2, shard.pf)
    assert pl["nPxl"] == shard.nPxl
    shard.reset_reference()
    vol = shard.vols[0].cpu().numpy()
    dat = shard.datP[:n].cpu().numpy()
    ctf = shard.ctfP[:n].cpu().numpy()
    sig = shard.sigRcpP[:n].cpu().numpy()
    if shard.use_pf:   # the particle filter's current support points, the same work in every phase
        st = shard.pf_state
        r1 = shard.ops.rotmat(st["r"][:n].reshape(-1, 4)).reshape(n, shard.mLR, 9)
        rot = torch.stack([r1] * shard.nPhase, dim=1).cpu().numpy()
        tran = torch.stack([st["t"][:n]] * shard.nPhase, dim=1).cpu().numpy()
    else:
        rot = torch.stack([r[:n] for r in shard.rotP], dim=1).cpu().numpy()      # [n][nPhase][mLR][9]
        tran = torch.stack([t[:n] for t in shard.tranP], dim=1).cpu().numpy()    # [n][nPhase][mLT][2]
    rng = np.random.default_rng(1)
    iR = rng.integers(0, shard.mLR, size=(n, shard.mReco))
    iT = rng.integers(0, shard.mLT, size=(n, shard.mReco))
    recoRot = np.ascontiguousarray(np.take_along_axis(rot[:, -1], iR[:, :, None], axis=1))
    recoTran = np.ascontiguousarray(np.take_along_axis(tran[:, -1], iT[:, :, None], axis=1))
    P = shard.P
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    T = np.zeros((P, P, P // 2 + 1), np.float32)
    t0 = time.perf_counter()
    O.baseline_block(vol, P, shard.pf, shard.N, pl, dat, ctf, sig, rot, tran, recoRot, recoTran, F, T)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "particles/s", "cores": cores, "kind": "port",
            "sample": "%d particles x (%d phases x %d rot x %d shifts + %d inserts), oracle C port with OpenMP over "
                      "images, E-step + insertion only (no reconstruct), %.1f s" % (n, shard.nPhase, shard.mLR,
                                                                                   shard.mLT, shard.mReco, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--box", type=int, default=256)
    ap.add_argument("--particles", type=int, default=10000, help="particles per GPU (weak scaling)")
    ap.add_argument("--mLR", type=int, default=125)
    ap.add_argument("--mLT", type=int, default=9)
    ap.add_argument("--phases", type=int, default=3)
    ap.add_argument("--mReco", type=int, default=100)
    ap.add_argument("--batch", type=int, default=10240, help="max images per kernel launch")
    ap.add_argument("--fixed-support", action="store_true",
                    help="feed fixed, tightly clustered support points instead of running the particle filter")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-particles", type=int, default=0)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook: THX_BENCH_ONE_DEVICE=1 runs every rank on GPU 0 with the gloo backend, so the multi-rank control flow
    # (half-set groups, F/T all-reduce, half-map exchange, max-over-ranks timing) can be exercised on a 1-GPU box
    one_dev = os.environ.get("THX_BENCH_ONE_DEVICE", "0") == "1"
    if one_dev:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from thunder_amd import capi
    from thunder_amd.refine import RefineShard
    capi.load()

    shard = RefineShard(args.box, args.particles, dev, rank=rank, world=world, mLR=args.mLR, mLT=args.mLT,
                        nPhase=args.phases, mReco=args.mReco, batch=args.batch, particle_filter=not args.fixed_support)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        shard.reset_reference()
        shard.run(args.warmup)
    shard.reset_reference()
    shard.insert_ms.clear()
    shard.expect_ms.clear()
    shard.stage_ms.clear()
    barrier()
    t0 = time.perf_counter()
    fsc = shard.run(args.steps, timed=True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total_particles = args.particles * world * args.steps
        value = total_particles / dt
        # dominant kernel: the insertion scatter; per-launch average from HIP events on the launch stream
        ins = [(a.elapsed_time(b), n) for a, b, n in shard.insert_ms]
        exp = [(a.elapsed_time(b), n) for a, b, n in shard.expect_ms]
        ins_ms = float(np.mean([m for m, _ in ins]))
        ins_bytes = float(np.mean([n for _, n in ins])) * shard.mReco * shard.nPxlM * INSERT_BYTES_PER_PIXEL_SAMPLE
        exp_ms = float(np.mean([m for m, _ in exp]))
        exp_bytes = float(np.mean([n for _, n in exp])) * shard.mLR * shard.nPxl * EXPECT_BYTES_PER_PIXEL_SAMPLE
        t_ins, t_exp = sum(m for m, _ in ins), sum(m for m, _ in exp)
        if t_ins >= t_exp:
            kname, kms, kbytes = "k_insert_win", ins_ms, ins_bytes
        else:
            kname, kms, kbytes = "k_expect_local", exp_ms, exp_bytes
        achieved = kbytes / (kms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the committed PMC profile (FETCH_SIZE + WRITE_SIZE, separate rocprofv3
        # passes, corrected as MI355X_MICROARCH.md prescribes), scaled to this run's images per launch; null if absent
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("box") == args.box:
                    nper = float(np.mean([n for _, n in (exp if kname == "k_expect_local" else ins)]))
                    if kname == "k_expect_local":
                        traffic = j["hbm_bytes_per_image_phase"] * nper
                    else:
                        traffic = j["insert_hbm_bytes_per_image"] * nper
            except Exception:
                traffic = None
        out = {
            "metric": "particles/sec per refinement iteration (256^3 box); achieved HBM GB/s",
            "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic %d^3 particles per GPU, 3D refinement iteration "
                                   "(local search %d phases x %d rot x %d shifts, %d inserts, 2 half-sets, "
                                   "2x reconstruct per half, FSC, projector refresh)" % (
                                       args.particles, args.box, args.phases, args.mLR, args.mLT, args.mReco),
                       "box": args.box, "particles_per_gpu": args.particles, "nPxl": shard.nPxl, "pf": 2,
                       "search_state": "fixed seeded support points" if args.fixed_support else
                                       "device particle filter (perturb / resample every phase, Philox-seeded)",
                       "parallelism": "particles sharded over %d GPU(s); half-set F/T all-reduce" % world},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": kms,
                         "algorithmic_bytes_per_launch": kbytes,
                         # tools/gather_probe.hip on MI355X: 64-byte gathers by the 64 lanes of a wave within +-R voxels of
                         # a walking centre (the E-step's pattern without any cross-sample reuse), GB/s for R = 2 / 4 / 8
                         "random_64B_gather_GBps": {"R2": 7000.0, "R4": 4190.0, "R8": 3530.0}},
            "kernels": {"k_insert_win": {"avg_launch_ms": ins_ms, "GBps_algorithmic": ins_bytes / (ins_ms * 1e-3) / 1e9,
                                     "total_ms": t_ins},
                        "k_expect_local": {"avg_launch_ms": exp_ms,
                                           "GBps_algorithmic": exp_bytes / (exp_ms * 1e-3) / 1e9, "total_ms": t_exp}},
            "stages_ms_per_step": {k: round(sum(a.elapsed_time(b) for a, b in v) / args.steps, 2)
                                   for k, v in shard.stage_ms.items()},
            "fsc_half_maps": [round(float(x), 4) for x in fsc[: 8]],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, shard)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
