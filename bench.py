#!/usr/bin/env python
"""bench.py -- particles/sec of one refinement iteration of the E/M hot path on N MI355X GPUs.

One "step" = one full iteration over the HBM-resident synthetic particles:
  row gathers from the masked image stack, nPhase particle-filter phases (mLR rotations x mLT shifts each, perturb /
  resample on the device), noise normalisation of the image stacks (Optimiser::normCorrection, from the second iteration on), sigma
  update, mReco insertions per particle, half-set reduce (RCCL when a half spans more
  than one rank), prepareTF, reconstruct (MAP off) -> FSC -> reconstruct (MAP on), projector refresh, re-centring and
  re-masking (rocFFT 2-D) of the particle images.
Workload = the configuration BASELINE.json's metric is quoted on: 100 000 synthetic 256^3 particles, 3-D refinement.
It fits one GPU (about 125 GB of particle data + 15 GB of volumes in 288 GB), so N = 1 runs all of it; with N > 1 the same
100 000 particles are sharded over the ranks (N = 8 is BASELINE configs[2]: 12 500 per GPU) -- total work fixed:
"scaling": "strong".  `--particles 10000` gives configs[1].
`--classification` times BASELINE configs[3] instead: one whole K = 4 classification iteration on one GPU's share of the images,
through the same native driver (thx_refine_iterate with nK = 4 and a global search).  The default command (the headline
workload on one GPU) then also runs configs[1], configs[3] and configs[4] for 2 + 1 iterations each and reports them under
`other_configs` of the same JSON line (`--other-configs off` skips them).
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

INIT_OUTSIDE_CONFIDENCE_AREA = 0.5   # include/Particle.h:59
TRANS_SEARCH_FACTOR = 0.25           # script/demo_3D.json "Translation Search Factor"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INSERT_ALGORITHMIC_BYTES_PER_PIXEL_DRAW = 204  # 12 B in + F 8x8 Bx(R+W) + T 8x4 Bx(R+W) per (listed pixel, draw); used ONLY for `step_algorithmic_TBps` (see there)
EXPECT_BYTES_PER_PIXEL_SAMPLE = 64   # 8 neighbours x 8 B
EXPECT_BYTES_PER_PIXEL = 16          # dat 8 + ctf 4 + sigRcp 4, read once per image-phase


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, shard, nat, rounds_per_iteration):
    """oracle (`kind: port`) on a bounded sample of the SAME workload, all host cores.
    E-step + insertion: `n` particles of the shard -- their rows, noise model and the particle filter's current support
    points copied out of the native driver's HBM state -- through the oracle's OpenMP loop over images (one image per
    thread at a time, F / T shared under `omp atomic`: the reference's own structure, src/Optimiser.cpp:1162,7038).
    Reconstruct leg: the oracle's gridding reconstruction at the full 512^3 grid is timed for 1 and for 3 balancing rounds
    (scipy pocketfft on all cores + the oracle's C sweeps); fixed cost and cost per round follow by difference and are
    scaled to the 4 reconstructions and the number of balancing rounds the GPU iteration actually ran."""
    from oracle import oracle as O
    import scipy.fft as sfft
    cores = os.cpu_count() or 1
    n = min(shard.nImg, max(4, int(args.cpu_particles) if args.cpu_particles else 2 * cores))
    v = nat.view()
    nPxl, P, N = v.nPxl, shard.P, shard.N
    iCol, iRow = nat.fetch(v.iCol, np.int32, (nPxl,)), nat.fetch(v.iRow, np.int32, (nPxl,))
    pl = dict(iCol=iCol, iRow=iRow, iColPad=iCol * shard.pf, iRowPad=iRow * shard.pf, nPxl=nPxl)
    vol = nat.fetch(v.vols, np.complex64, (P, P, P // 2 + 1))
    dat = nat.fetch(v.datP, np.complex64, (n, nPxl))
    ctf = nat.fetch(v.ctfP, np.float32, (n, nPxl))
    sig = nat.fetch(v.sigRcpP, np.float32, (n, nPxl))
    quat = nat.fetch(v.r, np.float64, (n, shard.mLR, 4))
    t1 = nat.fetch(v.t, np.float64, (n, shard.mLT, 2))
    r1 = np.stack([[O.rotate3D(q) for q in qs] for qs in quat])          # the filter's current support points,
    rot = np.ascontiguousarray(np.stack([r1] * shard.nPhase, axis=1))    # the same work in every phase
    tran = np.ascontiguousarray(np.stack([t1] * shard.nPhase, axis=1))
    rng = np.random.default_rng(1)
    iR = rng.integers(0, shard.mLR, size=(n, shard.mReco))
    iT = rng.integers(0, shard.mLT, size=(n, shard.mReco))
    recoRot = np.ascontiguousarray(np.take_along_axis(rot[:, -1], iR[:, :, None], axis=1))
    recoTran = np.ascontiguousarray(np.take_along_axis(tran[:, -1], iT[:, :, None], axis=1))
    # thread layout: the reference is deployed as several MPI ranks per node, each an OpenMP team with PRIVATE F / T that
    # MPI_Allreduce_Large sums afterwards (src/Parallel.cpp:26-36, src/Reconstructor.cpp:2383,2436) -- `groups` teams of
    # cores / groups threads here, their accumulators added up inside the timed region.  --cpu-shared also times the
    # single-team form (every thread on ONE F / T under `omp atomic`) that round 2 reported.
    groups = max(1, min(int(args.cpu_groups), cores))
    F = np.zeros((groups, P, P, P // 2 + 1), np.complex64)
    T = np.zeros((groups, P, P, P // 2 + 1), np.float32)
    # the sample runs as THREE consecutive parts, each timed on its own: value = the median of the three rates, with their spread
    # (one 256-thread run of a box shared with the GPU process moved by +-40 % between rounds 4 and 5)
    parts = [p_ for p_ in np.array_split(np.arange(n), 3 if n >= 3 * groups else 1) if len(p_)]
    rates, t_em = [], 0.0
    for p_ in parts:
        a_, b_ = int(p_[0]), int(p_[-1]) + 1
        t0 = time.perf_counter()
        O.baseline_block(vol, P, shard.pf, N, pl, dat[a_:b_], ctf[a_:b_], sig[a_:b_], rot[a_:b_], tran[a_:b_], recoRot[a_:b_],
                         recoTran[a_:b_], F, T, groups=groups)
        Fsum, Tsum = F.sum(axis=0), T.sum(axis=0)      # the hemisphere all-reduce of the groups' volumes
        dt_ = time.perf_counter() - t0
        rates.append((b_ - a_) / dt_)
        t_em += dt_
    em_rate = float(np.median(rates))
    shared = None
    if args.cpu_shared:
        n_sh = min(n, cores)
        F1 = np.zeros((P, P, P // 2 + 1), np.complex64)
        T1 = np.zeros((P, P, P // 2 + 1), np.float32)
        t0 = time.perf_counter()
        O.baseline_block(vol, P, shard.pf, N, pl, dat[:n_sh], ctf[:n_sh], sig[:n_sh], rot[:n_sh], tran[:n_sh], recoRot[:n_sh],
                         recoTran[:n_sh], F1, T1, groups=1)
        shared = {"particles": n_sh, "em_particles_per_s": n_sh / (time.perf_counter() - t0)}
        del F1, T1
    del F, T, Fsum, Tsum
    # ---- reconstruct leg (per iteration, independent of the particle count): the GPU's own F / T of half 0 ----
    reco = None
    if not args.cpu_no_reconstruct:
        Fh = nat.fetch(v.F, np.complex64, (P, P, P // 2 + 1))
        Th = np.maximum(nat.fetch(v.T, np.float32, (P, P, P // 2 + 1)), 0)
        with sfft.set_workers(cores):
            O.reconstruct(Fh, Th, P, N, shard.pf, shard.maxRadius, MAP=False, gridCorr=True, max_rounds=1)   # untimed: first touch
            t0 = time.perf_counter()
            O.reconstruct(Fh, Th, P, N, shard.pf, shard.maxRadius, MAP=False, gridCorr=True, max_rounds=1)
            t1_ = time.perf_counter() - t0
            t0 = time.perf_counter()
            O.reconstruct(Fh, Th, P, N, shard.pf, shard.maxRadius, MAP=False, gridCorr=True, max_rounds=3)
            t3_ = time.perf_counter() - t0
        per_round = max(0.0, (t3_ - t1_) / 2.0)
        fixed = max(0.0, t1_ - per_round)
        reco = {"fixed_s": fixed, "per_round_s": per_round, "rounds_per_iteration": rounds_per_iteration,
                "reconstructions_per_iteration": 4,
                "seconds_per_iteration": 4 * fixed + rounds_per_iteration * per_round}
    n_total = args.particles
    t_iter = n_total / em_rate + (reco["seconds_per_iteration"] if reco else 0.0)
    return {"value": n_total / t_iter, "unit": "particles/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "em_particles_per_s": em_rate, "em_particles_per_s_runs": [round(r_, 3) for r_ in rates],
            "em_spread": (max(rates) - min(rates)) / em_rate if em_rate > 0 else None, "reconstruct": reco,
            "thread_layout": "%d groups x %d threads, private F / T per group, summed at the end" % (groups, max(1, cores // groups)),
            "single_team_shared_FT": shared,
            "note": "a reported baseline, not the target: GPU / CPU says nothing about kernel quality, roofline.frac does",
            "sample": "E-step + insertion: %d particles x (%d phases x %d rot x %d shifts + %d inserts) through the oracle C "
                      "port, OpenMP over images on %d threads in %d groups with private F / T, in three consecutive parts timed separately "
                      "(median rate, `em_spread` = (max - min) / median), %.1f s in all; reconstruct leg: oracle gridding reconstruction on "
                      "the %d^3 grid timed for 1 and 3 balancing rounds (scipy pocketfft, %d workers), scaled to 4 "
                      "reconstructions / %d rounds per iteration; value = %d particles / (particles / EM rate + "
                      "reconstruct seconds)" % (n, shard.nPhase, shard.mLR, shard.mLT, shard.mReco, cores, groups, t_em, P, cores,
                                                rounds_per_iteration, n_total)}


def bench_global_scan(args, dev):
    """--classification: the global scanning stage of BASELINE configs[3] (3-D classification, K = 4) on one GPU's share of
    images: every image against 4 classes x 10 000 rotations x 30 shifts at the scan radius r = 24 (866 pixels, rL = 2)
    -- Optimiser::expectation's global loop (src/Optimiser.cpp:756-894), logDataVSPrior_m_n_huabin (:9931-9973).
    Per class: slices of the class volume for all rotations (thx_project_dev), then thx_expect_global_dev streams every
    image's rows against them.  This stage is a dense contraction over pixels (2 FMAs per pixel x rotation x shift), so
    its bound is the fp32 FMA rate of the chip -- 157.3 TFLOP/s, vector or f32 MFMA alike (MI355X_MICROARCH.md) --, not HBM."""
    import torch
    from thunder_amd import ops, synth
    from thunder_amd.refine import pixel_list
    N, K, nR, nT, rScan = args.box, 4, 10000, 30, 24
    nImg = args.scan_images
    P = 2 * N
    rng = np.random.default_rng(4)
    pl = pixel_list(N, rScan, 2)
    iCol, iRow = torch.from_numpy(pl["iCol"]).to(dev), torch.from_numpy(pl["iRow"]).to(dev)
    nPxl = pl["nPxl"]
    plan = ops.RecoPlan(N, N, 2)
    vols = [plan.set_projectee(torch.from_numpy(synth.blob_map(N, seed=300 + k, nblob=20)).to(dev)) for k in range(K)]
    mats = ops.rotmat(torch.from_numpy(synth.random_quats(nR, rng)).to(dev))
    shifts = np.ascontiguousarray(rng.normal(0, 3.0, size=(nT, 2)))
    traP = ops.translate(torch.from_numpy(shifts).to(dev), iCol, iRow, N)
    attr = torch.from_numpy(synth.ctf_params(nImg, rng)).to(dev)
    ctf = ops.ctf(attr, 1.32, iCol, iRow, N)
    # images: a slice of a random class at a random scanned rotation / shift, plus noise
    cls_true, r_true, t_true = rng.integers(0, K, nImg), rng.integers(0, nR, nImg), rng.integers(0, nT, nImg)
    dat = torch.empty((nImg, nPxl), dtype=torch.complex64, device=dev)
    for k in range(K):
        sel = np.nonzero(cls_true == k)[0]
        if len(sel) == 0:
            continue
        rot_k = mats[torch.from_numpy(r_true[sel]).to(dev)].contiguous()
        sl = ops.project(vols[k], rot_k, iCol, iRow, 2)
        s = torch.from_numpy(sel).to(dev)
        dat[s] = sl * traP[torch.from_numpy(t_true[sel]).to(dev)] * ctf[s]
    sd = 3.0 * float(dat.abs().pow(2).mean().sqrt())
    g = torch.Generator(device=dev); g.manual_seed(5)
    dat = (dat + torch.view_as_complex(torch.randn((nImg, nPxl, 2), generator=g, device=dev)) * (sd / np.sqrt(2))).contiguous()
    sigRcp = torch.full((nImg, nPxl), -0.5 / (sd * sd / 2), dtype=torch.float32, device=dev)
    pR = torch.full((nImg, nR), 1.0 / nR, dtype=torch.float64, device=dev)
    pT = torch.full((nImg, nT), 1.0 / nT, dtype=torch.float64, device=dev)
    wC = torch.zeros((nImg, K), dtype=torch.float32, device=dev)
    wR = torch.zeros((K, nImg, nR), dtype=torch.float32, device=dev)
    wT = torch.zeros((K, nImg, nT), dtype=torch.float32, device=dev)
    base = torch.empty((nImg,), dtype=torch.float32, device=dev)
    from thunder_amd import capi
    ws = torch.empty(capi.load().thx_expect_global_workspace(nImg, nR, nT), dtype=torch.uint8, device=dev)
    rotP = torch.empty((nR, nPxl), dtype=torch.complex64, device=dev)
    ev = []

    def scan(timed):
        wC.zero_(); wR.zero_(); wT.zero_(); base.fill_(float("nan"))
        for k in range(K):
            ops.project(vols[k], mats, iCol, iRow, 2, out=rotP)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.expect_global(rotP, traP, dat, ctf, sigRcp, pR, pT, wC, wR, wT, base, k, K, workspace=ws)
            if timed:
                e1.record()
                ev.append((e0, e1))
    for _ in range(args.warmup):
        scan(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scan(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = float((wC.argmax(1).cpu().numpy() == cls_true).mean())
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    flops = 4.0 * nImg * nR * nT * nPxl          # 2 FMAs per (pixel, rotation, shift) in the expanded likelihood
    out = {"metric": "images/sec through the global scanning stage (K = 4 classes x 10000 rotations x 30 shifts, r = 24)",
           "value": nImg * args.steps / dt, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": "configs[3] global scan: %d synthetic %d^3 images x %d classes x %d rotations x %d shifts at "
                                  "r = %d (%d pixels)" % (nImg, N, K, nR, nT, rScan, nPxl), "classes_recovered": ok},
           "roofline": {"bound": "mfma", "kernel": "thx_expect_global_dev = k_scan_tables + k_scan_gemm (f32 MFMA) + fold (one class: %d images x %d rot x %d shifts)" % (nImg, nR, nT),
                        "achieved": flops / (k_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": flops / (k_ms * 1e-3) / 1e12 / 157.3, "traffic": None, "avg_launch_ms": k_ms,
                        "note": "exact-f32 contraction on v_mfma_f32_32x32x2_f32 (bit-equal to the fmaf chain); peak = f32 MFMA = f32 vector rate"},
           "cpu_baseline": None}
    if not args.no_cpu_baseline:
        # the oracle's restatement of the scanning loop (src/Optimiser.cpp:756-894, logDataVSPrior_m_n_huabin) on a bounded
        # sample: every host core takes a block of images through ONE class (all 10 000 rotations x 30 shifts), the figure is
        # scaled to the K classes.  (The reference splits the rotations over OpenMP threads and locks per image; blocks of
        # images are the same arithmetic without the locks.)
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        cores = os.cpu_count() or 1
        per = max(1, int(args.scan_cpu_images_per_core))
        n_cpu = min(nImg, per * cores)
        ops.project(vols[0], mats, iCol, iRow, 2, out=rotP)
        rotP_h, traP_h = rotP.cpu().numpy(), traP.cpu().numpy()
        dat_h, ctf_h, sig_h = dat[:n_cpu].cpu().numpy(), ctf[:n_cpu].cpu().numpy(), sigRcp[:n_cpu].cpu().numpy()

        def block(b):
            lo, hi = b * per, min(n_cpu, (b + 1) * per)
            m = hi - lo
            wC_ = np.zeros((m, K), np.float32); wR_ = np.zeros((K, m, nR), np.float32); wT_ = np.zeros((K, m, nT), np.float32)
            base_ = np.full(m, np.nan, np.float32)
            O.expect_global(rotP_h, traP_h, np.ascontiguousarray(dat_h[lo:hi].T), np.ascontiguousarray(ctf_h[lo:hi].T),
                            np.ascontiguousarray(sig_h[lo:hi].T), K, 0, np.full((m, nR), 1.0 / nR), np.full((m, nT), 1.0 / nT), wC_, wR_, wT_,
                            base_)
            return int(wR_[0].argmax(1)[0])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(block, range((n_cpu + per - 1) // per)))
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_cpu / (t_cpu * K), "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
                               "sample": "%d images x 1 class x %d rotations x %d shifts through the oracle's scanning loop, %d host "
                                         "threads x %d images each, %.1f s; scaled to %d classes" % (n_cpu, nR, nT, cores, per, t_cpu, K)}
    print(json.dumps(out))


def bench_classification_iteration(args, dev, nImg=None, steps=None, warmup=None, cpu=True):
    """--classification: one whole iteration of BASELINE configs[3] (3-D classification, K = 4 references, global search) on ONE
    GPU's share of the images (50 000 / 8 = 6 250, both half-sets), through the SAME native driver as the refinement line
    (thx_refine_iterate with nK = 4, THX_SEARCH_GLOBAL): rows -> global scan of every image against K classes x 10 000 rotations x
    30 shifts at r = 24 (src/Optimiser.cpp:756-894) -> class of every image (:925-952) -> support points (:953-1079) -> local
    particle-filter phases against the assigned reference (volIdx) -> sigma update -> mReco draws per image -> multi-reference
    insertion (K pairs of F / T in one fixed-point session per half) -> prepareTF -> 2 reconstructions per class and half ->
    per-class FSC (core-mask corrected) -> averaging of the halves -> solvent flattening -> Model::refreshProj.  The same definition of
    "iteration" as the headline line (a global-search iteration has no re-centring / normCorrection in the reference either,
    src/Optimiser.cpp:3405-3413,3790-3810).  `value` = images per second through the whole iteration; the per-stage times and the
    roofline of each stage's dominant kernel ride along."""
    import torch
    from thunder_amd.native import NativeRefine, STAGES
    from thunder_amd.refine import RefineShard
    N, K, nR, nT, rScan = args.box, 4, 10000, 30, 24
    nImg = nImg or args.scan_images
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    if N < 64:
        rScan = max(4, N // 2 - 4)
    shard = RefineShard(N, nImg, dev, mLR=args.mLR, mLT=args.mLT, nPhase=args.phases, mReco=args.mReco, batch=min(args.batch, 3125),
                        particle_filter=True, allocate=False, sort_view=not args.unsorted, snr=0.1, K=K, scan=dict(nR=nR, nT=nT, rScan=rScan),
                        search="global", map_seed=300, nblob=20)
    shard.balanceClass = 1
    shard.release_generation_state()
    shard.wg_per_cu = args.wg_per_cu
    nat = NativeRefine(shard, pixel_order=1)
    for _ in range(warmup):
        nat.reset()
        nat.iterate(False)
    nat.reset()
    torch.cuda.synchronize()
    nat.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        fsc = nat.iterate(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st, v = nat.stats(), nat.view()
    cls = nat.fetch(v.cls, np.int32, (nImg,))
    topR = nat.fetch(v.topR, np.float64, (nImg, 4))
    ok = float((cls == shard.cls_true).mean())
    d = np.abs((topR * shard.quat).sum(1)).clip(0, 1)
    ang = np.degrees(2 * np.arccos(d))
    nPxlS, nPxlE, nPxlM = st.nPxlS, st.nPxl, st.nPxlM
    scan_ms = st.scanMs / max(1, st.scanLaunches)
    scan_n = st.scanImages / max(1, st.scanLaunches)
    loc_ms, loc_n = st.expectMs / max(1, st.expectLaunches), st.expectImages / max(1, st.expectLaunches)
    ins_ms, ins_n = st.insertMs / max(1, st.insertLaunches), st.insertImages / max(1, st.insertLaunches)
    flops = 4.0 * scan_n * nR * nT * nPxlS          # 2 FMAs per (pixel, rotation, shift) in the expanded likelihood, one class
    loc_bytes = loc_n * nPxlE * (EXPECT_BYTES_PER_PIXEL + EXPECT_BYTES_PER_PIXEL_SAMPLE * shard.mLR)
    stages = {k: round(st.stageMs[i] / steps, 2) for i, k in enumerate(STAGES)}
    dominant = max(("global_scan", "expectation"), key=lambda k: stages.get(k, 0.0))
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        j = json.load(open(pmc))
        if j.get("box") == N and "classification_hbm_bytes_per_image_phase" in j:
            traffic = j["classification_hbm_bytes_per_image_phase"] * loc_n
    except Exception:
        pass
    roof_scan = {"bound": "mfma", "kernel": "thx_expect_global_dev = k_scan_tables + k_scan_gemm (f32 MFMA) + fold (one class: %d images x %d rot x %d shifts)" % (scan_n, nR, nT),
                 "achieved": flops / (scan_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / (scan_ms * 1e-3) / 1e12 / 157.3,
                 "traffic": None, "avg_launch_ms": scan_ms,
                 "note": "exact-f32 contraction on v_mfma_f32_32x32x2_f32 (bit-equal to the fmaf chain); peak = f32 MFMA = f32 vector rate"}
    roof_local = {"bound": "hbm", "kernel": "k_expect_local<9, packed> with volIdx (K cell-packed references)", "achieved": loc_bytes / (loc_ms * 1e-3) / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": loc_bytes / (loc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                  "avg_launch_ms": loc_ms, "images_per_launch": loc_n, "traffic_measured_in_run": False,
                  "traffic_profile": "profiles/pmc_traffic.json"}
    out = {"metric": "images/sec through one 3-D classification iteration (K = 4, %d^3 box): global scan + class selection + %d local phases + "
                     "sigma update + multi-reference insertion + 2 reconstructions per class and half + per-class FSC / averaging / refresh" % (N, args.phases),
           "value": nImg * steps / dt, "unit": "images/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[3] on one GPU's share: %d synthetic %d^3 images (both half-sets), K = %d classes; scan %d rotations x %d shifts "
                                  "at r = %d (%d pixels); local search %d phases x %d rot x %d shifts on %d pixels; %d inserts per image into %d F / T pairs "
                                  "per half; %d reconstructions" % (nImg, N, K, nR, nT, rScan, nPxlS, args.phases, shard.mLR, shard.mLT, nPxlE, shard.mReco, K, 4 * K),
                      "classes_recovered": ok, "median_pose_error_deg": float(np.median(ang)), "images_per_class": [int(x) for x in st.classCount[:K]],
                      "particle_order": "random" if args.unsorted else "by class, then view direction",
                      "sequenced_by": "thx_refine_iterate (the one native C++ driver, thx_refine.hip): nK = 4, THX_SEARCH_GLOBAL"},
           "roofline": roof_scan if dominant == "global_scan" else roof_local,
           "rooflines": {"scan": roof_scan, "local_phases": roof_local},
           "kernels": {"insertion (k_bin + segment sort + k_acc), %d classes in one session" % K: {"avg_call_ms": ins_ms, "images_per_call": ins_n,
                                                                                             "us_per_image": ins_ms * 1e3 / max(1.0, ins_n)}},
           "stages_ms_per_step": stages, "balancing_rounds_per_step": st.balancingRounds / max(1, steps),
           "fsc_half_maps_class0": [round(float(x), 4) for x in np.atleast_2d(fsc)[0][:8]], "cpu_baseline": None}
    if cpu and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_classification(args, shard, nat, dict(K=K, nR=nR, nT=nT, rScan=rScan))
    nat.close()
    return out


def cpu_baseline_classification(args, shard, nat, c):
    """oracle (`kind: port`) on bounded samples of the classification iteration, all host cores: the scanning loop (every core a block
    of images through ONE class, scaled to K) and the local phases + insertion (oracle.baseline_block on the filter's support points
    of the sample, one class volume: the arithmetic does not depend on which class an image is in).  The reconstructions are left
    out (the refinement bench times that leg); value = images / (scan seconds + local-and-insertion seconds) for the job's images."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    K, nR, nT, rScan = c["K"], c["nR"], c["nT"], c["rScan"]
    N, P, pf, nImg = shard.N, shard.P, shard.pf, shard.nImg
    cores = os.cpu_count() or 1
    per = max(1, int(args.scan_cpu_images_per_core))
    n_cpu = min(nImg, per * cores)
    v = nat.view()
    plS = O.pixel_list(N, rScan, shard.rL, pf)
    vol = nat.fetch(v.vols, np.complex64, (P, P, P // 2 + 1))
    quat, shifts = shard.scan["quat"], shard.scan["shifts"]
    rotP_h = np.stack([O.project(vol, P, pf, O.rotate3D(q), plS["iCol"], plS["iRow"]) for q in quat])
    traP_h = np.stack([O.translate(np.float32(s[0]), np.float32(s[1]), N, plS["iCol"], plS["iRow"]) for s in shifts])
    img = nat.fetch(v.img, np.complex64, (n_cpu, N, N // 2 + 1))
    attr = shard.attr[:n_cpu].cpu().numpy()
    dat_h = np.ascontiguousarray(img.reshape(n_cpu, -1)[:, plS["iPxl"]])
    ctf_h = np.stack([O.ctf(shard.pixelSize, *attr[l], N, plS["iCol"], plS["iRow"]) for l in range(n_cpu)])
    sig_h = np.full(dat_h.shape, np.float32(-0.5 / shard.sigma2), np.float32)

    def block(b):
        lo, hi = b * per, min(n_cpu, (b + 1) * per)
        m = hi - lo
        wC_ = np.zeros((m, K), np.float32); wR_ = np.zeros((K, m, nR), np.float32); wT_ = np.zeros((K, m, nT), np.float32)
        base_ = np.full(m, np.nan, np.float32)
        O.expect_global(rotP_h, traP_h, np.ascontiguousarray(dat_h[lo:hi].T), np.ascontiguousarray(ctf_h[lo:hi].T),
                        np.ascontiguousarray(sig_h[lo:hi].T), K, 0, np.full((m, nR), 1.0 / nR), np.full((m, nT), 1.0 / nT), wC_, wR_, wT_, base_)
        return int(wR_[0].argmax(1)[0])
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(block, range((n_cpu + per - 1) // per)))
    t_scan = time.perf_counter() - t0
    scan_rate = n_cpu / (t_scan * K)
    # local phases + insertion
    n = min(nImg, max(4, int(args.cpu_particles) if args.cpu_particles else cores))
    nPxl = v.nPxl
    iCol, iRow = nat.fetch(v.iCol, np.int32, (nPxl,)), nat.fetch(v.iRow, np.int32, (nPxl,))
    pl = dict(iCol=iCol, iRow=iRow, iColPad=iCol * pf, iRowPad=iRow * pf, nPxl=nPxl)
    dat, ctf, sig = nat.fetch(v.datP, np.complex64, (n, nPxl)), nat.fetch(v.ctfP, np.float32, (n, nPxl)), nat.fetch(v.sigRcpP, np.float32, (n, nPxl))
    quatL, t1 = nat.fetch(v.r, np.float64, (n, shard.mLR, 4)), nat.fetch(v.t, np.float64, (n, shard.mLT, 2))
    r1 = np.stack([[O.rotate3D(q) for q in qs] for qs in quatL])
    rot = np.ascontiguousarray(np.stack([r1] * shard.nPhase, axis=1))
    tran = np.ascontiguousarray(np.stack([t1] * shard.nPhase, axis=1))
    rng = np.random.default_rng(1)
    iR, iT = rng.integers(0, shard.mLR, size=(n, shard.mReco)), rng.integers(0, shard.mLT, size=(n, shard.mReco))
    recoRot = np.ascontiguousarray(np.take_along_axis(rot[:, -1], iR[:, :, None], axis=1))
    recoTran = np.ascontiguousarray(np.take_along_axis(tran[:, -1], iT[:, :, None], axis=1))
    groups = max(1, min(int(args.cpu_groups), cores))
    F = np.zeros((groups, P, P, P // 2 + 1), np.complex64)
    T = np.zeros((groups, P, P, P // 2 + 1), np.float32)
    t0 = time.perf_counter()
    O.baseline_block(vol, P, pf, N, pl, dat, ctf, sig, rot, tran, recoRot, recoTran, F, T, groups=groups)
    F.sum(axis=0); T.sum(axis=0)
    t_em = time.perf_counter() - t0
    em_rate = n / t_em
    value = 1.0 / (1.0 / scan_rate + 1.0 / em_rate)
    return {"value": value, "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "scan_images_per_s": scan_rate, "local_and_insertion_images_per_s": em_rate,
            "sample": "scan: %d images x 1 class x %d rotations x %d shifts through the oracle's scanning loop (%d threads x %d images, %.1f s), scaled "
                      "to %d classes; local phases + insertion: %d images x (%d phases x %d rot x %d shifts + %d inserts) through the oracle C port, %d "
                      "threads in %d groups with private F / T, %.1f s; reconstructions not included" % (
                          n_cpu, nR, nT, cores, per, t_scan, K, n, shard.nPhase, shard.mLR, shard.mLT, shard.mReco, cores, groups, t_em)}


def bench_config0(args, dev, small=False):
    """BASELINE configs[0] as script/demo_3D.json defines the run (K = 4, C4, mS = 10 000 -> nR = 2 500 scanned rotations, nT = 30,
    mLR 125, mLT 9, mReco 100; a global-search iteration, then a local-search iteration) on 1 000 synthetic 128^3 particles through
    the native driver: what tests/test_next_gpu.py::test_config0_demo3d_as_specified checks for correctness, timed"""
    import torch
    from thunder_amd.native import NativeRefine, STAGES
    from thunder_amd.refine import RefineShard
    N, n, K = (32, 200, 4) if small else (128, 1000, 4)
    scan = dict(nR=200, nT=4, rScan=8, mS=800) if small else dict(nR=2500, nT=30, rScan=12, mS=10000)
    sh = RefineShard(N, n, dev, snr=0.1, K=K, sym="C4", scan=scan, search="global", allocate=False, nblob=24,
                     **(dict(mLR=40, mLT=4, mReco=16) if small else {}))
    sh.balanceClass = 1
    sh.release_generation_state()
    nat = NativeRefine(sh)
    out = {}
    for timed in (False, True):     # one untimed pass of the two iterations (plans, scratch), then the timed one
        nat.reset()
        nat.set_search("global")
        torch.cuda.synchronize()
        nat.stats(reset=True)
        t0 = time.perf_counter()
        nat.iterate(timed)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st1 = nat.stats()
        stages1 = {k: round(st1.stageMs[i], 2) for i, k in enumerate(STAGES)}
        cls = nat.fetch(nat.view().cls, np.int32, (n,))
        nat.set_search("local")
        nat.stats(reset=True)
        nat.iterate(timed)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        st2 = nat.stats()
        stages2 = {k: round(st2.stageMs[i], 2) for i, k in enumerate(STAGES)}
        out = {"metric": "particles/sec per iteration (script/demo_3D.json: K = 4, C4, global then local search; %d x %d^3 particles)" % (n, N),
               "value": 2.0 * n / (t2 - t0), "unit": "particles/s", "n_gpus": 1, "steps": 2, "warmup": 2,
               "ms_per_step": 1e3 * (t2 - t0) / 2.0, "ms_global_search_iteration": 1e3 * (t1 - t0), "ms_local_search_iteration": 1e3 * (t2 - t1),
               "higher_is_better": True, "dtype": "f32", "data": "synthetic", "vs_baseline": None,
               "config": {"workload": "%d synthetic %d^3 particles, K = %d classes, point group C4, nR = %d scanned rotations x nT = %d shifts, "
                                      "mLR 125 / mLT 9 / mReco 100; iteration 1 global search, iteration 2 local search" % (n, N, K, scan["nR"], scan["nT"]),
                          "driver": "thx_refine_iterate"},
               "classes_recovered": float((cls == sh.cls_true).mean()),
               "stages_ms_global_iteration": stages1, "stages_ms_local_iteration": stages2,
               "roofline": None, "cpu_baseline": None,
               "note": "the plumbing case: launch-bound at this size (every stage is milliseconds); correctness of this exact run is "
                       "tests/test_next_gpu.py::test_config0_demo3d_as_specified"}
    nat.close()
    return out


def bench_staged(args, dev):
    """--staged: the DROP-IN path timed -- what an UNCHANGED src/Optimiser.cpp drives through the reference's plug-in surface
    (gpu/interface/Interface.h, here the thx_*_host twins): the local search ONE IMAGE PER CALL from a C++ OpenMP caller loop
    (integration/expectationG_harness.cpp: thx_ExpectLocalP / RTD / PreI3D / M_host as src/Optimiser.cpp:2180-3393 calls them, with and
    without the reference's per-GPU lock), the insertion as InsertFT on host rows (src/Reconstructor.cpp:865-976) and ReconstructG_host on
    host volumes (:1835-2330), on a sample of BASELINE configs[1] (256^3 box).  Host arrays in and out at every call, as the reference's interface has it: this is the
    compatibility form, reported next to the native driver's number (the headline), never as `value` of the metric."""
    import ctypes as C
    import torch
    from thunder_amd import capi, ops
    from thunder_amd.refine import RefineShard
    N, pf = args.box, 2
    P = N * pf
    n = int(args.staged_images)
    threads = int(args.staged_threads)
    sh = RefineShard(N, n, dev, mLR=args.mLR, mLT=args.mLT, nPhase=args.phases, mReco=args.mReco, batch=min(n, 2048),
                     particle_filter=True, allocate=False)
    vol = sh.vols[0].cpu().numpy()                                   # projector volume (host, as Projector::_projectee3D)
    nPxl, nPxlM = sh.nPxl, sh.nPxlM
    iCol, iRow = np.ascontiguousarray(sh.pl["iCol"]), np.ascontiguousarray(sh.pl["iRow"])
    iColM, iRowM = np.ascontiguousarray(sh.plM["iCol"]), np.ascontiguousarray(sh.plM["iRow"])
    ctfM = ops.ctf(sh.attr, sh.pixelSize, sh.iColM, sh.iRowM, N)
    datM = ops.gather_pixels(sh.imgOri, sh.iPxlM, N)
    pos = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(iColM, iRowM))}
    e2m = np.asarray([pos[(int(i), int(j))] for i, j in zip(iCol, iRow)], np.int64)
    datP = np.ascontiguousarray(datM.cpu().numpy()[:, e2m])          # host rows, as Optimiser::_datP / _ctfP / _sigRcpP
    ctfP = np.ascontiguousarray(ctfM.cpu().numpy()[:, e2m])
    sigP = np.full((n, nPxl), np.float32(-0.5 / sh.sigma2), np.float32)
    datMh, ctfMh = np.ascontiguousarray(datM.cpu().numpy()), np.ascontiguousarray(ctfM.cpu().numpy())
    attr_h = np.ascontiguousarray(sh.attr.cpu().numpy())
    quat = np.ascontiguousarray(sh.pf0["r"].cpu().numpy())           # [n][mLR][4] support points of every image
    tran = np.ascontiguousarray(sh.pf0["t"].cpu().numpy())           # [n][mLT][2]
    sh.release_generation_state()
    del datM, ctfM
    nR, nT = sh.mLR, sh.mLT
    # ---- local search: the reference's caller loop (Optimiser::expectationG, src/Optimiser.cpp:2180-3393) restated in C++ over the C ABI
    #      (integration/expectationG_harness.cpp): OpenMP threads over the images, each with its own staging arrays, device slot and
    #      ManagedCalPoint; per image and phase ExpectLocalRTD -> ExpectLocalPreI3D -> ExpectLocalM.  lock = 1 takes the per-GPU lock the
    #      reference holds over the three calls (:2960-3077): ONE image-phase in flight, what an UNCHANGED Optimiser.cpp gets;
    #      lock = 0: the threads run free on their own streams (a caller that drops the lock) ----
    from thunder_amd import build as _b

    class HArgs(C.Structure):
        _fields_ = [(k, C.c_int) for k in ("gpu", "N", "pf", "nPxl", "nImg", "mLR", "mLT", "phases", "threads", "lock")] + \
                   [(k, C.c_void_p) for k in ("volume", "iCol", "iRow", "datP", "ctfP", "sigP", "quat", "tran", "attr", "wR", "wT")] + \
                   [("seconds", C.c_double), ("callSeconds", C.c_double * 4)]
    H = C.CDLL(_b.build_harness())
    H.thx_harness_expectation_local.restype = C.c_int
    wR_out = np.zeros((n, nR), np.float32)

    def run_e(images, threads_, lock):
        a = HArgs(gpu=0, N=N, pf=pf, nPxl=nPxl, nImg=int(images), mLR=nR, mLT=nT, phases=args.phases, threads=threads_, lock=lock,
                  volume=vol.ctypes.data, iCol=iCol.ctypes.data, iRow=iRow.ctypes.data, datP=datP.ctypes.data, ctfP=ctfP.ctypes.data,
                  sigP=sigP.ctypes.data, quat=quat.ctypes.data, tran=tran.ctypes.data, attr=attr_h.ctypes.data, wR=wR_out.ctypes.data, wT=None)
        rc = H.thx_harness_expectation_local(C.byref(a))
        if rc:
            raise RuntimeError("staged local search failed: %s" % capi.load().thx_last_error().decode(errors="replace"))
        calls.append([1e6 * x / (int(images) * args.phases) for x in a.callSeconds])
        return a.seconds
    calls = []
    run_e(min(n, 4 * threads), threads, 0)                           # untimed: first touch
    sweep, call_us = {}, {}
    for lock, ths in ((1, sorted({1, threads})), (0, sorted({threads, 4 * threads, 16 * threads}))):
        for th in ths:
            dt_ = run_e(n, th, lock)
            key = "%s, %d threads" % ("per-GPU lock as in Optimiser.cpp" if lock else "no lock", th)
            sweep[key] = 1e6 * dt_ / (n * args.phases)
            call_us[key] = dict(zip(("ExpectLocalP", "ExpectLocalRTD", "ExpectLocalPreI3D", "ExpectLocalM"), [round(x, 1) for x in calls[-1]]))
    t_e_locked = min(v_ for k_, v_ in sweep.items() if k_.startswith("per-GPU")) * 1e-6 * n * args.phases
    t_e_free = min(v_ for k_, v_ in sweep.items() if k_.startswith("no lock")) * 1e-6 * n * args.phases
    assert np.all(np.isfinite(wR_out)) and wR_out.max() > 0
    # ---- insertion: InsertFT on batches of host rows (quaternions / shifts of mReco draws per image) ----
    rng = np.random.default_rng(3)
    mReco = args.mReco
    iR, iT = rng.integers(0, nR, size=(n, mReco)), rng.integers(0, nT, size=(n, mReco))
    nRq = np.ascontiguousarray(np.take_along_axis(quat, iR[:, :, None], axis=1))
    nTt = np.ascontiguousarray(np.take_along_axis(tran, iT[:, :, None], axis=1))
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    Tc = np.zeros((P, P, P // 2 + 1), np.complex64)                   # the reference's T3D is a complex Volume
    O3, cnt = np.zeros(3), np.zeros(1, np.int32)
    w = np.full(n, np.float32(1.0 / mReco), np.float32)
    offS = np.zeros((n, 2))
    # InsertFT stages the caller's HOST volumes in and out at every call: a fixed cost per call (1.5 GB of F / T each way at 512^3
    # voxels of padded grid) next to a cost per image.  Reconstructor::insertI hands InsertFT every image of the process in ONE call
    # (src/Reconstructor.cpp:865-976), so the two are measured apart -- a call over half the sample and a call over all of it -- and
    # the iteration is priced as fixed + per_image x particles, not as (one small call) x particles / sample.
    def insert_call(b0, b1):
        t0_ = time.perf_counter()
        capi.call("thx_InsertFT_host", F.ctypes.data, Tc.ctypes.data, O3.ctypes.data, cnt.ctypes.data, datMh[b0:b1].ctypes.data,
                  ctfMh[b0:b1].ctypes.data, attr_h[b0:b1].ctypes.data, offS[b0:b1].ctypes.data, w[b0:b1].ctypes.data,
                  nRq[b0:b1].ctypes.data, nTt[b0:b1].ctypes.data, None, None, iColM.ctypes.data, iRowM.ctypes.data,
                  float(sh.pixelSize), 0, pf, nPxlM, mReco, N, P, 1, b1 - b0)
        return time.perf_counter() - t0_
    B = int(args.staged_insert_batch)
    if B:
        t_i = sum(insert_call(b0, min(n, b0 + B)) for b0 in range(0, n, B))
        ins_fixed, ins_per_image = 0.0, t_i / n
    else:
        insert_call(0, min(n, 64))                                   # untimed: first touch of the host volumes
        n_s = max(1, n // 8)
        t_small = min(insert_call(0, n_s), insert_call(0, n_s))      # (two calls each, the faster one: the fixed part is 5 x the difference)
        t_full = min(insert_call(0, n), insert_call(0, n))
        ins_per_image = max(0.0, (t_full - t_small) / (n - n_s))
        ins_fixed = max(0.0, t_full - ins_per_image * n)
        t_i = t_full
    # ---- reconstruction: ReconstructG on host volumes ----
    capi.call("thx_PrepareTF_host", 0, F.ctypes.data, Tc.ctypes.data, P, None, 0, sh.maxRadius, pf)
    fscv = np.ones(N // 2 - 2, np.float32)
    out = np.zeros((N, N, N), np.float32)
    t0 = time.perf_counter()
    capi.call("thx_ReconstructG_host", 0, F.ctypes.data, Tc.ctypes.data, N, N, pf, sh.maxRadius, 1.9, 15.0, fscv.ctypes.data, len(fscv), 1, 0, 1,
              out.ctypes.data)
    t_r = time.perf_counter() - t0
    total = args.particles
    t_ins_total = ins_fixed + ins_per_image * total
    t_iter = total * t_e_locked / n + t_ins_total + 4 * t_r            # an UNCHANGED Optimiser.cpp: its own per-GPU lock around every image-phase
    t_iter_free = total * t_e_free / n + t_ins_total + 4 * t_r
    return {"metric": "particles/sec per refinement iteration through the reference's plug-in surface (staged drop-in path; %d^3 box, %d particles)" % (N, total),
            "value": total / t_iter, "unit": "particles/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
            "vs_baseline": None,
            "value_without_the_callers_lock": total / t_iter_free,
            "config": {"workload": "sample of %d of %d synthetic %d^3 particles: local search %d phases x %d rot x %d shifts ONE IMAGE PER CALL from a C++ "
                                   "OpenMP caller loop over the C ABI (integration/expectationG_harness.cpp = Optimiser::expectationG's loop: "
                                   "thx_ExpectLocalP / RTD / PreI3D / M_host), under the reference's per-GPU lock (`value`) and without it; InsertFT on "
                                   "host rows x %d draws (fixed cost per call + cost per image, one call over all particles as Reconstructor::insertI "
                                   "makes it), ReconstructG on host volumes x 4" % (n, total, N, args.phases, nR, nT, mReco),
                       "driver": "the thx_*_host twins of gpu/interface/Interface.h, as an unchanged Optimiser.cpp calls them"},
            "e_step_us_per_image_phase": 1e6 * t_e_locked / (n * args.phases), "e_step_us_per_image_phase_sweep": {k_: round(v_, 1) for k_, v_ in sweep.items()},
            "e_step_host_us_inside_each_call_per_image_phase": call_us,
            "e_step_images_per_s": n / t_e_locked, "e_step_images_per_s_without_the_lock": n / t_e_free,
            "insert_us_per_image": 1e6 * ins_per_image, "insert_fixed_s_per_call": ins_fixed, "insert_sample_call_s": t_i, "reconstructG_s": t_r,
            "seconds_per_iteration_scaled": t_iter, "roofline": None, "cpu_baseline": None,
            "note": "compatibility form: one launch per image-phase and host staging at every call; the native driver "
                    "(thx_refine_iterate) is the product path and the headline"}


def traffic_in_run(box, timeout_s=900):
    """HBM traffic of the local-search kernel measured NOW, on this box: tools/pmc_traffic.sh (MI355X_MICROARCH.md's recipe: FETCH_SIZE
    and WRITE_SIZE in separate --pmc passes, kernel-trace only, each calibrated on tools/pmc_calib's known-byte-count dispatches in the
    same pass) over one iteration of 20 000 particles of the same workload -- 10 000 images per launch, as the timed run's.  Returns
    (bytes per image-phase, source string) or (None, reason).  Runs in its own process group with a hard timeout: a failure here must
    never cost the line that is already measured."""
    import signal
    import subprocess
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT, THX_PROBE_PARTICLES="20000", THX_PROBE_BOX=str(box), PMC_SKIP_L2="1",
               PMC_CMD="python bench.py --box %d --particles 20000 --steps 1 --warmup 0 --no-cpu-baseline --other-configs off --no-traffic-pass" % box)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        p = subprocess.Popen(["bash", os.path.join(ROOT, "tools", "pmc_traffic.sh")], cwd=ROOT, env=env, stdout=subprocess.DEVNULL,
                             stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)      # (exactly the process group started here)
            p.wait()
            return None, "in-run PMC passes timed out after %d s" % timeout_s
        if rc != 0:
            return None, "tools/pmc_traffic.sh exited with %d" % rc
        j = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_traffic", "pmc_traffic.json")))
        if j.get("box") != box or not j.get("hbm_bytes_per_image_phase"):
            return None, "the PMC passes produced no figure for the local-search kernel"
        return float(j["hbm_bytes_per_image_phase"]), j.get("source")
    except Exception as e:      # noqa: BLE001
        return None, "%s: %s" % (type(e).__name__, str(e)[:200])


def refinement_line(args, dev, rank, world, box, particles, steps, warmup, batch, cpu=True, cpu_particles=0):
    """one refinement configuration through the native driver -> the result dict (rank 0; None on the other ranks)"""
    import torch
    from thunder_amd.native import NativeRefine, make_comms, STAGES
    from thunder_amd.refine import RefineShard, shard_count
    # ---- synthetic particles of this rank (generation only; every other buffer belongs to the native driver) ----
    n_local = shard_count(particles, rank, world)
    shard = RefineShard(box, n_local, dev, rank=rank, world=world, mLR=args.mLR, mLT=args.mLT,
                        nPhase=args.phases, mReco=args.mReco, batch=batch, particle_filter=True, allocate=False,
                        sort_view=not args.unsorted)
    shard.release_generation_state()
    shard.wg_per_cu = args.wg_per_cu   # occupancy argument of the local-search kernel (thx_refine_config.wgPerCU; 2 = the library default)

    # ---- RCCL communicators in native code (thx_comm_*): the unique ids travel through the launcher's process group,
    #      as the reference broadcasts them over MPI (gpu/src/cuthunder.cu:4192-4206) ----
    def share_from(root, uid):
        import torch.distributed as dist
        box_ = [uid]
        dist.broadcast_object_list(box_, src=root)
        return box_[0]
    hemi = wcomm = None
    if world > 1:
        hemi, wcomm = make_comms(rank, world, share_from)
    nat = NativeRefine(shard, hemi, wcomm, norm_correction=not args.no_norm_correction)
    if args.cutoff:
        nat.set_cutoff(*args.cutoff)     # (stays in force over reset and every iteration)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if warmup:
        nat.reset()
        nat.run(warmup)
    nat.reset()
    nat.stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    fsc = nat.run(steps, timed=True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = None
    if rank == 0:
        total_particles = particles * steps
        value = total_particles / dt
        # per-launch averages of the two gather / scatter kernels: HIP events recorded by the native driver on its launch
        # stream around every thx_expect_local_dev / thx_insert_dev call of the timed iterations
        st = nat.stats()
        nPxl, nPxlM = st.nPxl, st.nPxlM
        ins_ms = st.insertMs / max(1, st.insertLaunches)
        ins_n = st.insertImages / max(1, st.insertLaunches)
        exp_ms = st.expectMs / max(1, st.expectLaunches)
        exp_n = st.expectImages / max(1, st.expectLaunches)
        exp_bytes = exp_n * nPxl * (EXPECT_BYTES_PER_PIXEL + EXPECT_BYTES_PER_PIXEL_SAMPLE * shard.mLR)
        t_ins, t_exp = st.insertMs, st.expectMs
        # dominant kernel by total time.  Only the E-step kernel has an HBM roofline that means something: the insertion
        # accumulates in LDS (its HBM traffic is the 56 bytes per record of the sort, a tenth of the 505 MB "algorithmic"
        # bytes per image) and is bound by the LDS atomic rate, reported below as lds_add_frac.
        kname, kms, kbytes = "k_expect_local", exp_ms, exp_bytes
        achieved = kbytes / (kms * 1e-3) / 1e9
        # HBM traffic of that kernel from the committed PMC profile (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes,
        # FETCH_SIZE calibrated on a known-byte-count 64-byte gather in the same passes -- tools/pmc_traffic.sh), scaled
        # to this run's images per launch; null if absent
        traffic, pmc_src, pmc_blob = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        lds_rate64 = None
        if os.path.exists(pmc):
            try:
                import hashlib
                raw = open(pmc, "rb").read()
                pmc_blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()      # = `git hash-object profiles/pmc_traffic.json`
                j = json.loads(raw)
                lds_rate64 = j.get("lds_add_u64_per_s")
                per_box = j.get("per_box", {}).get(str(box))
                if args.cutoff:
                    pass       # (the committed PMC passes ran at Nyquist: no traffic figure for a run below it)
                elif per_box:
                    traffic, pmc_src = per_box["hbm_bytes_per_image_phase"] * exp_n, per_box.get("source")
                elif j.get("box") == box:
                    traffic, pmc_src = j["hbm_bytes_per_image_phase"] * exp_n, j.get("source")
            except Exception:
                traffic = None
        # insertion against its own bound.  The brick-sorted form (k_bin + segment sort + k_acc, thx_insert_sort.hip) issues
        # 24 ds_add_u64 (8 voxels x re, im, T) per RECORD = per (listed pixel, GROUP of draws) -- the insert plan merges the
        # draws of an image that share a rotation; the driver counts the groups its launches processed
        # (thx_insert_groups_total) -- and moves each record through HBM once each way (28 bytes written by k_bin, read by
        # k_acc).  lds_add_frac prices the whole insertion call (plan, k_bin, sort, k_acc) against the chip's measured
        # ds_add_u64 rate (tools/probes/lds_atomic_bench.hip, recorded in profiles/pmc_traffic.json).
        groups_per_image = st.insertGroups / max(1, st.insertImages)
        ins_records_per_s = ins_n * groups_per_image * nPxlM / (ins_ms * 1e-3)
        ins_terms_per_s = ins_records_per_s * 24
        headline = (box, particles) == (256, 100000)
        out = {
            "metric": "particles/sec per refinement iteration (256\u00b3 box, 100k particles); achieved HBM GB/s" if headline
                      else "particles/sec per refinement iteration (%d\u00b3 box, %d particles); achieved HBM GB/s" % (box, particles),
            "value": value, "unit": "particles/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d synthetic %d^3 particles, 3D refinement iteration on %d GPU(s) (local search %d "
                                   "phases x %d rot x %d shifts, normCorrection%s, %d inserts, 2 half-sets, 2x reconstruct per half, FSC, "
                                   "projector refresh)" % (particles, box, world, args.phases, args.mLR,
                                                           args.mLT, " off" if args.no_norm_correction else "", args.mReco),
                       "box": box, "particles": particles, "particles_per_gpu": n_local, "nPxl": nPxl, "nPxlM": nPxlM,
                       "cutoff": dict(zip(("r", "rU", "reco_size", "rScan"), nat.cutoff())),
                       "hbm_in_use_GB": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1),
                       "pf": 2,
                       "search_state": "device particle filter (perturb / resample every phase, Philox-seeded)",
                       "particle_order": "random" if args.unsorted else "by view direction (thx_view_order_host)",
                       "driver": "native C++ iteration driver (thx_refine_iterate) through the C ABI",
                       "parallelism": "particles sharded over %d GPU(s); half-set reduce of the 64-bit fixed-point F/T accumulators in "
                                      "native RCCL (thx_reco_allreduce_acc_class, ncclInt64)" % world},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": kms,
                         "images_per_launch": exp_n, "algorithmic_bytes_per_launch": kbytes, "traffic_source": pmc_src,
                         # PMC passes cannot ride in a timed run: `traffic` is the per-image-phase figure of the committed profile
                         # (separate rocprofv3 --pmc passes over this same workload, tools/pmc_traffic.sh) scaled to this run's launch
                         "traffic_measured_in_run": False, "traffic_profile": "profiles/pmc_traffic.json", "traffic_profile_git_blob": pmc_blob,
                         "note": ("an EFFECTIVE rate (algorithmic bytes, no cache credit, over time): it can exceed the HBM peak -- below Nyquist the "
                                  "rotations of a cloud land within a voxel of each other and share their cells inside L2") if achieved > HBM_PEAK_GBS else None},
            "kernels": {"insertion (k_bin + segment sort + k_acc)": {
                            "avg_call_ms": ins_ms, "images_per_call": ins_n, "total_ms": t_ins,
                            "groups_per_image": groups_per_image, "us_per_image": ins_ms * 1e3 / max(1.0, ins_n),
                            "records_per_s": ins_records_per_s, "lds_adds_u64_per_s": ins_terms_per_s,
                            "lds_add_frac": (ins_terms_per_s / lds_rate64) if lds_rate64 else None,
                            "record_GBps_written_plus_read": ins_records_per_s * 56 / 1e9},
                        "k_expect_local": {"avg_launch_ms": exp_ms, "images_per_launch": exp_n,
                                           "GBps_algorithmic": exp_bytes / (exp_ms * 1e-3) / 1e9, "total_ms": t_exp}},
            # ranks of the library's own communicators (thx_comm_size): what the collectives of this run actually spanned
            "rccl_ranks": {"world": wcomm.size if wcomm is not None else 1, "hemi": hemi.size if hemi is not None else 1,
                           "transport": os.environ.get("THX_COMM_TRANSPORT", "rccl") if world > 1 else None},
            # SURVEY 8(d)'s whole-step algorithmic bytes (E-step: phases x nPxl x (16 + 64 mLR); insertion: nPxlM x mReco x 204 B)
            # over the step time.  This EXCEEDS the HBM peak on purpose and is NOT a bandwidth: the insertion does not move those
            # bytes -- draws that share a rotation are merged (distributivity) and the voxel sums are accumulated in LDS bricks,
            # ~88 MB of real HBM traffic per image against 505 MB algorithmic at 256^3 (F / T parity at these settings:
            # tests/test_fullsize_gpu.py::test_insert_vs_oracle_n256) -- and the E-step's cloud shares cells (traffic 0.44 x algorithmic)
            "step_algorithmic_TBps": particles * (args.phases * nPxl * (EXPECT_BYTES_PER_PIXEL + EXPECT_BYTES_PER_PIXEL_SAMPLE * shard.mLR)
                                                  + nPxlM * args.mReco * INSERT_ALGORITHMIC_BYTES_PER_PIXEL_DRAW) / (dt / steps) / 1e12,
            "step_algorithmic_note": "exceeds the 8 TB/s peak because the insertion merges draws and accumulates in LDS (it never moves its "
                                     "204 B per pixel-draw) and the E-step's rotations share cells; not a bandwidth claim -- roofline.frac is",
            "stages_ms_per_step": {k: round(st.stageMs[i] / steps, 2) for i, k in enumerate(STAGES)},
            "balancing_rounds_per_step": st.balancingRounds / max(1, steps),
            "fsc_half_maps": [round(float(x), 4) for x in fsc[: 8]],
        }
        if cpu and not args.no_cpu_baseline and world == 1 and not args.cutoff:
            args_c = argparse.Namespace(**vars(args))
            args_c.cpu_particles, args_c.particles = cpu_particles, particles
            if box >= 512:
                args_c.cpu_groups = min(args.cpu_groups, 4)    # (a private F / T pair is 6.4 GB of host memory at the 1024^3 grid)
            out["cpu_baseline"] = cpu_baseline(args_c, shard, nat, out["balancing_rounds_per_step"])
        else:
            out["cpu_baseline"] = None
    nat.close()
    if hemi is not None:
        hemi.close()
    if wcomm is not None:
        wcomm.close()
    del nat, shard
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--box", type=int, default=256)
    ap.add_argument("--particles", type=int, default=100000,
                    help="TOTAL particles of the job, sharded over the GPUs (100000 = BASELINE metric / configs[2]; "
                         "10000 = configs[1])")
    ap.add_argument("--mLR", type=int, default=125)
    ap.add_argument("--mLT", type=int, default=9)
    ap.add_argument("--phases", type=int, default=3)
    ap.add_argument("--mReco", type=int, default=100)
    ap.add_argument("--batch", type=int, default=10240, help="max images per kernel launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-particles", type=int, default=0, help="CPU baseline sample (default: 2048 particles at the headline workload -- SURVEY 8d asks for >= 2000 --, two per core otherwise)")
    ap.add_argument("--cpu-no-reconstruct", action="store_true")
    ap.add_argument("--cpu-groups", type=int, default=16, help="CPU baseline: thread groups with private F / T (MPI ranks of the reference)")
    ap.add_argument("--cpu-shared", action="store_true", help="CPU baseline: also time the single-team form (one shared F / T)")
    ap.add_argument("--cutoff", type=int, nargs=2, metavar=("R", "RU"), default=None,
                    help="frequency cut-offs of the timed iterations (thx_refine_set_cutoff): r = Optimiser::_r (E-step list, projector radius), "
                         "rU = Model::_rU (M-step list, Reconstructor::_maxRadius, reconstruction grid size min(N, (rU + 2) * 2)); default: Nyquist, "
                         "N / 2 - 2 for both -- what the metric is quoted on")
    ap.add_argument("--no-traffic-pass", action="store_true",
                    help="skip the in-run PMC passes (headline workload on one GPU: after the timed iterations, tools/pmc_traffic.sh counts "
                         "FETCH_SIZE / WRITE_SIZE of a 20 000-particle iteration of the same workload in separate rocprofv3 passes, calibrated in the "
                         "same passes, and `roofline.traffic` becomes that run's figure; otherwise the committed profile's)")
    ap.add_argument("--no-norm-correction", action="store_true",
                    help="leave Optimiser::normCorrection (OPTIMISER_NORM_CORRECTION, on in the reference's Config.h) out of the iteration")
    ap.add_argument("--unsorted", action="store_true",
                    help="keep the particles in random order (default: stored by view direction, thx_view_order_host)")
    ap.add_argument("--classification", action="store_true",
                    help="one whole K = 4 classification iteration of configs[3] on one GPU's share of the images instead "
                         "(scan, class selection, local phases, sigma update, multi-reference insertion, reconstructions, FSC, refresh)")
    ap.add_argument("--wg-per-cu", type=int, default=2, help="occupancy argument of the local-search kernel (workgroups of 4 waves per CU; 0 = unlimited; default 2, DESIGN 4.1)")
    ap.add_argument("--scan-only", action="store_true", help="with --classification: the global scanning stage on its own")
    ap.add_argument("--scan-images", type=int, default=0, help="images of the classification bench (default 6250 = 50 000 / 8 GPUs; 1024 with --scan-only)")
    ap.add_argument("--scan-cpu-images-per-core", type=int, default=2)
    ap.add_argument("--config0-small", action="store_true", help="--config0 on toy sizes (tests)")
    ap.add_argument("--config0", action="store_true", help="BASELINE configs[0] on its own: script/demo_3D.json's run (K = 4, C4, global then local search) on 1 000 x 128^3 particles")
    ap.add_argument("--staged", action="store_true", help="time the drop-in path instead: the per-image / per-stage thx_*_host entry points of the "
                    "reference's plug-in surface on a sample of configs[1] (see bench_staged)")
    ap.add_argument("--staged-images", type=int, default=2048)
    ap.add_argument("--staged-threads", type=int, default=8)
    ap.add_argument("--staged-insert-batch", type=int, default=0, help="images per InsertFT call (0 = all of the sample in ONE call, as "
                    "Reconstructor::insertI hands InsertFT every image of the process, src/Reconstructor.cpp:865-976)")
    ap.add_argument("--other-configs", choices=("auto", "on", "off"), default="auto",
                    help="after the headline line, also run BASELINE configs[1] (10 000 x 256^3), configs[3] (one GPU's share of the K = 4 "
                         "classification) and configs[4] (20 000 x 512^3) for 2 + 1 iterations each and report them under `other_configs` of the "
                         "same JSON line (auto: when the command times the headline workload on one GPU)")
    args = ap.parse_args()

    # --gpus N is the number of ranks of the job.  Under a launcher (WORLD_SIZE set) the two must agree; without one, N > 1
    # re-executes this command under torch.distributed.run with N ranks on a free port -- `python bench.py --gpus 8` IS the 8-rank
    # job, never a one-rank run that prints "n_gpus": 1.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): the two must agree" % (args.gpus, world))
    if world > 1 and (args.config0 or args.staged or args.classification):
        raise SystemExit("bench.py: --config0 / --staged / --classification are one-GPU lines; the N-rank job is the refinement line")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # THX_BENCH_ONE_DEVICE=1 (tests only, never a measurement): every rank on cuda:0 -- the launcher's process group over gloo and,
    # with THX_COMM_TRANSPORT=shm, the native communicators over the library's shared-memory transport -- so that the N > 1 code path
    # of this file (sharding, id exchange, communicators, max-over-ranks timing) runs on a 1-GPU box
    one_dev = os.environ.get("THX_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
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
    capi.load()

    if args.config0:
        print(json.dumps(bench_config0(args, dev, small=args.config0_small)))
        return
    if args.staged:
        if args.particles == 100000:
            args.particles = 10000      # configs[1]
        print(json.dumps(bench_staged(args, dev)))
        return
    if args.classification:
        if not args.scan_images:
            args.scan_images = 1024 if args.scan_only else 6250
        if args.scan_only:
            bench_global_scan(args, dev)
        else:
            print(json.dumps(bench_classification_iteration(args, dev)))
        return
    headline = (args.box, args.particles) == (256, 100000)
    cpu_n = args.cpu_particles or (2048 if headline else 0)
    out = refinement_line(args, dev, rank, world, args.box, args.particles, args.steps, args.warmup, args.batch, cpu=True, cpu_particles=cpu_n)
    others = args.other_configs == "on" or (args.other_configs == "auto" and headline and world == 1)
    if others and world == 1:
        # the other BASELINE configs, driver-visible: 2 timed iterations after 1 warm-up each, every line with its own roofline and
        # cpu_baseline (bounded samples).  Each runs in its OWN process, after this one has given its device memory back: a fresh
        # allocator state per configuration (round 4's in-process run timed the 512^3 E-step launch 5 % slower than the standalone
        # command: 34 GB of cell-packed volume allocated into a heap the earlier configurations had fragmented), and a failure there
        # cannot cost the headline line, which is already measured.
        import gc
        import subprocess
        from thunder_amd.capi import stream_ptr
        gc.collect()
        torch.cuda.synchronize()
        capi.call("thx_release_stream", stream_ptr())
        torch.cuda.empty_cache()
        oc = {}
        t0 = time.perf_counter()
        small = os.environ.get("THX_BENCH_SMALL_OTHERS") == "1"     # (tests/test_next_gpu.py: the same code path on toy sizes)
        if headline and not small and not args.no_traffic_pass:
            # roofline.traffic of THIS run (round-5 review: the committed profile's figure could not be contradicted by the driver's run)
            per_phase, src = traffic_in_run(args.box)
            rf = out["roofline"]
            if per_phase is not None:
                rf["traffic_committed_profile"] = rf["traffic"]
                rf["traffic"] = per_phase * rf["images_per_launch"]
                rf["traffic_measured_in_run"] = True
                rf["traffic_source"] = "measured in this run, after the timed iterations: " + str(src)
                rf["traffic_over_algorithmic"] = rf["traffic"] / rf["algorithmic_bytes_per_launch"]
            else:
                rf["traffic_in_run_failed"] = src
            out["traffic_pass_wall_s"] = round(time.perf_counter() - t0, 1)
        b1, n1, b3, n3, b4, n4 = (32, 300, 64, 96, 64, 200) if small else (256, 10000, 256, 6250, 512, 20000)
        common = ["--mLR", str(args.mLR), "--mLT", str(args.mLT), "--phases", str(args.phases), "--mReco", str(args.mReco), "--batch", str(args.batch),
                  "--wg-per-cu", str(args.wg_per_cu), "--cpu-groups", str(args.cpu_groups), "--other-configs", "off"]
        if args.no_cpu_baseline:
            common.append("--no-cpu-baseline")
        runs = (("configs[0] script/demo_3D.json as specified", ["--config0"] + (["--config0-small"] if small else [])),
                ("configs[1] %d x %d^3 refinement" % (n1, b1), ["--box", str(b1), "--particles", str(n1), "--steps", "2", "--warmup", "1", "--cpu-particles", "256"]),
                ("configs[3] K=4 classification, one GPU's share (%d of 50k images)" % n3,
                 ["--classification", "--box", str(b3), "--scan-images", str(n3), "--steps", "2", "--warmup", "1"]),
                ("configs[4] %d x %d^3 refinement" % (n4, b4), ["--box", str(b4), "--particles", str(n4), "--steps", "2", "--warmup", "1", "--cpu-particles", "64"]))
        # an iteration BELOW Nyquist, as a caller following Model::updateR runs most of them: r = rU = 48 of 126 at 256^3 -- E-step on the
        # 3 600-pixel list, insertion / reduce / gridding loop on the resized 200^3 grid (Reconstructor::resizeSpace), 48 FSC shells
        rc = max(6, b1 // 2 - 2 - (b1 * 78) // 256)
        runs += (("configs[1] %d x %d^3 refinement at cut-offs r = rU = %d (resized reconstruction grid)" % (n1, b1, rc),
                  ["--box", str(b1), "--particles", str(n1), "--steps", "2", "--warmup", "1", "--cutoff", str(rc), str(rc)]),)
        if not small:
            runs += (("configs[1] through the reference's plug-in surface (staged drop-in path)", ["--staged"]),)
        for name, extra in runs:
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + common + extra, capture_output=True, text=True, timeout=3600)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not lines:
                    oc[name] = {"error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
                else:
                    oc[name] = json.loads(lines[-1])
            except Exception as e:      # noqa: BLE001
                oc[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        out["other_configs"] = oc
        out["other_configs_wall_s"] = round(time.perf_counter() - t0, 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
