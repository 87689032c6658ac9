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
sequenced by the native driver thx_classify_iterate (thunder_amd/csrc/thx_classify.hip; `--python-sequencing` for the A/B form).
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
INSERT_BYTES_PER_PIXEL_SAMPLE = 204  # 12 B in + F 8x8 Bx(R+W) + T 8x4 Bx(R+W)
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
    t0 = time.perf_counter()
    O.baseline_block(vol, P, shard.pf, N, pl, dat, ctf, sig, rot, tran, recoRot, recoTran, F, T, groups=groups)
    Fsum, Tsum = F.sum(axis=0), T.sum(axis=0)      # the hemisphere all-reduce of the groups' volumes
    t_em = time.perf_counter() - t0
    em_rate = n / t_em
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
            "em_particles_per_s": em_rate, "reconstruct": reco,
            "thread_layout": "%d groups x %d threads, private F / T per group, summed at the end" % (groups, max(1, cores // groups)),
            "single_team_shared_FT": shared,
            "note": "a reported baseline, not the target: GPU / CPU says nothing about kernel quality, roofline.frac does",
            "sample": "E-step + insertion: %d particles x (%d phases x %d rot x %d shifts + %d inserts) through the oracle C "
                      "port, OpenMP over images on %d threads in %d groups with private F / T, %.1f s; reconstruct leg: oracle gridding reconstruction on "
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


def bench_classification_iteration(args, dev):
    """--classification: one whole iteration of BASELINE configs[3] (3-D classification, K = 4 references) on ONE GPU's share
    of the images, sequenced over the batched `*_dev` entry points the way Optimiser::expectation does for a classification
    (src/Optimiser.cpp:631-1660): global scan of every image against K classes x 10 000 rotations x 30 shifts at r = 24
    (:756-894) -> class of every image (keepHalfHeightPeak(PAR_C) / resample / rand, :925-952: k_pf_class_select) -> support
    points from the selected class's scan posterior (keepHalfHeightPeak / resample(mLR, PAR_R) / resample(mLT, PAR_T) / calVari,
    :953-1008: k_pf_scan_support) -> 3 local particle-filter phases
    against the assigned reference (volIdx; k_pf_perturb / k_expect_local<9, packed> / k_pf_update) -> mReco draws per image ->
    multi-reference insertion (cls per draw, K pairs of F / T in one session: k_bin / sort / k_acc) -> normalise + 2
    reconstructions per class (MAP off / on, Reconstructor::reconstruct).  Left out of the timed region: the sigma update and
    the re-centring of the images (timed in the refinement bench), half-set splitting (one GPU's share is one half here).
    `value` = images per second through the whole iteration; the per-stage times and the roofline of each stage's dominant
    kernel ride along."""
    import torch
    from thunder_amd import capi, ops, synth
    from thunder_amd.refine import pixel_list
    N, K, nR, nT, rScan = args.box, 4, 10000, 30, 24
    mLR, mLT, mReco, nPhase = args.mLR, args.mLT, args.mReco, args.phases
    nImg = args.scan_images
    P, pf, rU = 2 * N, 2, N // 2 - 2
    rng = np.random.default_rng(4)
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    plS, plE, plM = pixel_list(N, rScan, 2), pixel_list(N, rU, 2), pixel_list(N, rU, 0)
    posM = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(plM["iCol"], plM["iRow"]))}
    e2m = T_(np.asarray([posM[(int(i), int(j))] for i, j in zip(plE["iCol"], plE["iRow"])], np.int64))
    s2m = T_(np.asarray([posM[(int(i), int(j))] for i, j in zip(plS["iCol"], plS["iRow"])], np.int64))
    iColS, iRowS, iColE, iRowE, iColM, iRowM = (T_(plS["iCol"]), T_(plS["iRow"]), T_(plE["iCol"]), T_(plE["iRow"]),
                                                T_(plM["iCol"]), T_(plM["iRow"]))
    nPxlS, nPxlE, nPxlM = plS["nPxl"], plE["nPxl"], plM["nPxl"]
    plan = ops.RecoPlan(N, N, pf)
    refs = torch.stack([T_(synth.blob_map(N, seed=300 + k, nblob=20)) for k in range(K)]).contiguous()
    vols = torch.stack([plan.set_projectee(refs[k]) for k in range(K)]).contiguous()
    cells = ops.pack_projector(vols, P)
    quat = synth.random_quats(nR, rng)
    mats = ops.rotmat(T_(quat))
    quatD = T_(quat)
    shifts = np.ascontiguousarray(rng.normal(0, 3.0, size=(nT, 2)))
    shiftsD = T_(shifts)
    traS = ops.translate(shiftsD, iColS, iRowS, N)
    traM = ops.translate(shiftsD, iColM, iRowM, N)
    attr = T_(synth.ctf_params(nImg, rng))
    ctfM = ops.ctf(attr, 1.32, iColM, iRowM, N)
    # images: a slice of a random class at a random scanned rotation / shift, plus noise (rL = 0 list; the E-step and scan rows
    # are its sub-lists)
    cls_true, r_true, t_true = rng.integers(0, K, nImg), rng.integers(0, nR, nImg), rng.integers(0, nT, nImg)
    if not args.unsorted:
        # stored by class and view direction of the previous iteration's poses (here: the generating ones), as the refinement
        # bench stores its particles by view (thx_view_order_host): images next to each other in a launch gather from the
        # same reference along nearly the same plane
        from thunder_amd.refine import view_order
        perm = view_order(quat[r_true])
        perm = perm[np.argsort(cls_true[perm], kind="stable")]
        cls_true, r_true, t_true = cls_true[perm], r_true[perm], t_true[perm]
    datM = torch.empty((nImg, nPxlM), dtype=torch.complex64, device=dev)
    for k in range(K):
        sel = np.nonzero(cls_true == k)[0]
        for c0 in range(0, len(sel), 1024):
            ss = sel[c0:c0 + 1024]
            sl = ops.project(vols[k], mats[T_(r_true[ss])].contiguous(), iColM, iRowM, pf)
            st_ = T_(ss)
            datM[st_] = sl * traM[T_(t_true[ss])] * ctfM[st_]
    sd = 3.0 * float(datM.abs().pow(2).mean().sqrt())
    g = torch.Generator(device=dev); g.manual_seed(5)
    for c0 in range(0, nImg, 2048):
        c1 = min(nImg, c0 + 2048)
        datM[c0:c1] += torch.view_as_complex(torch.randn((c1 - c0, nPxlM, 2), generator=g, device=dev)) * (sd / np.sqrt(2))
    datE, ctfE = datM[:, e2m].contiguous(), ctfM[:, e2m].contiguous()
    datS, ctfS = datM[:, s2m].contiguous(), ctfM[:, s2m].contiguous()
    sig = -0.5 / (sd * sd / 2)
    sigM = torch.full((nImg, nPxlM), sig, dtype=torch.float32, device=dev)
    sigE = torch.full((nImg, nPxlE), sig, dtype=torch.float32, device=dev)
    sigS = torch.full((nImg, nPxlS), sig, dtype=torch.float32, device=dev)
    pR = torch.full((nImg, nR), 1.0 / nR, dtype=torch.float64, device=dev)
    pT = torch.full((nImg, nT), 1.0 / nT, dtype=torch.float64, device=dev)
    native = not args.python_sequencing
    py = (not native) or args.check_native
    wC = torch.zeros((nImg, K), dtype=torch.float32, device=dev)
    wR = torch.zeros((K, nImg, nR) if py else (1,), dtype=torch.float32, device=dev)
    wT = torch.zeros((K, nImg, nT) if py else (1,), dtype=torch.float32, device=dev)
    base = torch.empty((nImg,), dtype=torch.float32, device=dev)
    wsG = torch.empty(capi.load().thx_expect_global_workspace(nImg, nR, nT), dtype=torch.uint8, device=dev)
    batch = min(nImg, args.batch)
    wsL = torch.empty(capi.load().thx_expect_local_workspace(batch, mLR, mLT, 1), dtype=torch.uint8, device=dev)
    rotP = torch.empty((nR, nPxlS), dtype=torch.complex64, device=dev)
    F = torch.zeros((K, P, P, P // 2 + 1) if py else (1,), dtype=torch.complex64, device=dev)
    Tt = torch.zeros((K, P, P, P // 2 + 1) if py else (1,), dtype=torch.float32, device=dev)
    w = torch.full((nImg,), 1.0 / mReco, dtype=torch.float32, device=dev)
    ar = torch.arange(nImg, device=dev)
    seed, state = 20240607, {"call": 0}
    ev = {"scan": [], "local": [], "insert": []}
    stage_ms = {}
    maps = {}

    class Stage:
        def __init__(self, name, timed): self.name, self.timed = name, timed
        def __enter__(self):
            if self.timed:
                torch.cuda.synchronize(); self.t0 = time.perf_counter()
        def __exit__(self, *a):
            if self.timed:
                torch.cuda.synchronize(); stage_ms[self.name] = stage_ms.get(self.name, 0.0) + (time.perf_counter() - self.t0) * 1e3

    def timed_call(key, timed, fn, n):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        r = fn()
        if timed:
            e1.record(); ev[key].append((e0, e1, n))
        return r

    def iteration(timed):
        with Stage("scan", timed):
            wC.zero_(); wR.zero_(); wT.zero_(); base.fill_(float("nan"))
            for k in range(K):
                ops.project(vols[k], mats, iColS, iRowS, pf, out=rotP)
                timed_call("scan", timed, lambda: ops.expect_global(rotP, traS, datS, ctfS, sigS, pR, pT, wC, wR, wT, base, k, K, workspace=wsG), nImg)
        with Stage("class_select_and_support_points", timed):
            state["call"] += 1
            cls = ops.pf_class_select(wC, seed, state["call"])
            state["call"] += 1
            # Particle::keepHalfHeightPeak(PAR_R) / resample(mLR, PAR_R) / resample(mLT, PAR_T) / calVari on the scan posterior of the
            # image's class (src/Optimiser.cpp:953-1008)
            # with the scanning phase's minimum spread (OPTIMISER_SCAN_SET_MIN_STD_WITH_PERTURB, :1032-1079): scanMinStdR = nR^(-1/3),
            # scanMinStdT = 1 / Qinv(INIT_OUTSIDE_CONFIDENCE_AREA, 2) / sqrt(transSearchFactor pi), over perturbFactorSGlobal = 0.5
            minK = (nR ** (-1.0 / 3) / 0.5) ** 2
            minS = 1.0 / (-2.0 * np.log(INIT_OUTSIDE_CONFIDENCE_AREA)) / np.sqrt(TRANS_SEARCH_FACTOR * np.pi) / 0.5
            st = ops.pf_scan_support(quatD, shiftsD, wR, wT, cls, mLR, mLT, 1e-3, seed, state["call"], minK, minS)
        with Stage("local_phases", timed):
            for p_ in range(nPhase):
                for b0 in range(0, nImg, batch):
                    b1 = min(nImg, b0 + batch); sl = slice(b0, b1)
                    state["call"] += 1
                    f = 2.0 if p_ == 0 else 0.5
                    ops.pf_perturb(st["r"][sl], st["t"][sl], st["wR"][sl], st["wT"][sl], st["k"][sl], st["s"][sl], f, f, 2.0, 0.05, seed, state["call"])
                    rotB = ops.rotmat(st["r"][sl].reshape(-1, 4)).reshape(b1 - b0, mLR, 9)
                    r = timed_call("local", timed, lambda: ops.expect_local(cells, P, pf, N, iColE, iRowE, datE[sl], ctfE[sl], sigE[sl], rotB, st["t"][sl],
                                                                            volIdx=cls[sl], pR=st["wR"][sl], pT=st["wT"][sl], workspace=wsL, packed=True, wg_per_cu=args.wg_per_cu), b1 - b0)
                    state["call"] += 1
                    ops.pf_update(st["r"][sl], st["t"][sl], st["wR"][sl], st["wT"][sl], r.wR, r.wT, st["k"][sl], st["s"][sl], st["topR"][sl], st["topT"][sl],
                                  1e-3, seed, state["call"])
        with Stage("insertion", timed):
            F.zero_(); Tt.zero_()
            for b0 in range(0, nImg, batch):
                b1 = min(nImg, b0 + batch); sl = slice(b0, b1)
                state["call"] += 1
                rot, tran = ops.draw_reco(st["r"][sl], st["t"][sl], mReco, seed, state["call"], b0)
                clsD = cls[sl][:, None].expand(-1, mReco).contiguous()
                timed_call("insert", timed, lambda: ops.insert(F, Tt, P, datM[sl], ctfM[sl], w[sl], rot, tran, iColM, iRowM, pf, N, cls=clsD, nK=K), b1 - b0)
        with Stage("reconstruct", timed):
            rounds = 0
            for k in range(K):
                if float(Tt[k, 0, 0, 0]) <= 0:
                    continue
                ops.normalise_TF(F[k], Tt[k], P)
                for MAP in (False, True):   # (reconstruct works on the class's ONE T in place, as the reference does)
                    maps[(k, MAP)] = plan.reconstruct(F[k], Tt[k], rU, FSC=np.ones(rU, np.float32) if MAP else None, joinHalf=False, MAP=MAP, gridCorr=True)
                    rounds += int(plan.last_iters)
        return cls, st, rounds

    ref_run = None
    if native and args.check_native:
        # A/B: the same iteration sequenced in Python first (Philox call counter from 0, as the fresh native handle's)
        cls_p, st_p, rounds_p = iteration(False)
        torch.cuda.synchronize()
        ref_run = dict(cls=cls_p.cpu().numpy(), r=st_p["r"].cpu().numpy(), t=st_p["t"].cpu().numpy(), topR=st_p["topR"].cpu().numpy(),
                       F=F.cpu().numpy(), T=Tt.cpu().numpy(), maps={k: m.cpu().numpy() for k, m in maps.items()}, rounds=rounds_p)
    if native:
        # the iteration in native code (thx_classify_iterate, thunder_amd/csrc/thx_classify.hip): Python hands over the rows, the
        # scanned grid and the references once and calls the driver once per iteration
        from thunder_amd.native import CLASSIFY_STAGES, NativeClassify
        F = Tt = wR = wT = None
        nat = NativeClassify(N, K, nImg, nR, nT, rScan, rL=2, pf=pf, mLR=mLR, mLT=mLT, nPhase=nPhase, mReco=mReco, batch=batch, pixel_order=0,
                             wg_per_cu=args.wg_per_cu, seed=seed)
        nat.set_grid(quatD, shiftsD); nat.set_particles(datM, ctfM, sigM, w); nat.set_references(refs)
        for _ in range(args.warmup):
            nat.iterate(False)
        torch.cuda.synchronize()
        nat.stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            nat.iterate(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ns, v = nat.stats(), nat.view()
        cls = T_(nat.fetch(v.cls, np.int32, (nImg,)))
        st = dict(r=T_(nat.fetch(v.r, np.float64, (nImg, mLR, 4))), t=T_(nat.fetch(v.t, np.float64, (nImg, mLT, 2))),
                  topR=T_(nat.fetch(v.topR, np.float64, (nImg, 4))))
        rounds = int(ns.balancingRounds) // max(1, args.steps)
        for i, name in enumerate(CLASSIFY_STAGES):
            stage_ms[name] = float(ns.stageMs[i])
        rotP = torch.empty((nR, nPxlS), dtype=torch.complex64, device=dev)   # (the CPU baseline's sample of slices)
        check = None
        if ref_run is not None:
            volN, mapN = P * P * (P // 2 + 1), N * N * N
            Fn = nat.fetch(v.F, np.complex64, (K, P, P, P // 2 + 1)); Tn = nat.fetch(v.T, np.float32, (K, P, P, P // 2 + 1))
            m0 = nat.fetch(v.maps, np.float32, (K, N, N, N)); m1 = nat.fetch(v.mapsMAP, np.float32, (K, N, N, N))
            dm = 0.0
            for (k, MAP), m in ref_run["maps"].items():
                dm = max(dm, float(np.abs((m1 if MAP else m0)[k] - m).max()))
            check = {"cls_equal": bool((cls.cpu().numpy() == ref_run["cls"]).all()),
                     "r_max_abs_diff": float(np.abs(st["r"].cpu().numpy() - ref_run["r"]).max()),
                     "t_max_abs_diff": float(np.abs(st["t"].cpu().numpy() - ref_run["t"]).max()),
                     "topR_max_abs_diff": float(np.abs(st["topR"].cpu().numpy() - ref_run["topR"]).max()),
                     "F_max_abs_diff": float(np.abs(Fn - ref_run["F"]).max()), "T_max_abs_diff": float(np.abs(Tn - ref_run["T"]).max()),
                     "F_max_abs": float(np.abs(ref_run["F"]).max()), "maps_max_abs_diff": dm,
                     "rounds_native": int(ns.balancingRounds), "rounds_python": int(ref_run["rounds"]),
                     "single_batch": bool(nImg <= batch)}
    else:
        for _ in range(args.warmup):
            iteration(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cls, st, rounds = iteration(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ok = float((cls.cpu().numpy() == cls_true).mean())
    # poses after the local phases against the generating ones
    d = np.abs((st["topR"].cpu().numpy() * quat[r_true]).sum(1)).clip(0, 1)
    ang = np.degrees(2 * np.arccos(d))
    ms = lambda key: float(np.mean([a.elapsed_time(b) for a, b, _ in ev[key]]))
    n_of = lambda key: float(np.mean([n for _, _, n in ev[key]]))
    if native:
        scan_ms, loc_ms, ins_ms = ns.scanMs / max(1, ns.scanLaunches), ns.localMs / max(1, ns.localLaunches), ns.insertMs / max(1, ns.insertLaunches)
        n_of = lambda key: {"local": ns.localImages / max(1, ns.localLaunches), "insert": ns.insertImages / max(1, ns.insertLaunches), "scan": float(nImg)}[key]
    else:
        scan_ms, loc_ms, ins_ms = ms("scan"), ms("local"), ms("insert")
    flops = 4.0 * nImg * nR * nT * nPxlS
    loc_bytes = n_of("local") * nPxlE * (EXPECT_BYTES_PER_PIXEL + EXPECT_BYTES_PER_PIXEL_SAMPLE * mLR)
    stages = {k: round(v / args.steps, 2) for k, v in stage_ms.items()}
    dominant = max(("scan", "local_phases"), key=lambda k: stages.get(k, 0.0))
    roof_scan = {"bound": "mfma", "kernel": "thx_expect_global_dev = k_scan_tables + k_scan_gemm (f32 MFMA) + fold (one class: %d images x %d rot x %d shifts)" % (nImg, nR, nT),
                 "achieved": flops / (scan_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / (scan_ms * 1e-3) / 1e12 / 157.3,
                 "traffic": None, "avg_launch_ms": scan_ms,
                 "note": "exact-f32 contraction on v_mfma_f32_32x32x2_f32 (bit-equal to the fmaf chain); peak = f32 MFMA = f32 vector rate"}
    roof_local = {"bound": "hbm", "kernel": "k_expect_local<9, packed> with volIdx (K cell-packed references)", "achieved": loc_bytes / (loc_ms * 1e-3) / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": loc_bytes / (loc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                  "avg_launch_ms": loc_ms, "images_per_launch": n_of("local")}
    out = {"metric": "images/sec through one 3-D classification iteration (K = 4, %d^3 box): global scan + class selection + %d local phases + "
                     "multi-reference insertion + 2 reconstructions per class" % (N, nPhase),
           "value": nImg * args.steps / dt, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[3] on one GPU's share: %d synthetic %d^3 images, K = %d classes; scan %d rotations x %d shifts at r = %d "
                                  "(%d pixels); local search %d phases x %d rot x %d shifts on %d pixels; %d inserts per image into %d F / T pairs; "
                                  "%d reconstructions" % (nImg, N, K, nR, nT, rScan, nPxlS, nPhase, mLR, mLT, nPxlE, mReco, K, 2 * K),
                      "classes_recovered": ok, "median_pose_error_deg": float(np.median(ang)), "particle_order": "random" if args.unsorted else "by class, then view direction",
                      "sequenced_by": "thx_classify_iterate (native C++ driver, thx_classify.hip)" if native else "Python over the *_dev entry points (--python-sequencing)"},
           "roofline": roof_scan if dominant == "scan" else roof_local,
           "rooflines": {"scan": roof_scan, "local_phases": roof_local},
           "kernels": {"insertion (k_bin + segment sort + k_acc), %d classes in one session" % K: {"avg_call_ms": ins_ms, "images_per_call": n_of("insert"),
                                                                                             "us_per_image": ins_ms * 1e3 / max(1.0, n_of("insert"))}},
           "stages_ms_per_step": stages, "balancing_rounds_per_step": rounds, "cpu_baseline": None}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_classification(args, dev, dict(K=K, nR=nR, nT=nT, N=N, P=P, pf=pf, nImg=nImg, mLR=mLR, mLT=mLT, mReco=mReco, nPhase=nPhase,
                                                                      vols=vols, mats=mats, iColS=iColS, iRowS=iRowS, rotP=rotP, traS=traS, datS=datS, ctfS=ctfS, sigS=sigS,
                                                                      plE=plE, datE=datE, ctfE=ctfE, sigE=sigE, st=st, cls=cls))
    if native and check is not None:
        out["native_vs_python"] = check
    print(json.dumps(out))
    plan.close()


def cpu_baseline_classification(args, dev, c):
    """oracle (`kind: port`) on bounded samples of the classification iteration, all host cores: the scanning loop (every core a block
    of images through ONE class, scaled to K) and the local phases + insertion (oracle.baseline_block on the filter's support points
    of the sample, one class volume: the arithmetic does not depend on which class an image is in).  The reconstructions are left
    out (the refinement bench times that leg); value = images / (scan seconds + local-and-insertion seconds) for the job's images."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from thunder_amd import ops
    K, nR, nT, nImg = c["K"], c["nR"], c["nT"], c["nImg"]
    cores = os.cpu_count() or 1
    per = max(1, int(args.scan_cpu_images_per_core))
    n_cpu = min(nImg, per * cores)
    ops.project(c["vols"][0], c["mats"], c["iColS"], c["iRowS"], c["pf"], out=c["rotP"])
    rotP_h, traP_h = c["rotP"].cpu().numpy(), c["traS"].cpu().numpy()
    dat_h, ctf_h, sig_h = c["datS"][:n_cpu].cpu().numpy(), c["ctfS"][:n_cpu].cpu().numpy(), c["sigS"][:n_cpu].cpu().numpy()

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
    P, N, pf, plE = c["P"], c["N"], c["pf"], c["plE"]
    pl = dict(iCol=plE["iCol"], iRow=plE["iRow"], iColPad=plE["iColPad"], iRowPad=plE["iRowPad"], nPxl=plE["nPxl"])
    vol = c["vols"][0].cpu().numpy()
    dat, ctf, sig = c["datE"][:n].cpu().numpy(), c["ctfE"][:n].cpu().numpy(), c["sigE"][:n].cpu().numpy()
    quat, t1 = c["st"]["r"][:n].cpu().numpy(), c["st"]["t"][:n].cpu().numpy()
    r1 = np.stack([[O.rotate3D(q) for q in qs] for qs in quat])
    rot = np.ascontiguousarray(np.stack([r1] * c["nPhase"], axis=1))
    tran = np.ascontiguousarray(np.stack([t1] * c["nPhase"], axis=1))
    rng = np.random.default_rng(1)
    iR, iT = rng.integers(0, c["mLR"], size=(n, c["mReco"])), rng.integers(0, c["mLT"], size=(n, c["mReco"]))
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
                          n_cpu, nR, nT, cores, per, t_scan, K, n, c["nPhase"], c["mLR"], c["mLT"], c["mReco"], cores, groups, t_em)}


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
    ap.add_argument("--cpu-particles", type=int, default=0, help="CPU baseline sample (default: two particles per core)")
    ap.add_argument("--cpu-no-reconstruct", action="store_true")
    ap.add_argument("--cpu-groups", type=int, default=16, help="CPU baseline: thread groups with private F / T (MPI ranks of the reference)")
    ap.add_argument("--cpu-shared", action="store_true", help="CPU baseline: also time the single-team form (one shared F / T)")
    ap.add_argument("--no-norm-correction", action="store_true",
                    help="leave Optimiser::normCorrection (OPTIMISER_NORM_CORRECTION, on in the reference's Config.h) out of the iteration")
    ap.add_argument("--unsorted", action="store_true",
                    help="keep the particles in random order (default: stored by view direction, thx_view_order_host)")
    ap.add_argument("--classification", action="store_true",
                    help="one whole K = 4 classification iteration of configs[3] on one GPU's share of the images instead "
                         "(scan, class selection, local phases, multi-reference insertion, reconstructions)")
    ap.add_argument("--wg-per-cu", type=int, default=2, help="occupancy argument of the local-search kernel (workgroups of 4 waves per CU; 0 = unlimited; default 2, DESIGN 4.1)")
    ap.add_argument("--python-sequencing", action="store_true", help="with --classification: sequence the iteration in Python over the *_dev calls "
                                                                         "instead of the native driver thx_classify_iterate (A/B)")
    ap.add_argument("--check-native", action="store_true", help="with --classification --warmup 0 --steps 1: run the Python sequencing once first and "
                                                                    "report the native driver's differences from it (bit-identical when one batch holds every image)")
    ap.add_argument("--scan-only", action="store_true", help="with --classification: the global scanning stage on its own")
    ap.add_argument("--scan-images", type=int, default=0, help="images of the classification bench (default 6250 = 50 000 / 8 GPUs; 1024 with --scan-only)")
    ap.add_argument("--scan-cpu-images-per-core", type=int, default=2)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    if args.classification:
        from thunder_amd import capi
        capi.load()
        if not args.scan_images:
            args.scan_images = 1024 if args.scan_only else 6250
        if args.scan_only:
            bench_global_scan(args, dev)
        else:
            args.batch = min(args.batch, 3125)
            bench_classification_iteration(args, dev)
        return
    from thunder_amd import capi
    from thunder_amd.native import NativeRefine, make_comms, STAGES
    from thunder_amd.refine import RefineShard, shard_count
    capi.load()

    # ---- synthetic particles of this rank (generation only; every other buffer belongs to the native driver) ----
    n_local = shard_count(args.particles, rank, world)
    shard = RefineShard(args.box, n_local, dev, rank=rank, world=world, mLR=args.mLR, mLT=args.mLT,
                        nPhase=args.phases, mReco=args.mReco, batch=args.batch, particle_filter=True, allocate=False,
                        sort_view=not args.unsorted)
    shard.release_generation_state()
    shard.wg_per_cu = args.wg_per_cu   # occupancy argument of the local-search kernel (thx_refine_config.wgPerCU; 2 = the library default)

    # ---- RCCL communicators in native code (thx_comm_*): the unique ids travel through the launcher's process group,
    #      as the reference broadcasts them over MPI (gpu/src/cuthunder.cu:4192-4206) ----
    def share_from(root, uid):
        import torch.distributed as dist
        box = [uid]
        dist.broadcast_object_list(box, src=root)
        return box[0]
    hemi = wcomm = None
    if world > 1:
        hemi, wcomm = make_comms(rank, world, share_from)
    nat = NativeRefine(shard, hemi, wcomm, norm_correction=not args.no_norm_correction)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        nat.reset()
        nat.run(args.warmup)
    nat.reset()
    nat.stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    fsc = nat.run(args.steps, timed=True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total_particles = args.particles * args.steps
        value = total_particles / dt
        # per-launch averages of the two gather / scatter kernels: HIP events recorded by the native driver on its launch
        # stream around every thx_expect_local_dev / thx_insert_dev call of the timed iterations
        st = nat.stats()
        nPxl, nPxlM = st.nPxl, st.nPxlM
        ins_ms = st.insertMs / max(1, st.insertLaunches)
        ins_n = st.insertImages / max(1, st.insertLaunches)
        ins_bytes = ins_n * shard.mReco * nPxlM * INSERT_BYTES_PER_PIXEL_SAMPLE
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
        traffic, pmc_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        lds_rate = None
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("box") == args.box:
                    traffic = j["hbm_bytes_per_image_phase"] * exp_n
                    pmc_src = j.get("source")
                    lds_rate = j.get("lds_add_u32_per_s")
            except Exception:
                traffic = None
        # insertion against its own bound.  The brick-sorted form (k_bin + segment sort + k_acc, thx_insert_sort.hip) issues
        # 24 ds_add_u64 (8 voxels x re, im, T) per RECORD = per (listed pixel, GROUP of draws) -- the insert plan merges the
        # draws of an image that share a rotation; the driver counts the groups its launches processed
        # (thx_insert_groups_total) -- and moves each record through HBM once each way (28 bytes written by k_bin, read by
        # k_acc).  lds_add_frac prices the whole insertion call (plan, k_bin, sort, k_acc) against the chip's measured
        # ds_add_u64 rate (tools/lds_atomic_bench.hip, recorded in profiles/pmc_traffic.json).
        groups_per_image = st.insertGroups / max(1, st.insertImages)
        lds_rate64 = None
        try:
            lds_rate64 = json.load(open(pmc)).get("lds_add_u64_per_s")
        except Exception:
            pass
        ins_records_per_s = ins_n * groups_per_image * nPxlM / (ins_ms * 1e-3)
        ins_terms_per_s = ins_records_per_s * 24
        out = {
            "metric": "particles/sec per refinement iteration (256\u00b3 box, 100k particles); achieved HBM GB/s" if (args.box, args.particles) == (256, 100000)
                      else "particles/sec per refinement iteration (%d\u00b3 box, %d particles); achieved HBM GB/s" % (args.box, args.particles),
            "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d synthetic %d^3 particles, 3D refinement iteration on %d GPU(s) (local search %d "
                                   "phases x %d rot x %d shifts, normCorrection%s, %d inserts, 2 half-sets, 2x reconstruct per half, FSC, "
                                   "projector refresh)" % (args.particles, args.box, world, args.phases, args.mLR,
                                                           args.mLT, " off" if args.no_norm_correction else "", args.mReco),
                       "box": args.box, "particles": args.particles, "particles_per_gpu": n_local, "nPxl": nPxl,
                       "hbm_in_use_GB": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1),
                       "pf": 2,
                       "search_state": "device particle filter (perturb / resample every phase, Philox-seeded)",
                       "particle_order": "random" if args.unsorted else "by view direction (thx_view_order_host)",
                       "driver": "native C++ iteration driver (thx_refine_iterate) through the C ABI",
                       "parallelism": "particles sharded over %d GPU(s); half-set F/T all-reduce in native RCCL "
                                      "(thx_reco_allreduce)" % world},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": kms,
                         "images_per_launch": exp_n, "algorithmic_bytes_per_launch": kbytes, "traffic_source": pmc_src},
            "kernels": {"insertion (k_bin + segment sort + k_acc)": {
                            "avg_call_ms": ins_ms, "images_per_call": ins_n, "total_ms": t_ins,
                            "groups_per_image": groups_per_image, "us_per_image": ins_ms * 1e3 / max(1.0, ins_n),
                            "records_per_s": ins_records_per_s, "lds_adds_u64_per_s": ins_terms_per_s,
                            "lds_add_frac": (ins_terms_per_s / lds_rate64) if lds_rate64 else None,
                            "record_GBps_written_plus_read": ins_records_per_s * 56 / 1e9,
                            "GBps_algorithmic_204B_per_draw": ins_bytes / (ins_ms * 1e-3) / 1e9},
                        "k_expect_local": {"avg_launch_ms": exp_ms, "images_per_launch": exp_n,
                                           "GBps_algorithmic": exp_bytes / (exp_ms * 1e-3) / 1e9, "total_ms": t_exp}},
            "stages_ms_per_step": {k: round(st.stageMs[i] / args.steps, 2) for i, k in enumerate(STAGES)},
            "balancing_rounds_per_step": st.balancingRounds / max(1, args.steps),
            "fsc_half_maps": [round(float(x), 4) for x in fsc[: 8]],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, shard, nat, out["balancing_rounds_per_step"])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
