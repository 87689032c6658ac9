"""One refinement iteration of the hot path on one GPU's shard of particles.

Host-side mirror of the reference's per-iteration control flow, restricted to the path in scope:
    Optimiser::expectation   (src/Optimiser.cpp:1141-1660, local particle-filter phases)
    Optimiser::maximization  -> reconstructRef (src/Optimiser.cpp:6711-7766): insert, prepareTF, reconstruct x2
    Model::compareTwoHemispheres FSC (src/Functions/Spectrum.cpp:302) and Model::refreshProj (src/Model.cpp:1013-1044)
and the callers either side of it, in the reference's order (src/Optimiser.cpp:3405-3480,3800-3835):
    allocPreCal row gathers (:8043-8171), allReduceSigma (:6395-6710), reCentreImg (:6065-6090), reMaskImg (:6093-6149)
All arithmetic is in the HIP library (thunder_amd.ops -> C ABI); this file only sequences launches, owns the
HBM-resident particle shard and does the half-set exchange over torch.distributed (RCCL on the GPU box, gloo in
the CPU tests -- where `backend` is injected so the orchestration can be exercised without a GPU).

The particle filter proper (Particle::perturb / resample, src/Particle.cpp) is host-side stochastic control and out of
scope: the support points of every phase are fixed, seeded inputs (SURVEY 8d "fixed-work variant").
Sharding: particle i belongs to half i mod 2; with world >= 2, rank r owns half r mod 2 (mirrors the
reference's odd/even MPI hemispheres, src/Parallel.cpp:26-36) and the ranks of a half all-reduce F and T
(MPI_Allreduce_Large over _hemi, src/Reconstructor.cpp:2383,2436); every rank of the half then reconstructs
redundantly, as the reference's ranks do.
"""
import os

import numpy as np
import torch

from . import synth


class HalfGroups:
    """the two hemisphere communicators (src/Parallel.cpp:38-57) on top of torch.distributed"""

    def __init__(self, rank=0, world=1):
        self.rank, self.world = rank, world
        self.half = rank % 2
        self.group = None
        self.leaders = (0, 1)
        if world > 2:   # with two ranks each half is one rank: nothing to reduce, no sub-communicator to build
            import torch.distributed as dist
            groups = []
            for h in (0, 1):   # every rank creates both groups, in the same order (new_group is collective)
                ranks = [r for r in range(world) if r % 2 == h]
                groups.append(dist.new_group(ranks=ranks))
            self.group = groups[self.half]

    def local_halves(self):
        return (0, 1) if self.world == 1 else (self.half,)

    def allreduce_half(self, t):
        if self.world > 2:
            import torch.distributed as dist
            # complex volumes go over the wire as their float pairs (same bytes; no reliance on complex support
            # of the collective backend)
            buf = torch.view_as_real(t) if t.is_complex() else t
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def exchange_half_maps(self, maps):
        """every rank ends up with both half maps (the reference sends them to the master for FSC,
        src/Model.cpp:375-391; here they are broadcast from the two half leaders)"""
        if self.world == 1:
            return maps[0], maps[1]
        import torch.distributed as dist
        own = maps[self.half]
        a = own.clone() if self.half == 0 else torch.empty_like(own)
        b = own.clone() if self.half == 1 else torch.empty_like(own)
        dist.broadcast(a, src=0)
        dist.broadcast(b, src=1)
        return a, b


class RefineShard:
    """HBM-resident shard of synthetic particles + one EM iteration over it (SURVEY 8d inputs)."""

    def __init__(self, N, nImg, device, rank=0, world=1, pf=2, mLR=125, mLT=9, nPhase=3, mReco=100, seed=1000,
                 batch=10240, pixelSize=1.32, snr=0.01, rL=2, ops=None, oracle_pixel_list=None, nGroup=8, groupSig=True,
                 maskFrac=0.45, particle_filter=True, transS=2.0, data=None, allocate=True, sort_view=False, coreFSC=True,
                 goldenAverage=True, solventFlatten=True, K=1, sym=None, scan=None, search="local", map_seed=20240601, nblob=40,
                 K_used=None):
        """data (optional): particles read from files instead of synthesised here -- dict(imgOri complex64 device stack
        [nImg][N][N/2+1] as Optimiser::initImg leaves _imgOri, attr float32 [nImg][7], quat [nImg][4], shift [nImg][2]
        (initial poses, e.g. the .thu columns), gid int32 [nImg] 1-based group ids, ref float32 [N]^3 initial map).
        allocate=False: only the particles are generated (imgOri, attr, poses, initial support points, reference) and
        handed to the native iteration driver (thunder_amd.native.NativeRefine), which owns every other buffer.
        K > 1: K references, every particle a slice of a uniformly drawn class (cls_true); sym: a point-group name, the references
        carry it; scan = dict(nR, nT, rScan[, mS]): a scanned grid of nR random rotations and nT shifts is drawn and the particles'
        true poses are grid points (the workload of a global search); search: what the native driver starts with."""
        if ops is None:
            from . import ops as _ops
            ops = _ops
        self.ops = ops
        self.N, self.pf, self.P = N, pf, N * pf
        self.nImg, self.dev, self.rank, self.world = nImg, device, rank, world
        # images per launch: balanced batches of at most `batch` (a short last batch leaves the chip half empty)
        nb = max(1, -(-nImg // max(1, batch)))
        batch = -(-nImg // nb)
        self.mLR, self.mLT, self.nPhase, self.mReco, self.batch = mLR, mLT, nPhase, mReco, batch
        self.rU, self.rL = N // 2 - 2, rL
        self.maxRadius = self.rU
        self.groups = HalfGroups(rank, world)
        self.halves = self.groups.local_halves()
        rng = np.random.default_rng(seed + 7919 * rank)
        # ---- pixel list (Optimiser::allocPreCalIdx): integer work, built on the host ----
        # expectation: allocPreCalIdx(_r, _rL) (src/Optimiser.cpp:631); reconstruction: allocPreCalIdx(rU, 0) (:6722)
        pl = pixel_list(N, self.rU, rL, pf)
        plM = pixel_list(N, self.rU, 0, pf)
        # The E-step does not care in which order the listed pixels are visited.  Visiting them along a Morton (Z-order)
        # curve instead of row by row keeps the volume cells consecutive pixels touch close together (reuse between
        # neighbouring pixels and rotations while the lines are still cached): 158 ms per launch against 162 ms for
        # 16x16 tiles in row-major order (THX_PIXEL_ORDER=tile, THX_TILE_ORDER=T); the reference's row-major list
        # (THX_PIXEL_ORDER=tile THX_TILE_ORDER=0) is 6 % slower than the tiles.
        order = pixel_visit_order(pl, N)
        self.e_order = order
        if order is not None:
            for k in ("iCol", "iRow", "iPxl", "iSig", "iColPad", "iRowPad"):
                pl[k] = np.ascontiguousarray(pl[k][order])
        self.pl, self.plM = pl, plM
        self.nPxl, self.nPxlM = pl["nPxl"], plM["nPxl"]
        self.iCol = torch.from_numpy(pl["iCol"]).to(device)
        self.iRow = torch.from_numpy(pl["iRow"]).to(device)
        self.iColM = torch.from_numpy(plM["iCol"]).to(device)
        self.iRowM = torch.from_numpy(plM["iRow"]).to(device)
        pos = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(plM["iCol"], plM["iRow"]))}
        e2m = torch.from_numpy(np.asarray([pos[(int(i), int(j))] for i, j in zip(pl["iCol"], pl["iRow"])],
                                          np.int64)).to(device)
        # ---- reference map(s) and their projector volumes (one per local half) ----
        self.nK, self.sym, self.search = int(K), sym, search
        assert K == 1 or (data is None and not allocate), "the Python sequencing harness runs one class; K classes go through NativeRefine"
        symR = None
        if sym:
            from .native import symmetry
            symR = symmetry(sym)["R"]
        if data is None:
            refs = np.stack([synth.blob_map(N, seed=map_seed + 100 * k, nblob=nblob, symR=symR) for k in range(K)])
            self.refs = torch.from_numpy(refs).to(device).contiguous()
            self.ref = self.refs[0]
        else:
            self.ref = torch.as_tensor(data["ref"]).to(device).contiguous()
            self.refs = self.ref[None]
        self.plan = ops.RecoPlan(N, N, pf)
        v = self.plan.set_projectee(self.ref)
        self.vols = torch.stack([v] * len(self.halves)).contiguous()
        vK = [v] + [self.plan.set_projectee(self.refs[k]) for k in range(1, K)]
        self.scan = None
        if scan is not None:
            gq = synth.random_quats(scan["nR"], rng)
            gt = np.ascontiguousarray(rng.normal(0, 3.0, size=(scan["nT"], 2)))
            self.scan = dict(scan, quat=gq, shifts=gt)
            self.scan.setdefault("mS", scan["nR"] * (1 + (len(symR) if symR is not None else 0)))
        # cell-packed copies for the E-step gathers (8x the memory, one contiguous 64-byte read per sample)
        self.use_packed = os.environ.get("THX_PACKED", "1") == "1"
        self.cells = None
        # ---- particles: pose, shift, CTF, noisy image on the pixel list ----
        self.cls_true = np.zeros(nImg, np.int32)
        if data is None:
            if self.scan is None:
                self.quat = synth.random_quats(nImg, rng)
                self.shift = rng.normal(0, 2.0, size=(nImg, 2))
            else:
                self.r_true, t_true = rng.integers(0, self.scan["nR"], nImg), rng.integers(0, self.scan["nT"], nImg)
                self.quat, self.shift = self.scan["quat"][self.r_true].copy(), self.scan["shifts"][t_true].copy()
            if K > 1:
                # (K_used < K leaves the last classes without particles: what OPTIMISER_BALANCE_CLASS exists for)
                self.cls_true = rng.integers(0, K if K_used is None else K_used, nImg).astype(np.int32)
            if sort_view:   # shard layout: particles ordered by [class, then] view direction within each half (see view_order)
                nA = (nImg + 1) // 2 if world == 1 else nImg
                for lo, hi in ((0, nA), (nA, nImg)):
                    if hi > lo:
                        perm = view_order(self.quat[lo:hi])
                        perm = perm[np.argsort(self.cls_true[lo:hi][perm], kind="stable")]
                        self.quat[lo:hi], self.shift[lo:hi] = self.quat[lo:hi][perm], self.shift[lo:hi][perm]
                        self.cls_true[lo:hi] = self.cls_true[lo:hi][perm]
            self.attr = torch.from_numpy(synth.ctf_params(nImg, rng)).to(device)
        else:
            self.quat = np.ascontiguousarray(np.asarray(data["quat"], np.float64).reshape(nImg, 4))
            self.shift = np.ascontiguousarray(np.asarray(data["shift"], np.float64).reshape(nImg, 2))
            self.attr = torch.as_tensor(np.asarray(data["attr"], np.float32).reshape(nImg, 7)).to(device).contiguous()
        # particle -> half-set.  With one rank both halves live here: the first ceil(n/2) particles are half 0, the rest
        # half 1 (contiguous ranges, so each half is one slice of every per-particle array); with world >= 2 the whole
        # shard belongs to half rank mod 2.
        if world == 1:
            nA = (nImg + 1) // 2
            self.ranges = {0: (0, nA), 1: (nA, nImg)}
        else:
            self.ranges = {self.groups.half: (0, nImg)}
        # ---- full image stacks, as the reference holds them: _imgOri (as read) and _img (re-centred + masked) ----
        # signal = CTF x slice x ramp on the rL = 0 list (+ the Hermitian mirror of the kx = 0 column, which the list
        # drops) + white noise generated in real space so that the FT is that of a real image
        self.pixelSize, self.rSig = pixelSize, N // 2 - 1
        self.maskRadiusPx = float(np.float32(maskFrac * N))
        self.nGroup, self.groupSig = nGroup, groupSig
        # script/demo_3D.json:23 "Calculate FSC Using Core Region": true; :50 "Using Golden Standard FSC": true
        self.coreFSC, self.goldenAverage, self.solventFlatten = coreFSC, goldenAverage, solventFlatten
        self.gid = rng.integers(1, nGroup + 1, nImg).astype(np.int32)          # Optimiser::_groupID (1-based, host)
        if data is not None:
            self.gid = np.ascontiguousarray(np.asarray(data["gid"], np.int32).reshape(nImg))
            self.nGroup = nGroup = int(self.gid.max())
        self.gid0 = torch.from_numpy(self.gid.astype(np.int64) - 1).to(device)
        nc = N // 2 + 1
        iPxlM = torch.from_numpy(plM["iPxl"].astype(np.int64)).to(device)
        col0 = np.nonzero((plM["iCol"] == 0) & (plM["iRow"] > 0))[0]
        mirror_src = torch.from_numpy(col0.astype(np.int64)).to(device)
        mirror_dst = torch.from_numpy(((N - plM["iRow"][col0]).astype(np.int64)) * nc).to(device)
        self.iPxlE = torch.from_numpy(pl["iPxl"]).to(device)
        self.iPxlM = torch.from_numpy(plM["iPxl"]).to(device)
        self.iSigE = torch.from_numpy(pl["iSig"].astype(np.int64)).to(device)
        self.imgOri = torch.zeros((nImg, N, nc), dtype=torch.complex64, device=device)
        gen = torch.Generator(device=device)
        gen.manual_seed(seed + 31 * rank)
        if data is not None:
            self.imgOri = torch.as_tensor(data["imgOri"]).to(device).contiguous()
            flat = self.imgOri.view(nImg, -1)[:, iPxlM]
            self.sigma2 = float((flat.abs() ** 2).mean().item()) / 2.0     # initial noise model: the images' own power
            del flat
        for b0 in range(0, nImg if data is None else 0, batch):
            b1 = min(nImg, b0 + batch)
            rot = ops.rotmat(torch.from_numpy(self.quat[b0:b1]).to(device))
            sl = ops.project(v, rot, self.iColM, self.iRowM, pf)
            for k in range(1, K):   # particles of the other classes: slices of their own reference
                sel = torch.from_numpy(np.nonzero(self.cls_true[b0:b1] == k)[0]).to(device)
                if sel.numel():
                    sl[sel] = ops.project(vK[k], rot[sel].contiguous(), self.iColM, self.iRowM, pf)
            ramp = ops.translate(torch.from_numpy(self.shift[b0:b1]).to(device), self.iColM, self.iRowM, N)
            sig = sl * ramp * ops.ctf(self.attr[b0:b1].contiguous(), pixelSize, self.iColM, self.iRowM, N)
            if b0 == 0:
                p_sig = float((sig.abs() ** 2).mean().item())
                self.sigma2 = p_sig / snr / 2.0  # per real component of an FT coefficient
            flat = self.imgOri[b0:b1].view(b1 - b0, -1)
            flat[:, iPxlM] = sig
            flat[:, mirror_dst] = sig[:, mirror_src].conj()
            for c0 in range(b0, b1, 1024):   # var(FT component) = N^2 var(pixel) / 2
                c1 = min(b1, c0 + 1024)
                rl = torch.randn((c1 - c0, N, N), generator=gen, device=device, dtype=torch.float32)
                self.imgOri[c0:c1] += torch.fft.rfft2(rl) * float(np.sqrt(2.0 * self.sigma2) / N)
            del sig, sl, ramp, flat
        self.transS, self.transQ = transS, 0.05                       # TRANS_Q, include/Optimiser.h:67
        self.pfL, self.pfS, self.peakFactorR = 2.0, 0.5, 1e-3         # script/demo_3D.json:71-73, PEAK_FACTOR_MIN
        self.use_pf = particle_filter
        self.wg_per_cu = 2 if particle_filter else 0
        self.pf_seed, self.pf_call = seed + 104729 * rank, 0
        self.job_seed = seed  # the native driver's Philox seed: ONE number for the whole job (images are told apart by their index over all ranks)
        self.img_id0 = 0      # this rank's first image in the Philox numbering (the native driver: images of the ranks before it)
        if self.use_pf:
            q0 = synth.perturb_quats(self.quat, mLR, 0.02, rng)
            t0 = self.shift[:, None, :] + rng.normal(0, 0.5, size=(nImg, mLT, 2))
            self.pf0 = dict(r=torch.from_numpy(np.ascontiguousarray(q0)).to(device),
                            t=torch.from_numpy(np.ascontiguousarray(t0)).to(device))
            self.pf_state = {}
        self.allocated = allocate
        if not allocate:
            return
        # M-step rows (_imgOri on the rL = 0 list) never change; E-step rows are re-gathered from _img every iteration
        self.ctfM = ops.ctf(self.attr, pixelSize, self.iColM, self.iRowM, N)
        self.datM = ops.gather_pixels(self.imgOri, self.iPxlM, N)
        self.ctfP = self.ctfM[:, e2m].contiguous()
        self.img = torch.empty_like(self.imgOri)
        self.datP = torch.empty((nImg, self.nPxl), dtype=torch.complex64, device=device)
        self.sigRcpP = torch.empty((nImg, self.nPxl), dtype=torch.float32, device=device)
        self.offset = torch.zeros((nImg, 2), dtype=torch.float64, device=device)
        # ---- fixed-work support points for every phase (seeded) ----
        stds = [0.02 / (2 ** p) for p in range(nPhase)]
        self.rotP, self.tranP = [], []
        for p in range(0 if particle_filter else nPhase):   # (the particle filter brings its own support points)
            q = synth.perturb_quats(self.quat, mLR, stds[p], rng)
            self.rotP.append(ops.rotmat(torch.from_numpy(q.reshape(-1, 4)).to(device)).reshape(nImg, mLR, 9))
            t = self.shift[:, None, :] + rng.normal(0, 0.5 / (2 ** p), size=(nImg, mLT, 2))
            t[:, 0, :] = self.shift
            self.tranP.append(torch.from_numpy(np.ascontiguousarray(t)).to(device))
        self.tranP0 = [t.clone() for t in self.tranP]
        self.w = torch.full((nImg,), 1.0 / mReco, dtype=torch.float32, device=device)
        nV = len(self.halves)
        self.F = torch.zeros((nV, self.P, self.P, self.P // 2 + 1), dtype=torch.complex64, device=device)
        self.T = torch.zeros((nV, self.P, self.P, self.P // 2 + 1), dtype=torch.float32, device=device)
        from . import capi
        nmax = max(hi - lo for lo, hi in self.ranges.values())
        nb = max(1, -(-nmax // self.batch))
        self.batch = -(-nmax // nb)
        need = capi.load().thx_expect_local_workspace(min(self.batch, nmax), mLR, mLT, 1)
        # one workspace / RNG / reconstruction plan per local half so that the two halves can run as independent chains
        self.ws = [torch.empty(need, dtype=torch.uint8, device=device) for _ in self.halves]
        self.gens = []
        for vi in range(nV):
            gh = torch.Generator(device=device)
            gh.manual_seed(seed + 31 * rank + 977 * vi)
            self.gens.append(gh)
        self.plans = [self.plan] + [ops.RecoPlan(N, N, pf) for _ in range(nV - 1)]
        self.insert_ms = []   # per-launch durations of the insertion kernel (HIP events on the launch stream)
        self.expect_ms = []
        self.stage_ms = {}    # name -> list of (event, event): coarse per-stage timing of the timed run
        self.reco_rounds = []  # balancing rounds of every reconstruction (Reconstructor::reconstruct's gridding loop)
        self.last = {}
        self.sig = torch.empty((nV, nGroup, self.rSig), dtype=torch.float32, device=device)
        self.sigRcp = torch.empty_like(self.sig)
        self.reset_reference()

    def release_generation_state(self):
        """after the particles have been handed to the native driver: drop the projector volume and the reconstruction
        plan that generated them (about 2.5 GB at N = 256)"""
        self.vols = None
        self.cells = None
        if self.plan is not None:
            self.plan.close()
            self.plan = None
        torch.cuda.empty_cache()

    # -----------------------------------------------------------------------------------------
    def _stage(self, name, timed):
        """context manager recording a HIP-event pair around a stage on the current stream when `timed`"""
        shard = self

        class _S:
            def __enter__(self_):
                if timed:
                    self_.e0 = torch.cuda.Event(enable_timing=True)
                    self_.e0.record()

            def __exit__(self_, *a):
                if timed:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    shard.stage_ms.setdefault(name, []).append((self_.e0, e1))
                return False
        return _S()

    def refresh_rows(self, vi):
        """Optimiser::allocPreCal(mask = true, ...) rows of local half `vi` (src/Optimiser.cpp:8043-8171): _datP from
        the masked stack through iPxl, _sigRcpP[l][p] = _sigRcp(groupID[l] - 1, iSig[p])"""
        lo, hi = self.ranges[self.halves[vi]]
        self.ops.gather_pixels(self.img[lo:hi], self.iPxlE, self.N, out=self.datP[lo:hi])
        self.sigRcpP[lo:hi] = self.sigRcp[vi][self.gid0[lo:hi]][:, self.iSigE]

    def top_pose(self, vi, wR, wT):
        """Particle::rank1st of the last phase: the support point with the largest weight"""
        p = self.nPhase - 1
        lo, hi = self.ranges[self.halves[vi]]
        if self.use_pf:   # Particle::calRank1st already stored them (thx_pf_update_dev)
            st = self.pf_state
            return self.ops.rotmat(st["topR"][lo:hi].contiguous()), st["topT"][lo:hi].clone()
        ar = torch.arange(hi - lo, device=self.dev)
        return (self.rotP[p][lo:hi][ar, wR.argmax(1)].contiguous(), self.tranP[p][lo:hi][ar, wT.argmax(1)].contiguous())

    def sigma_update(self, vi, rotTop, tranTop):
        """Optimiser::allReduceSigma (src/Optimiser.cpp:6395-6710) for local half `vi`"""
        ops = self.ops
        lo, hi = self.ranges[self.halves[vi]]
        spec = ops.sigma_spectra(self.vols[vi:vi + 1], self.P, self.pf, self.rU, self.rSig, self.img[lo:hi],
                                 self.imgOri[lo:hi], self.attr[lo:hi], self.pixelSize, rotTop, tranTop,
                                 self.offset[lo:hi])
        acc = tuple(torch.zeros((self.nGroup, self.rSig + 1), dtype=torch.float32, device=self.dev) for _ in range(3))
        ops.sigma_accum(acc, spec, self.gid[lo:hi], self.groupSig)
        for a in acc:
            self.groups.allreduce_half(a)
        sig, rcp = ops.sigma_final(acc, self.maskRadiusPx * self.pixelSize, self.N, self.pixelSize, self.groupSig)
        self.sig[vi], self.sigRcp[vi] = sig, rcp

    def recentre_and_remask(self, vi, tranTop):
        """Optimiser::reCentreImg (src/Optimiser.cpp:6065-6090) + reMaskImg (:6093-6149) for local half `vi`"""
        ops = self.ops
        lo, hi = self.ranges[self.halves[vi]]
        self.offset[lo:hi] -= tranTop
        for t in self.tranP:                              # _par[l].setT(t - tran)
            t[lo:hi] -= tranTop[:, None, :]
        if self.use_pf:
            self.pf_state["t"][lo:hi] -= tranTop[:, None, :]
            self.pf_state["topT"][lo:hi] -= tranTop       # setTopT(topT - tran)
        ops.translate_image(self.imgOri[lo:hi], self.offset[lo:hi], out=self.img[lo:hi])
        ops.remask(self.img[lo:hi], self.maskRadiusPx, 6.0)

    # -----------------------------------------------------------------------------------------
    def expectation(self, vi, timed=False):
        """nPhase particle-filter phases over the images of local half `vi` (HOT LOOP B)"""
        ops = self.ops
        lo, hi = self.ranges[self.halves[vi]]
        n = hi - lo
        wR = torch.empty((n, self.mLR), dtype=torch.float32, device=self.dev)
        wT = torch.empty((n, self.mLT), dtype=torch.float32, device=self.dev)
        vol = self.vols[vi:vi + 1]
        if self.use_packed and self.cells is None:
            self.cells = ops.pack_projector(self.vols, self.P)
        st = self.pf_state if self.use_pf else None
        for p in range(self.nPhase):
            for b0 in range(lo, hi, self.batch):
                b1 = min(hi, b0 + self.batch)
                pR = pT = None
                if self.use_pf:
                    # Particle::perturb, then the phase's support points are the filter's own (src/Optimiser.cpp:1186-1208)
                    sl = slice(b0, b1)
                    call = self.iter_count * 1024 + 8 + 2 * p      # thx_refine_iterate's numbering: perturb 8 + 2 p, update 9 + 2 p
                    f = self.pfL if p == 0 else self.pfS
                    ops.pf_perturb(st["r"][sl], st["t"][sl], st["wR"][sl], st["wT"][sl], st["k"][sl], st["s"][sl], f, f,
                                   self.transS, self.transQ, self.pf_seed, call, img0=self.img_id0 + b0)
                    rotB = ops.rotmat(st["r"][sl].reshape(-1, 4)).reshape(b1 - b0, self.mLR, 9)
                    tranB, pR, pT = st["t"][sl], st["wR"][sl], st["wT"][sl]
                else:
                    rotB, tranB = self.rotP[p][b0:b1], self.tranP[p][b0:b1]
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                r = ops.expect_local(self.cells[vi:vi + 1] if self.use_packed else vol, self.P, self.pf, self.N, self.iCol,
                                     self.iRow, self.datP[b0:b1], self.ctfP[b0:b1], self.sigRcpP[b0:b1], rotB, tranB, nD=1,
                                     pR=pR, pT=pT, workspace=self.ws[vi], packed=self.use_packed, wg_per_cu=self.wg_per_cu)
                if timed:
                    e1.record()
                    self.expect_ms.append((e0, e1, b1 - b0))
                if self.use_pf:
                    ops.pf_update(st["r"][sl], st["t"][sl], st["wR"][sl], st["wT"][sl], r.wR, r.wT, st["k"][sl], st["s"][sl],
                                  st["topR"][sl], st["topT"][sl], self.peakFactorR, self.pf_seed, call + 1, img0=self.img_id0 + b0)
                if p == self.nPhase - 1:
                    wR[b0 - lo:b1 - lo] = r.wR
                    wT[b0 - lo:b1 - lo] = r.wT
        return wR, wT

    def draw_reco(self, vi, wR, wT):
        """mReco draws per image for the insertion, as the reference makes them: the particle filter is first RESAMPLED
        by weight to mLR / mLT support points (Particle::resample at the end of the phase, src/Optimiser.cpp:1470-1488),
        then Particle::rand picks uniformly among the resampled points (src/Particle.cpp:2109-2178).  Seeded."""
        p = self.nPhase - 1
        lo, hi = self.ranges[self.halves[vi]]
        n, gen = hi - lo, self.gens[vi]
        if self.use_pf:   # the filter has been resampled by thx_pf_update_dev: Particle::rand = a uniform pick
            st = self.pf_state
            return self.ops.draw_reco(st["r"][lo:hi], st["t"][lo:hi], self.mReco, self.pf_seed, self.iter_count * 1024 + 1000,
                                      self.img_id0 + lo)
        rsR = torch.multinomial(wR.clamp_min(1e-30), self.mLR, replacement=True, generator=gen)   # resample
        rsT = torch.multinomial(wT.clamp_min(1e-30), self.mLT, replacement=True, generator=gen)
        uR = torch.randint(0, self.mLR, (n, self.mReco), device=self.dev, generator=gen)          # rand
        uT = torch.randint(0, self.mLT, (n, self.mReco), device=self.dev, generator=gen)
        iR = torch.gather(rsR, 1, uR)
        iT = torch.gather(rsT, 1, uT)
        rot = torch.gather(self.rotP[p][lo:hi], 1, iR[:, :, None].expand(-1, -1, 9)).contiguous()
        tran = torch.gather(self.tranP[p][lo:hi], 1, iT[:, :, None].expand(-1, -1, 2)).contiguous()
        return rot, tran

    def insertion(self, vi, rot, tran, timed=False):
        """HOT LOOP C: mReco trilinear insertions per image of local half `vi` into its F / T"""
        ops = self.ops
        lo, hi = self.ranges[self.halves[vi]]
        F, T = self.F[vi], self.T[vi]
        F.zero_()
        T.zero_()
        for b0 in range(lo, hi, self.batch):
            b1 = min(hi, b0 + self.batch)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.insert(F, T, self.P, self.datM[b0:b1], self.ctfM[b0:b1], self.w[b0:b1], rot[b0 - lo:b1 - lo],
                       tran[b0 - lo:b1 - lo], self.iColM, self.iRowM, self.pf, self.N, offS=self.offset[b0:b1])
            if timed:
                e1.record()
                self.insert_ms.append((e0, e1, b1 - b0))

    def reduce_and_first_map(self, vi):
        """half-set reduce, prepareTF (normalisation; C1: no symmetry sweep), reconstruct with MAP off
        (setJoinHalf(true): OPTIMISER_RECONSTRUCT_JOIN_HALF, src/Optimiser.cpp:7326-7352)"""
        ops, g = self.ops, self.groups
        g.allreduce_half(self.T[vi])
        g.allreduce_half(self.F[vi])
        ops.normalise_TF(self.F[vi], self.T[vi], self.P)
        m = self.plans[vi].reconstruct(self.F[vi], self.T[vi], self.maxRadius, joinHalf=True, MAP=False, gridCorr=True)
        self.reco_rounds.append(self.plans[vi].last_iters)
        return m

    def fsc_of(self, a, b):
        """compareTwoHemispheres(true, false) (src/Optimiser.cpp:7547): Model::_FSC over rU shells, core-mask corrected
        when coreFSC; returned padded to N / 2 entries"""
        ops = self.ops
        coreR = float(np.rint(np.float32(self.maskRadiusPx))) if self.coreFSC else 0.0
        f = ops.compare_hemispheres(ops.fft3d_fw(a), ops.fft3d_fw(b), self.N, self.rU, coreR=coreR, seed=self.pf_seed,
                                    call_id=0x40000000 + 32 * self.iter_count)
        out = np.zeros(self.N // 2, np.float32)
        out[:self.rU] = f
        return out

    def final_map(self, vi):
        """reconstruct with MAP on and the FSC Model::resetReco handed to the reconstructor at the end of the PREVIOUS
        iteration (src/Model.cpp:1086,1122), joinHalf on (src/Optimiser.cpp:7574-7600)"""
        m = self.plans[vi].reconstruct(self.F[vi], self.T[vi], self.maxRadius, FSC=self.fsc_reco, joinHalf=True, MAP=True,
                                       gridCorr=True)
        self.reco_rounds.append(self.plans[vi].last_iters)
        return m

    def average_flatten_refresh(self, maps):
        """compareTwoHemispheres(false, true, AVERAGE_TWO_HEMISPHERE_THRES) (gold-standard averaging inside r = Model::resolutionP(0.95)
        of this iteration's FSC: MODEL_RESOLUTION_BASE_AVERAGE, src/Model.cpp:616-674), Optimiser::solventFlatten's spherical
        mask, Model::refreshProj; maps = {half: MAP-on map} with BOTH halves present"""
        ops, N = self.ops, self.N
        if self.goldenAverage:
            f = self.last_fsc
            avgR = 1
            while avgR < self.rU and not (f[avgR] < np.float32(0.95)):     # resP(fsc, 0.95, 1, 1, false), src/Functions/Spectrum.cpp:339-363
                avgR += 1
            avgR -= 1
            A, B = ops.fft3d_fw(maps[0]), ops.fft3d_fw(maps[1])
            ops.compare_hemispheres(A, B, N, self.rU, fsc=False, avg_r=avgR)
            maps = {0: ops.fft3d_bw(A, N), 1: ops.fft3d_bw(B, N)}
        for vi, h in enumerate(self.halves):
            m = maps[h]
            if self.solventFlatten:
                ops.soft_mask_volume(m, self.maskRadiusPx, 6.0, 0.0)
            self.vols[vi] = self.plans[vi].set_projectee(m)   # Model::refreshProj
            if self.use_packed and self.cells is not None:
                self.cells[vi] = ops.pack_projector(self.vols[vi:vi + 1], self.P)[0]
        return maps

    def em_stage(self, vi, timed=False):
        """rows -> expectation -> sigma update -> draws -> insertion for local half `vi`; returns the top shifts"""
        with self._stage("rows", timed):
            self.refresh_rows(vi)
        with self._stage("expectation", timed):
            wR, wT = self.expectation(vi, timed)
        with self._stage("sigma", timed):
            rotTop, tranTop = self.top_pose(vi, wR, wT)
            self.sigma_update(vi, rotTop, tranTop)
        with self._stage("insertion", timed):
            rot, tran = self.draw_reco(vi, wR, wT)
            self.insertion(vi, rot, tran, timed)
        return tranTop

    def iteration(self, timed=False):
        """one EM iteration, the local halves one after the other on the current stream"""
        maps, top = {}, {}
        for vi, h in enumerate(self.halves):
            top[vi] = self.em_stage(vi, timed)
        with self._stage("reconstruct", timed):
            for vi, h in enumerate(self.halves):
                maps[h] = self.reduce_and_first_map(vi)
            a, b = self.groups.exchange_half_maps(maps)
            fsc = self.fsc_of(a, b)
            self.last_fsc = fsc
            for vi, h in enumerate(self.halves):
                maps[h] = self.final_map(vi)
            if self.goldenAverage:
                a, b = self.groups.exchange_half_maps(maps)
                maps = {0: a, 1: b}
            maps = self.average_flatten_refresh(maps)
            self.fsc_reco = fsc[:self.rU].copy()           # Model::resetReco
            self.iter_count += 1
        with self._stage("recentre_remask", timed):
            for vi, h in enumerate(self.halves):
                self.recentre_and_remask(vi, top[vi])
        self.last["fsc"], self.last["maps"] = fsc, maps
        return fsc

    def run(self, steps, timed=False):
        """`steps` EM iterations, the local halves back to back.  (Running the two half-set chains on two streams was measured
        in round 1: +1.7 %, every stage already keeps the chip busy; removed.)"""
        fsc = None
        for _ in range(steps):
            fsc = self.iteration(timed)
        return fsc

    def reset_reference(self):
        """back to the state before the first iteration: initial reference, no re-centring offset, masked copies of the
        images as read (Optimiser::initImg masks them on load), flat initial noise model, initial support points"""
        v = self.plan.set_projectee(self.ref)
        for vi in range(self.vols.shape[0]):
            self.vols[vi] = v
        self.cells = None   # rebuilt on the next expectation
        self.fsc_reco = np.ones(self.rU, np.float32)       # Model::initProjReco: setFSC(vec::Constant(_rU, 1))
        self.iter_count = 0
        self.offset.zero_()
        for t, t0 in zip(self.tranP, self.tranP0):
            t.copy_(t0)
        self.img.copy_(self.imgOri)
        self.ops.remask(self.img, self.maskRadiusPx, 6.0)
        self.sig.fill_(self.sigma2)
        self.sigRcp.fill_(-0.5 / self.sigma2)
        if self.use_pf:
            st, n = self.pf_state, self.nImg
            st["r"], st["t"] = self.pf0["r"].clone(), self.pf0["t"].clone()
            st["wR"] = torch.full((n, self.mLR), 1.0 / self.mLR, dtype=torch.float64, device=self.dev)
            st["wT"] = torch.full((n, self.mLT), 1.0 / self.mLT, dtype=torch.float64, device=self.dev)
            st["k"], st["s"] = self.ops.pf_cal_vari(st["r"], st["t"], self.pf_seed, 3, img0=self.img_id0)   # Particle::load -> calVari
            st["topR"] = st["r"][:, 0].contiguous()
            st["topT"] = st["t"][:, 0].contiguous()
            self.pf_call = 0
        for vi in range(len(self.halves)):
            self.refresh_rows(vi)


def view_order(quat):
    """Permutation that orders particles along a Morton curve over their view direction (the normal of the central slice,
    R e_z, folded to one hemisphere; Lambert equal-area map to a 1024 x 1024 grid): images next to each other in a launch
    cut the volume along nearly the same plane, so the cells one image's cloud of rotations fetches are still in L2 /
    Infinity Cache when its neighbours ask for them (-8 % E-step time at 100 k particles).  Host-side layout work, like the
    pixel-visit order; thx_view_order_host is the C twin (tests compare the two).  In a running refinement the key is the
    previous iteration's top rotation."""
    q0, q1, q2, q3 = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    n = np.stack([2 * (q1 * q3 + q0 * q2), 2 * (q2 * q3 - q0 * q1), 1 - 2 * (q1 * q1 + q2 * q2)], axis=1)
    n = n * np.where(n[:, 2:3] < 0, -1.0, 1.0)
    s = np.sqrt(1.0 / (1.0 + n[:, 2]))
    X = np.floor((n[:, 0] * s + 1) * 0.5 * 1023).clip(0, 1023).astype(np.uint32)
    Y = np.floor((n[:, 1] * s + 1) * 0.5 * 1023).clip(0, 1023).astype(np.uint32)

    def spread(v):
        o = np.zeros_like(v)
        for b in range(10):
            o |= ((v >> b) & 1) << (2 * b)
        return o
    return np.argsort(spread(X) | (spread(Y) << 1), kind="stable")


def pixel_visit_order(pl, N):
    """Permutation of the E-step pixel list: Morton (Z-order) curve over (iCol, iRow + N) by default,
    THX_PIXEL_ORDER=tile -> TxT tiles in row-major order (THX_TILE_ORDER=T, 0 = the reference's row-major list: None)."""
    if os.environ.get("THX_PIXEL_ORDER", "morton") == "morton":
        def _spread(v):
            v = v.astype(np.uint32)
            out = np.zeros_like(v)
            for b in range(10):
                out |= ((v >> b) & 1) << (2 * b)
            return out
        return np.argsort(_spread(pl["iCol"]) | (_spread(pl["iRow"] + N) << 1), kind="stable")
    tile = int(os.environ.get("THX_TILE_ORDER", "16"))
    if tile > 0:
        return np.lexsort((pl["iCol"], pl["iRow"], pl["iCol"] // tile, (pl["iRow"] + N) // tile))
    return None


def pixel_list(N, rU, rL, pf=2):
    """Optimiser::allocPreCalIdx (src/Optimiser.cpp:7991-8041) -- host-side integer work of the product path
    (its oracle twin is orc_pixel_list; tests compare the two)."""
    iCol, iRow, iPxl, iSig = [], [], [], []
    rU2, rL2 = np.float32(rU) ** 2, np.float32(rL) ** 2
    lim = int(rU + 1)
    for j in range(-lim, lim):
        for i in range(0, lim + 1):
            if i == 0 and j < 0:
                continue
            u = np.float32(i * i + j * j)
            if u < rU2 and u >= rL2:
                v = int(np.rint(np.hypot(float(i), float(j))))
                if v < rU and v >= rL:
                    iPxl.append((j if j >= 0 else j + N) * (N // 2 + 1) + i)
                    iCol.append(i)
                    iRow.append(j)
                    iSig.append(v)
    a = lambda x: np.asarray(x, np.int32)
    return dict(iCol=a(iCol), iRow=a(iRow), iPxl=a(iPxl), iSig=a(iSig), iColPad=a(iCol) * pf, iRowPad=a(iRow) * pf,
                nPxl=len(iCol))


def half_part(nTotal, rank, world):
    """[lo, hi) of `rank`'s particles in the ONE-RANK layout of a job of nTotal particles ([0, ceil(n / 2)) = half 0, the rest = half
    1): rank r holds the (r // 2)-th contiguous part of half r mod 2.  Dealing a job this way -- and telling the native driver where
    the shard starts (thx_refine_set_image_base) -- makes an N-rank run draw for every image what the one-rank run draws."""
    if world == 1:
        return 0, nTotal
    nA = (nTotal + 1) // 2
    h = rank % 2
    lo, hi = (0, nA) if h == 0 else (nA, nTotal)
    H = (world - h + 1) // 2
    j, n = rank // 2, hi - lo
    return lo + (n * j) // H, lo + (n * (j + 1)) // H


def take_shard(full, rank, world):
    """the part of a one-rank RefineShard (allocate=False) that `rank` of `world` owns (half_part): a shallow copy with the
    per-particle arrays cut, for thunder_amd.native.NativeRefine -- tests/_rank_worker.py runs 2 and 4 ranks on the particles of a
    one-rank job this way"""
    import copy
    import types
    assert full.world == 1 and not full.allocated
    lo, hi = half_part(full.nImg, rank, world)
    s = copy.copy(full)
    s.rank, s.world, s.nImg = rank, world, hi - lo
    s.groups = types.SimpleNamespace(rank=rank, world=world, half=rank % 2, group=None)
    s.halves = (rank % 2,) if world > 1 else (0, 1)
    s.ranges = {rank % 2: (0, hi - lo)} if world > 1 else full.ranges
    s.img_base = lo
    for name in ("imgOri", "attr"):
        setattr(s, name, getattr(full, name)[lo:hi].contiguous())
    for name in ("gid", "quat", "shift", "cls_true"):
        setattr(s, name, np.ascontiguousarray(getattr(full, name)[lo:hi]))
    if getattr(full, "r_true", None) is not None:
        s.r_true = full.r_true[lo:hi]
    s.pf0 = {k: v[lo:hi].contiguous() for k, v in full.pf0.items()}
    if getattr(full, "cls0", None) is not None:
        s.cls0 = np.ascontiguousarray(full.cls0[lo:hi])
    nb = max(1, -(-s.nImg // max(1, full.batch)))
    s.batch = -(-s.nImg // nb)
    return s


def shard_count(nTotal, rank, world):
    """number of particles `rank` owns when nTotal particles are sharded as shard_indices does"""
    return int(len(shard_indices(nTotal, rank, world)))


def shard_indices(nTotal, rank, world):
    """global particle indices owned by `rank`: particle i belongs to half i mod 2 (gold-standard split); with
    world >= 2 the ranks r with r mod 2 == h share half h round-robin (mirrors src/Parallel.cpp:26-36 +
    Database::assign).  world == 1 owns everything."""
    idx = np.arange(nTotal)
    if world == 1:
        return idx
    h = rank % 2
    mine = idx[idx % 2 == h]
    peers = [r for r in range(world) if r % 2 == h]
    return mine[peers.index(rank)::len(peers)]
