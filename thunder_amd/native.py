"""Front-end of the native iteration driver (thx_refine_*, thunder_amd/csrc/thx_refine.hip) and of the native RCCL
communicators (thx_comm_*, thx_comm.hip).  No arithmetic and no sequencing happens here: the EM iteration -- rows,
particle-filter phases, sigma update, draws, insertion, half-set reduce, reconstructions, FSC, projector refresh,
re-centring / re-masking -- runs in C++ behind one call, thx_refine_iterate.  Python only generates the synthetic
particles (RefineShard(allocate=False)), shares the RCCL unique id between the ranks and prints the result.
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import RefineConfig, RefineStats, RefineView, ptr, stream_ptr

STAGES = ("rows", "expectation", "sigma", "insertion", "reconstruct", "recentre_remask", "norm_correction", "global_scan")


class Comm:
    """thx_comm: an RCCL communicator created from a 128-byte unique id (gpu/src/cuthunder.cu:4192-4206).
    `share(id_bytes or None)` is the launcher's broadcast of the id from the group's root (MPI_Bcast in the reference;
    torch.distributed's object broadcast in bench.py; a pipe in tests/cpp/iteration.cpp)."""

    def __init__(self, rank, size, share):
        self.rank, self.size = rank, size
        uid = None
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            capi.call("thx_comm_unique_id", C.cast(buf, C.c_void_p))
            uid = bytes(buf)
        uid = share(uid)
        assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
        h = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        capi.call("thx_comm_init", C.byref(h), C.cast(buf, C.c_void_p), rank, size)
        self._h = h

    @property
    def handle(self):
        return self._h

    def allreduce(self, t):
        n = t.numel() * (2 if t.is_complex() else 1)
        name = {torch.float32: "thx_comm_allreduce_f32", torch.complex64: "thx_comm_allreduce_f32",
                torch.float64: "thx_comm_allreduce_f64", torch.int32: "thx_comm_allreduce_i32"}[t.dtype]
        capi.call(name, self._h, ptr(t), n, stream_ptr())
        return t

    def broadcast(self, t, root):
        capi.call("thx_comm_broadcast", self._h, ptr(t), t.numel() * t.element_size(), root, stream_ptr())
        return t

    def close(self):
        if self._h is not None:
            capi.call("thx_comm_destroy", self._h)
            self._h = None


def make_comms(rank, world, share_from):
    """(hemi, world) communicators of one rank: hemi spans the ranks r with r % 2 == rank % 2 (the reference's odd / even
    hemispheres, src/Parallel.cpp:26-36); share_from(root_world_rank, id_or_None) -> id is the launcher's broadcast.
    With world <= 2 a half is one rank: no hemisphere communicator."""
    if world == 1:
        return None, None
    wcomm = Comm(rank, world, lambda uid: share_from(0, uid))
    hemi = None
    if world > 2:
        h = rank % 2
        peers = [r for r in range(world) if r % 2 == h]
        # both hemispheres bootstrap (every rank takes part in both broadcasts, uses its own half's id)
        ids = {}
        for hh in (0, 1):
            root = hh   # world rank hh leads half hh
            uid = None
            if rank == root:
                buf = (C.c_ubyte * 128)()
                capi.call("thx_comm_unique_id", C.cast(buf, C.c_void_p))
                uid = bytes(buf)
            ids[hh] = share_from(root, uid)
        hemi = Comm.__new__(Comm)
        hemi.rank, hemi.size = peers.index(rank), len(peers)
        hh = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(ids[h])
        capi.call("thx_comm_init", C.byref(hh), C.cast(buf, C.c_void_p), hemi.rank, hemi.size)
        hemi._h = hh
    return hemi, wcomm


class NativeRefine:
    """thx_refine handle over the particles of a RefineShard (which only has to have GENERATED them: allocate=False).
    Optional attributes of the shard select what round 4 added to the driver: nK + refs [K][N]^3 (classes), scan = dict(quat
    [nR][4], shifts [nT][2], rScan[, minK, minS, batch]) (global search), sym = a point-group name ("C4", "D2", ...),
    search = "local" | "global" | "ctf", pfSGlobal, peakFactorC, balanceClass, mLD / ctfRefineS / pfSCTF, cls0 [nImg]."""

    def __init__(self, shard, hemi=None, world=None, pixel_order=1, max_phase=0, norm_correction=False):
        self.shard = shard
        s = shard
        g = lambda name, default=None: getattr(s, name, default)
        cfg = RefineConfig()
        cfg.N, cfg.pf, cfg.nImg = s.N, s.pf, s.nImg
        if s.world == 1:
            cfg.halfOfRank, cfg.nHalfA = -1, s.ranges[0][1]
        else:
            cfg.halfOfRank, cfg.nHalfA = s.groups.half, 0
        cfg.mLR, cfg.mLT, cfg.nPhase, cfg.mReco, cfg.batch = s.mLR, s.mLT, s.nPhase, s.mReco, s.batch
        cfg.maxPhase = max_phase
        cfg.rL, cfg.nGroup, cfg.groupSig = s.rL, s.nGroup, 1 if s.groupSig else 0
        cfg.pixelOrder, cfg.wgPerCU = pixel_order, s.wg_per_cu
        cfg.pixelSize, cfg.maskRadiusPx, cfg.sigma2Init = s.pixelSize, s.maskRadiusPx, s.sigma2
        cfg.transS, cfg.transQ, cfg.pfL, cfg.pfS, cfg.peakFactorR = s.transS, s.transQ, s.pfL, s.pfS, s.peakFactorR
        cfg.seed = g("job_seed", s.pf_seed)     # one seed for the whole job (thx_refine_set_particles checks it over `world`)
        cfg.coreFSC, cfg.goldenAverage, cfg.solventFlatten = int(s.coreFSC), int(s.goldenAverage), int(s.solventFlatten)
        cfg.normCorrection = 1 if norm_correction else 0   # Optimiser::normCorrection: rescales shard.imgOri IN PLACE from iteration 2 on
        # ---- classes, global search, point group, CTF search ----
        cfg.nK = int(g("nK", 1))
        self.search = {"local": capi.SEARCH_LOCAL, "global": capi.SEARCH_GLOBAL, "ctf": capi.SEARCH_CTF}[g("search", "local")]
        cfg.searchType = self.search
        scan = g("scan")
        if scan is not None:
            cfg.nR, cfg.nT, cfg.rScan = len(scan["quat"]), len(scan["shifts"]), int(scan["rScan"])
            cfg.scanBatch = int(scan.get("batch", 0))
            mk, ms = scan_min_spread(scan.get("mS", cfg.nR), g("pfSGlobal", 0.5))
            cfg.scanMinK, cfg.scanMinS = scan.get("minK", mk), scan.get("minS", ms)
        cfg.pfSGlobal, cfg.peakFactorC = g("pfSGlobal", 0.5), g("peakFactorC", 1.0 - 1e-2)
        cfg.balanceClass = int(g("balanceClass", 0))
        self.sym = symmetry(g("sym")) if g("sym") else None
        if self.sym is not None and self.sym["n"] > 0:
            cfg.nSym, cfg.symMat, cfg.symQuat = self.sym["n"], self.sym["R"].ctypes.data, self.sym["quat"].ctypes.data
        cfg.mLD, cfg.ctfRefineS, cfg.pfSCTF = int(g("mLD", 0)), g("ctfRefineS", 0.01), g("pfSCTF", 0.5)
        assert s.use_pf, "the native driver runs the device particle filter"
        self.cfg = cfg
        h = C.c_void_p()
        capi.call("thx_refine_create", C.byref(h), C.byref(cfg), hemi.handle if hemi is not None else None,
                  world.handle if world is not None else None)
        self._h = h
        if g("img_base") is not None:            # where this rank's shard starts in the job's image numbering (refine.take_shard)
            capi.call("thx_refine_set_image_base", self._h, int(s.img_base))
        gid = np.ascontiguousarray(s.gid.astype(np.int32))
        capi.call("thx_refine_set_particles", self._h, ptr(s.imgOri), ptr(s.attr), gid.ctypes.data, ptr(s.pf0["r"]),
                  ptr(s.pf0["t"]), stream_ptr())
        refs = g("refs")
        capi.call("thx_refine_set_reference", self._h, ptr(refs if refs is not None else s.ref), stream_ptr())
        if scan is not None:
            q, sh = np.ascontiguousarray(scan["quat"], np.float64), np.ascontiguousarray(scan["shifts"], np.float64)
            capi.call("thx_refine_set_grid", self._h, q.ctypes.data, sh.ctypes.data, stream_ptr())
        if g("cls0") is not None:
            c0 = np.ascontiguousarray(g("cls0"), np.int32)
            capi.call("thx_refine_set_classes", self._h, c0.ctypes.data, stream_ptr())
        torch.cuda.synchronize()

    def set_search(self, search):
        self.search = {"local": capi.SEARCH_LOCAL, "global": capi.SEARCH_GLOBAL, "ctf": capi.SEARCH_CTF}[search]
        capi.call("thx_refine_set_search_type", self._h, self.search)

    def reset(self):
        capi.call("thx_refine_reset", self._h, stream_ptr())

    def set_cutoff(self, r, rU):
        """the frequency cut-offs of the next iteration (thx_refine_set_cutoff): r = Optimiser::_r, rU = Model::_rU"""
        capi.call("thx_refine_set_cutoff", self._h, int(r), int(rU), stream_ptr())

    def cutoff(self):
        """(r, rU, Reconstructor::_size, scan radius) as set"""
        v = [C.c_int(0) for _ in range(4)]
        capi.call("thx_refine_get_cutoff", self._h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def capture(self, maps=True, scan=False, sym=False):
        """per-phase trace of the local search (thx_refine_set_capture): returns the dict of device tensors the following
        iterations fill -- uR, uT, r, t, k123, s01 indexed [phase][image], mapsFsc [2][K][N]^3; scan: the scan's weights and the
        support points it left; sym: F / T after prepareTF"""
        s, c = self.shard, self.cfg
        n, dev, K = s.nImg, s.dev, c.nK
        nP = max(c.nPhase, c.maxPhase)     # (with the per-image stop rule the trace follows every image to its last phase)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        cap = dict(uR=z((nP, n, c.mLR), torch.float32), uT=z((nP, n, c.mLT), torch.float32),
                   r=z((nP, n, c.mLR, 4), torch.float64), t=z((nP, n, c.mLT, 2), torch.float64),
                   k123=z((nP, n, 3), torch.float64), s01=z((nP, n, 2), torch.float64),
                   mapsFsc=z((2, K, s.N, s.N, s.N), torch.float32) if maps else None,
                   rP=z((nP, n, c.mLR, 4), torch.float64), tP=z((nP, n, c.mLT, 2), torch.float64),
                   wRP=z((nP, n, c.mLR), torch.float64), wTP=z((nP, n, c.mLT), torch.float64))
        P, nV = s.N * s.pf, 2 if s.world == 1 else 1
        if maps:
            cap["Fraw"] = z((nV, K, P, P, P // 2 + 1), torch.complex64)
            cap["Traw"] = z((nV, K, P, P, P // 2 + 1), torch.float32)
        if sym:
            cap["Fsym"] = z((nV, K, P, P, P // 2 + 1), torch.complex64)
            cap["Tsym"] = z((nV, K, P, P, P // 2 + 1), torch.float32)
        if c.mLD > 0:
            cap.update(uD=z((nP, n, c.mLD), torch.float32), dP=z((nP, n, c.mLD), torch.float64), dR=z((nP, n, c.mLD), torch.float64))
        if scan:
            cap.update(scanUC=z((n, K), torch.float32), scanUR=z((n, K, c.nR), torch.float32), scanUT=z((n, K, c.nT), torch.float32),
                       r0=z((n, c.mLR, 4), torch.float64), t0=z((n, c.mLT, 2), torch.float64), k0=z((n, 3), torch.float64),
                       s0=z((n, 2), torch.float64))
        st = capi.RefineCapture()
        for k, v in cap.items():
            setattr(st, k, ptr(v) if v is not None else None)
        st.phases = nP
        capi.call("thx_refine_set_capture", self._h, C.byref(st))
        self._cap = cap
        return cap

    def iterate(self, timed=False):
        """one iteration; returns this iteration's FSC [N / 2] (one class) or [K][N / 2]"""
        K = self.cfg.nK
        fsc = np.zeros((K, self.shard.N // 2), np.float32)
        capi.call("thx_refine_iterate", self._h, fsc.ctypes.data, 1 if timed else 0, stream_ptr())
        return fsc[0] if K == 1 else fsc

    def run(self, steps, timed=False):
        fsc = None
        for _ in range(steps):
            fsc = self.iterate(timed)
        return fsc

    def stats(self, reset=False):
        st = RefineStats()
        capi.call("thx_refine_get_stats", self._h, C.byref(st), 1 if reset else 0)
        return st

    def rounds(self):
        """balancing rounds of the last iteration as [MAP off / on][local half][class]"""
        a = np.asarray(list(self.stats().lastRoundsK), np.int64).reshape(2, 2, 16)
        return a[:, :, :self.cfg.nK].copy()

    def view(self):
        v = RefineView()
        capi.call("thx_refine_get_view", self._h, C.byref(v))
        return v

    def map(self, half, k=0):
        N = self.shard.N
        m = torch.empty((N, N, N), dtype=torch.float32, device=self.shard.dev)
        capi.call("thx_refine_get_map_k", self._h, half, k, ptr(m), stream_ptr())
        return m

    def state(self):
        n, dev = self.shard.nImg, self.shard.dev
        off = torch.empty((n, 2), dtype=torch.float64, device=dev)
        topR = torch.empty((n, 4), dtype=torch.float64, device=dev)
        topT = torch.empty((n, 2), dtype=torch.float64, device=dev)
        capi.call("thx_refine_get_state", self._h, ptr(off), ptr(topR), ptr(topT), None, stream_ptr())
        return off, topR, topT

    def fetch(self, dev_ptr, dtype, shape, offset_elems=0):
        """host copy of a slice of one of view()'s device arrays"""
        a = np.empty(shape, dtype)
        torch.cuda.synchronize()
        capi.call("thx_memcpy_d2h", a.ctypes.data, int(dev_ptr) + offset_elems * a.itemsize, a.nbytes)
        return a

    def close(self):
        if self._h is not None:
            capi.call("thx_refine_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def symmetry(name, cap=128):
    """thx_symmetry_host = Symmetry::init(name): dict(n, R [n][9] column-major, quat [n][4], name)"""
    n = C.c_int(0)
    R, q = np.zeros((cap, 9)), np.zeros((cap, 4))
    capi.call("thx_symmetry_host", name.encode(), R.ctypes.data, q.ctypes.data, cap, C.byref(n))
    return dict(n=n.value, R=np.ascontiguousarray(R[:n.value]), quat=np.ascontiguousarray(q[:n.value]), name=name)


INIT_OUTSIDE_CONFIDENCE_AREA = 0.5   # include/Particle.h:59
TRANS_SEARCH_FACTOR = 0.25           # script/demo_3D.json "Translation Search Factor"


def scan_min_spread(mS, perturbFactorSGlobal=0.5):
    """the scanning phase's minimum spread (OPTIMISER_SCAN_SET_MIN_STD_WITH_PERTURB, src/Optimiser.cpp:667-690,1032-1079): scanMinStdR =
    mS^(-1/3) in MODE_3D (mS = "Number of Sampling Points for Scanning in Global Search": the scanned rotations BEFORE the symmetry
    reduction nR = mS / (1 + nSym), :652), scanMinStdT = 1 / Qinv(INIT_OUTSIDE_CONFIDENCE_AREA, 2) / sqrt(transSearchFactor pi), over perturbFactorSGlobal
    -> (minK, minS) of thx_pf_scan_support_dev"""
    minK = (mS ** (-1.0 / 3) / perturbFactorSGlobal) ** 2
    minS = 1.0 / (-2.0 * np.log(INIT_OUTSIDE_CONFIDENCE_AREA)) / np.sqrt(TRANS_SEARCH_FACTOR * np.pi) / perturbFactorSGlobal
    return float(minK), float(minS)
