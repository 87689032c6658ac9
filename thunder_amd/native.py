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
from .capi import ClassifyConfig, ClassifyStats, ClassifyView, RefineConfig, RefineStats, RefineView, ptr, stream_ptr

STAGES = ("rows", "expectation", "sigma", "insertion", "reconstruct", "recentre_remask", "norm_correction")


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
    """thx_refine handle over the particles of a RefineShard (which only has to have GENERATED them: allocate=False)."""

    def __init__(self, shard, hemi=None, world=None, pixel_order=1, max_phase=0, norm_correction=False):
        self.shard = shard
        s = shard
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
        cfg.seed = s.pf_seed
        cfg.coreFSC, cfg.goldenAverage, cfg.solventFlatten = int(s.coreFSC), int(s.goldenAverage), int(s.solventFlatten)
        cfg.normCorrection = 1 if norm_correction else 0   # Optimiser::normCorrection: rescales shard.imgOri IN PLACE from iteration 2 on
        assert s.use_pf, "the native driver runs the device particle filter"
        self.cfg = cfg
        h = C.c_void_p()
        capi.call("thx_refine_create", C.byref(h), C.byref(cfg), hemi.handle if hemi is not None else None,
                  world.handle if world is not None else None)
        self._h = h
        gid = np.ascontiguousarray(s.gid.astype(np.int32))
        capi.call("thx_refine_set_particles", self._h, ptr(s.imgOri), ptr(s.attr), gid.ctypes.data, ptr(s.pf0["r"]),
                  ptr(s.pf0["t"]), stream_ptr())
        capi.call("thx_refine_set_reference", self._h, ptr(s.ref), stream_ptr())
        torch.cuda.synchronize()

    def reset(self):
        capi.call("thx_refine_reset", self._h, stream_ptr())

    def capture(self, maps=True):
        """per-phase trace of the local search (thx_refine_set_capture): returns the dict of device tensors the following
        iterations fill -- uR, uT, r, t, k123, s01 indexed [phase][image], mapsFsc [2][N]^3"""
        s, c = self.shard, self.cfg
        n, dev = s.nImg, s.dev
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        cap = dict(uR=z((c.nPhase, n, c.mLR), torch.float32), uT=z((c.nPhase, n, c.mLT), torch.float32),
                   r=z((c.nPhase, n, c.mLR, 4), torch.float64), t=z((c.nPhase, n, c.mLT, 2), torch.float64),
                   k123=z((c.nPhase, n, 3), torch.float64), s01=z((c.nPhase, n, 2), torch.float64),
                   mapsFsc=z((2, s.N, s.N, s.N), torch.float32) if maps else None,
                   rP=z((c.nPhase, n, c.mLR, 4), torch.float64), tP=z((c.nPhase, n, c.mLT, 2), torch.float64),
                   wRP=z((c.nPhase, n, c.mLR), torch.float64), wTP=z((c.nPhase, n, c.mLT), torch.float64))
        if maps:
            P, nV = s.N * s.pf, 2 if s.world == 1 else 1
            cap["Fraw"] = z((nV, P, P, P // 2 + 1), torch.complex64)
            cap["Traw"] = z((nV, P, P, P // 2 + 1), torch.float32)
        st = capi.RefineCapture()
        for k, v in cap.items():
            setattr(st, k, ptr(v) if v is not None else None)
        capi.call("thx_refine_set_capture", self._h, C.byref(st))
        self._cap = cap
        return cap

    def iterate(self, timed=False):
        fsc = np.zeros(self.shard.N // 2, np.float32)
        capi.call("thx_refine_iterate", self._h, fsc.ctypes.data, 1 if timed else 0, stream_ptr())
        return fsc

    def run(self, steps, timed=False):
        fsc = None
        for _ in range(steps):
            fsc = self.iterate(timed)
        return fsc

    def stats(self, reset=False):
        st = RefineStats()
        capi.call("thx_refine_get_stats", self._h, C.byref(st), 1 if reset else 0)
        return st

    def view(self):
        v = RefineView()
        capi.call("thx_refine_get_view", self._h, C.byref(v))
        return v

    def map(self, half):
        N = self.shard.N
        m = torch.empty((N, N, N), dtype=torch.float32, device=self.shard.dev)
        capi.call("thx_refine_get_map", self._h, half, ptr(m), stream_ptr())
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


CLASSIFY_STAGES = ("scan", "class_select_and_support_points", "local_phases", "insertion", "reconstruct")
INIT_OUTSIDE_CONFIDENCE_AREA = 0.5   # include/Particle.h:59
TRANS_SEARCH_FACTOR = 0.25           # script/demo_3D.json "Translation Search Factor"


def scan_min_spread(nR, perturbFactorSGlobal=0.5):
    """the scanning phase's minimum spread (OPTIMISER_SCAN_SET_MIN_STD_WITH_PERTURB, src/Optimiser.cpp:667-690,1032-1079): scanMinStdR =
    mS^(-1/3) in MODE_3D (mS = the scanned rotations before symmetry reduction: nR for C1, which is what this takes), scanMinStdT = 1 / Qinv(INIT_OUTSIDE_CONFIDENCE_AREA, 2) / sqrt(transSearchFactor pi), over perturbFactorSGlobal
    -> (minK, minS) of thx_pf_scan_support_dev"""
    minK = (nR ** (-1.0 / 3) / perturbFactorSGlobal) ** 2
    minS = 1.0 / (-2.0 * np.log(INIT_OUTSIDE_CONFIDENCE_AREA)) / np.sqrt(TRANS_SEARCH_FACTOR * np.pi) / perturbFactorSGlobal
    return float(minK), float(minS)


class NativeClassify:
    """thx_classify handle (thunder_amd/csrc/thx_classify.hip): one K-class classification iteration -- global scan, class of
    every image, support points, local phases against the assigned reference, multi-reference insertion session, 2
    reconstructions per class -- behind one call.  The tensors handed to set_particles are borrowed: keep them alive."""

    def __init__(self, N, K, nImg, nR, nT, rScan, rL=2, pf=2, mLR=125, mLT=9, nPhase=3, mReco=100, batch=10240, pixel_order=0,
                 wg_per_cu=2, refresh=False, pixelSize=1.32, transS=2.0, transQ=0.05, pfL=2.0, pfS=0.5, peakFactorR=1e-3,
                 peakFactorC=1.0 - 1e-2, seed=20240607, hemi=None, nImgHemi=0, scan_min=None):
        cfg = ClassifyConfig()
        cfg.N, cfg.pf, cfg.nK, cfg.nImg, cfg.nImgHemi = N, pf, K, nImg, nImgHemi
        cfg.nR, cfg.nT, cfg.rScan, cfg.rL = nR, nT, rScan, rL
        cfg.mLR, cfg.mLT, cfg.nPhase, cfg.mReco, cfg.batch = mLR, mLT, nPhase, mReco, batch
        cfg.pixelOrder, cfg.wgPerCU, cfg.refresh, cfg.pixelSize = pixel_order, wg_per_cu, 1 if refresh else 0, pixelSize
        cfg.transS, cfg.transQ, cfg.pfL, cfg.pfS = transS, transQ, pfL, pfS
        cfg.peakFactorR, cfg.peakFactorC = peakFactorR, peakFactorC
        cfg.scanMinK, cfg.scanMinS = scan_min_spread(nR) if scan_min is None else scan_min   # (minK, minS) of thx_pf_scan_support_dev
        cfg.seed = seed
        self.cfg = cfg
        h = C.c_void_p()
        capi.call("thx_classify_create", C.byref(h), C.byref(cfg), hemi.handle if hemi is not None else None)
        self._h = h
        self._keep = []

    def set_grid(self, quat, shifts):
        capi.call("thx_classify_set_grid", self._h, ptr(quat), ptr(shifts), stream_ptr())

    def set_particles(self, datM, ctfM, sigRcpM, w):
        assert datM.dtype == torch.complex64 and ctfM.dtype == torch.float32 and sigRcpM.dtype == torch.float32 and w.dtype == torch.float32
        self._keep = [datM, ctfM, sigRcpM, w]
        capi.call("thx_classify_set_particles", self._h, ptr(datM), ptr(ctfM), ptr(sigRcpM), ptr(w), stream_ptr())

    def set_references(self, refRL):
        assert refRL.dtype == torch.float32 and refRL.shape[0] == self.cfg.nK
        capi.call("thx_classify_set_references", self._h, ptr(refRL), stream_ptr())

    def set_fsc(self, fsc):
        f = np.ascontiguousarray(np.asarray(fsc, np.float32))
        capi.call("thx_classify_set_fsc", self._h, f.ctypes.data, f.size)

    def capture(self):
        """stage trace (thx_classify_set_capture): dict of device tensors the following iterations fill -- r0, t0 (support points
        after the scan), Fraw, Traw (accumulators before prepareTF)"""
        c = self.cfg
        P, dev = c.N * c.pf, torch.device("cuda", torch.cuda.current_device())
        cap = dict(r0=torch.zeros((c.nImg, c.mLR, 4), dtype=torch.float64, device=dev),
                   t0=torch.zeros((c.nImg, c.mLT, 2), dtype=torch.float64, device=dev),
                   Fraw=torch.zeros((c.nK, P, P, P // 2 + 1), dtype=torch.complex64, device=dev),
                   Traw=torch.zeros((c.nK, P, P, P // 2 + 1), dtype=torch.float32, device=dev))
        st = capi.ClassifyCapture()
        for k, v in cap.items():
            setattr(st, k, ptr(v))
        capi.call("thx_classify_set_capture", self._h, C.byref(st))
        self._cap = cap
        return cap

    def iterate(self, timed=False):
        capi.call("thx_classify_iterate", self._h, 1 if timed else 0, stream_ptr())

    def stats(self, reset=False):
        st = ClassifyStats()
        capi.call("thx_classify_get_stats", self._h, C.byref(st), 1 if reset else 0)
        return st

    def view(self):
        v = ClassifyView()
        capi.call("thx_classify_get_view", self._h, C.byref(v))
        return v

    def fetch(self, dev_ptr, dtype, shape):
        """host copy of one of view()'s device arrays"""
        a = np.empty(shape, dtype)
        torch.cuda.synchronize()
        capi.call("thx_memcpy_d2h", a.ctypes.data, int(dev_ptr), a.nbytes)
        return a

    def close(self):
        if self._h is not None:
            capi.call("thx_classify_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
