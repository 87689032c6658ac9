"""Record / replay of what the chain tests read from the native driver (tests/test_iteration_gpu.py).

The device's run does not depend on the oracle: with THX_CHAIN_DUMP=<dir> every iteration's state -- the capture buffers, the
arrays behind thx_refine_get_view the checks fetch, maps, rounds, statistics -- is written to <dir>/<case>.npz on the GPU box, and
tools/probes/replay_chain.py runs the SAME checks against such a file on a machine without a GPU.  Test infrastructure: it
exists so that a failing comparison can be taken apart where the oracle runs, without another GPU call."""
import os
import types

import numpy as np

DUMP_DIR = os.environ.get("THX_CHAIN_DUMP")
VIEW_ARRAYS = ("cls", "r", "t", "norm", "sig", "vols", "img", "k123", "F", "T", "d", "nP")
STAT_FIELDS = ("balancingRounds", "normRadius", "normMedian", "imagePhases")


def _shapes(c, K):
    N, n, P = c["N"], c["nImg"], 2 * c["N"]
    return dict(cls=(np.int32, (n,)), r=(np.float64, (n, c["mLR"], 4)), t=(np.float64, (n, c["mLT"], 2)), norm=(np.float32, (n,)),
                sig=(np.float32, (2, c["nGroup"], N // 2 - 1)), vols=(np.complex64, (2 * K, P, P, P // 2 + 1)),
                img=(np.complex64, (n, N, N // 2 + 1)), k123=(np.float64, (n, 3)), F=(np.complex64, (2 * K, P, P, P // 2 + 1)),
                T=(np.float32, (2 * K, P, P, P // 2 + 1)), d=(np.float64, (n, max(c.get("mLD", 0), 1))), nP=(np.int32, (n,)))


class Recorder:
    def __init__(self, case, c):
        self.path, self.c, self.K, self.data, self.step = os.path.join(DUMP_DIR, case + ".npz"), c, c["nK"], {}, 0
        os.makedirs(DUMP_DIR, exist_ok=True)

    def record(self, nat, cap, fsc=None, imgOri_before=None):
        """state after reset (step 0) or after an iteration (step 1, 2, ...)"""
        import torch
        torch.cuda.synchronize()
        d, p = self.data, "s%d/" % self.step
        v = nat.view()
        for name, (dt, shape) in _shapes(self.c, self.K).items():
            if getattr(v, name) and not (self.step == 0 and name in ("F", "T")):      # (norm: only with normCorrection)
                d[p + "view/" + name] = nat.fetch(getattr(v, name), dt, shape)
        d[p + "nPxl"] = np.asarray([v.nPxl, v.nPxlM])
        if self.step > 0:
            for k, t in cap.items():
                if t is not None:
                    d[p + "cap/" + k] = t.cpu().numpy()
            d[p + "fsc"] = np.asarray(fsc)
            d[p + "rounds"] = nat.rounds()
            st = nat.stats()
            d[p + "stats"] = np.asarray([float(getattr(st, f)) for f in STAT_FIELDS])
            for i, x in enumerate(nat.state()):
                d[p + "state%d" % i] = x.cpu().numpy()
            for h in (0, 1):
                for k in range(self.K):
                    d[p + "map%d_%d" % (h, k)] = nat.map(h, k).cpu().numpy()
            d[p + "imgOri"] = nat.shard.imgOri.cpu().numpy()
            if imgOri_before is not None:
                d[p + "imgOri_before"] = imgOri_before
        self.step += 1
        np.savez_compressed(self.path, **d)


class _T:
    """numpy array with the two tensor methods the checks call"""
    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self

    def numpy(self):
        return self.a


class _Cap(dict):
    def __init__(self, rep):
        super().__init__()
        self.rep = rep

    def items(self):
        p = "s%d/cap/" % self.rep.step
        return [(k[len(p):], _T(self.rep.d[k])) for k in self.rep.d.files if k.startswith(p)]


class ReplayNative:
    """stands in for thunder_amd.native.NativeRefine in _check_iteration / _run_chain: serves a Recorder's file"""
    def __init__(self, path, c):
        self.d, self.c, self.K, self.step = np.load(path), c, c["nK"], 0
        self.cfg = types.SimpleNamespace(mLD=c.get("mLD", 0), nK=c["nK"])
        self.shard = types.SimpleNamespace()
        self._ori()

    def _ori(self):
        # (before an iterate() the checks read the stack the NEXT iteration starts from)
        nxt, cur = "s%d/imgOri_before" % (self.step + 1), "s%d/imgOri" % self.step
        a = self.d[nxt] if nxt in self.d.files else (self.d[cur] if cur in self.d.files else None)
        self.shard.imgOri = _T(a)

    def set_search(self, search):
        pass

    def reset(self):
        pass

    def capture(self, **kw):
        return _Cap(self)

    def iterate(self):
        self.step += 1
        fsc = self.d["s%d/fsc" % self.step]
        self.shard.imgOri = _T(self.d["s%d/imgOri" % self.step])
        return fsc[0] if self.K == 1 and fsc.ndim == 2 else fsc

    def after_iterate(self):
        self._ori()

    def stats(self, reset=False):
        key = "s%d/stats" % self.step
        vals = self.d[key] if key in self.d.files else np.zeros(len(STAT_FIELDS))
        return types.SimpleNamespace(**dict(zip(STAT_FIELDS, vals)))

    def rounds(self):
        return self.d["s%d/rounds" % self.step]

    def view(self):
        nP = self.d["s%d/nPxl" % self.step]
        return types.SimpleNamespace(nPxl=int(nP[0]), nPxlM=int(nP[1]), fdim=2 * self.c["N"], **{k: k for k in VIEW_ARRAYS})

    def fetch(self, name, dtype, shape, offset_elems=0):
        a = self.d["s%d/view/%s" % (self.step, name)].reshape(-1)
        n = int(np.prod(shape))
        return a[offset_elems:offset_elems + n].reshape(shape).astype(dtype, copy=True)

    def map(self, half, k=0):
        return _T(self.d["s%d/map%d_%d" % (self.step, half, k)])

    def state(self):
        return tuple(_T(self.d["s%d/state%d" % (self.step, i)]) for i in range(3))
