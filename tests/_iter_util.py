"""Iteration-level parity: seeded inputs for one EM iteration, the oracle chain (oracle.Iteration) and the rule by which a
GPU test follows the device's discrete choices where they hinge on rounding (test infrastructure, numpy + oracle only).

The chain the bench times (thx_refine_iterate) makes three kinds of discrete decisions from floating-point weights: the
systematic resampling of the rotations and of the shifts after every phase, and the top support point.  The device's
likelihoods agree with the oracle's to ~1e-5 of max|L| (tests/test_parity_gpu.py), so a cumulative weight that lands
within that distance of a resampling threshold can fall on the other side.  TIE RULE: where the device's resampled indices
(or its top point) differ from the oracle's, they must be exactly what the ORACLE's resampler produces from the DEVICE's
weights, and those weights must agree with the oracle's within the weight bar; the oracle then continues from the device's
choice.  Everything else -- every weight of every phase, sigma tables, F / T, the four reconstructions, the FSC, the
refreshed projectors, offsets and re-masked images -- is compared value by value.
"""
import ctypes as C

import numpy as np
import scipy.fft as sfft

from thunder_amd import synth

import _philox as PH


def make_inputs(O, N, n, seed, mLR=125, mLT=9, nPhase=3, mReco=20, batch=64, nGroup=3, snr=0.05, rL=2, pixelSize=1.32):
    """n synthetic particles (SURVEY 8d recipe at a small size) + the configuration of one local-search iteration.
    Images: CTF x slice x ramp on the rL = 0 pixel list (+ the Hermitian mirror of the kx = 0 column) + white noise made in
    real space, so that every image is the FT of a real image.  numpy / oracle only: the same bytes reach the device and
    the oracle."""
    rng = np.random.default_rng(seed)
    pf, P, nc = 2, 2 * N, N // 2 + 1
    ref = synth.blob_map(N, seed=seed + 1, nblob=14)
    vol = O.set_projectee(ref, pf)
    plM = O.pixel_list(N, N // 2 - 2, 0, pf)
    quat = synth.random_quats(n, rng)
    shift = rng.normal(0, 2.0, size=(n, 2))
    attr = synth.ctf_params(n, rng)
    col0 = np.nonzero((plM["iCol"] == 0) & (plM["iRow"] > 0))[0]
    mirror_dst = (N - plM["iRow"][col0]) * nc
    imgOri = np.zeros((n, N, nc), np.complex64)
    sigs = []
    for l in range(n):
        s = O.project(vol, P, pf, O.rotate3D(quat[l]), plM["iCol"], plM["iRow"])
        s = s * O.ctf(pixelSize, *attr[l], N, plM["iCol"], plM["iRow"]) * O.translate(shift[l, 0], shift[l, 1], N, plM["iCol"], plM["iRow"])
        sigs.append(s.astype(np.complex64))
    sigma2 = float(np.mean(np.abs(np.stack(sigs)) ** 2)) / snr / 2.0
    for l in range(n):
        flat = imgOri[l].reshape(-1)
        flat[plM["iPxl"]] = sigs[l]
        flat[mirror_dst] = np.conj(sigs[l][col0])
        rl = rng.standard_normal((N, N)).astype(np.float32)
        imgOri[l] += (sfft.rfft2(rl) * np.float32(np.sqrt(2.0 * sigma2) / N)).astype(np.complex64)
    gid = rng.integers(1, nGroup + 1, n).astype(np.int32)
    gid[:nGroup] = np.arange(1, nGroup + 1)
    q0 = np.ascontiguousarray(synth.perturb_quats(quat, mLR, 0.03, rng))
    t0 = np.ascontiguousarray(shift[:, None, :] + rng.normal(0, 0.6, size=(n, mLT, 2)))
    cfg = dict(N=N, pf=pf, nImg=n, nHalfA=(n + 1) // 2, mLR=mLR, mLT=mLT, nPhase=nPhase, mReco=mReco, batch=batch, rL=rL,
               nGroup=nGroup, groupSig=1, pixelSize=pixelSize, maskRadiusPx=float(np.float32(0.45 * N)), sigma2Init=float(np.float32(sigma2)),
               transS=2.0, transQ=0.05, pfL=2.0, pfS=0.5, peakFactorR=1e-3, seed=1234567 + seed, coreFSC=1, goldenAverage=1,
               solventFlatten=1)
    return dict(cfg=cfg, imgOri=imgOri, attr=attr, gid=gid, quat0=q0, tran0=t0, ref=ref, quat=quat, shift=shift)


def oracle_chain(O, inp):
    return O.Iteration(inp["cfg"], inp["imgOri"], inp["attr"], inp["gid"], inp["quat0"], inp["tran0"], inp["ref"], PH)


def weight_bar(scaleL):
    """relative bar on exp(L - max L): an absolute error e in L is a relative error e in the weight (test_parity_gpu.py)"""
    return 3 * max(2e-5 * scaleL, 1e-4)


class Follower:
    """resolve-callback for oracle.Iteration.iterate: checks every phase's weights against the device's captured ones and
    applies the tie rule to the discrete decisions.  cap: dict of numpy arrays uR, uT [phase][image][.], r, t (after
    resampling), k123, s01 as thx_refine_set_capture fills them."""

    def __init__(self, O, cap, cfg):
        self.O, self.cap, self.c = O, cap, cfg
        self.n_checked = 0
        self.adopted = []          # (phase, image, what)
        self.max_rel = 0.0

    def _match(self, dev_rows, own_rows):
        """index of every device row among the oracle's rows (support points are distinct after a perturbation)"""
        d = np.abs(dev_rows[:, None, :] - own_rows[None, :, :]).max(axis=2)
        idx = d.argmin(axis=1)
        assert d[np.arange(len(idx)), idx].max() <= 1e-9, "a resampled support point of the device is not one of the oracle's"
        return idx

    def __call__(self, p, l, own):
        O, cap, c = self.O, self.cap, self.c
        uR, uT = cap["uR"][p, l], cap["uT"][p, l]
        # every weight of the phase (Particle::setUR / setUT inputs): the device's likelihood sums vs the oracle's
        bar = weight_bar(own.get("scaleL", 1.0))
        for dev, mine, name in ((uR, own["uR"], "uR"), (uT, own["uT"], "uT")):
            rel = np.abs(dev - mine) / np.maximum(np.abs(mine), 1e-30)
            big = mine > 1e-25 * mine.max()
            self.max_rel = max(self.max_rel, float(rel[big].max()))
            assert np.all(rel[big] <= bar), "phase %d image %d %s: %.3g > %.3g" % (p, l, name, rel[big].max(), bar)
        self.n_checked += 1
        # discrete decisions: resampled indices
        srcR = self._match(cap["r"][p, l], own["qPre"])
        srcT = self._match(cap["t"][p, l], own["tPre"])
        if not (np.array_equal(srcR, own["srcR"]) and np.array_equal(srcT, own["srcT"])):
            # tie rule: the oracle's resampler on the device's weights must give the device's indices
            seed, mLR, mLT = c["seed"], c["mLR"], c["mLT"]
            li, call = own["li"], own["callU"]
            alt = O.pf_update(own["qIn"], own["tPre"], own["wRIn"], own["wTIn"], uR, uT, c["peakFactorR"],
                              PH.shuffle_ranks(seed, li, call, 2, mLR), PH.draw_u4(seed, li, call, 3, 0)[0] / mLR,
                              PH.shuffle_ranks(seed, li, call, 4, mLT), PH.draw_u4(seed, li, call, 5, 0)[0] / mLT)
            assert np.array_equal(alt["srcR"], srcR) and np.array_equal(alt["srcT"], srcT), \
                "phase %d image %d: the device's resampling is not what its own weights give" % (p, l)
            self.adopted.append((p, l, "resample"))
            for k in ("q", "t", "wR", "wT", "srcR", "srcT", "topR", "topT", "iTopR", "iTopT"):
                own[k] = alt[k]
        return own


def fsc_curve(O, a, b, N, n):
    return O.fsc(sfft.rfftn(a).astype(np.complex64), sfft.rfftn(b).astype(np.complex64), N, n)


def as_struct_cfg(cfg, RefineConfig, half_of_rank=-1):
    c = RefineConfig()
    for k in ("N", "pf", "nImg", "nHalfA", "mLR", "mLT", "nPhase", "mReco", "batch", "rL", "nGroup", "groupSig", "pixelSize",
              "maskRadiusPx", "sigma2Init", "transS", "transQ", "pfL", "pfS", "peakFactorR", "seed", "coreFSC", "goldenAverage",
              "solventFlatten"):
        setattr(c, k, cfg[k])
    c.halfOfRank = half_of_rank
    c.maxPhase, c.pixelOrder, c.wgPerCU = 0, 1, 2
    return c
