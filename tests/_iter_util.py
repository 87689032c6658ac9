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


def make_inputs(O, N, n, seed, mLR=125, mLT=9, nPhase=3, mReco=20, batch=64, nGroup=3, snr=0.05, rL=2, pixelSize=1.32, q_spread=0.03,
                t_spread=0.6, sigma_scale=1.0, K=1, sym=None, scan=None, norm_correction=0, balance=0, amp_spread=0.0, max_phase=0):
    """n synthetic particles (SURVEY 8d recipe at a small size) + the configuration of one iteration.
    Images: CTF x slice x ramp on the rL = 0 pixel list (+ the Hermitian mirror of the kx = 0 column) + white noise made in
    real space, so that every image is the FT of a real image.  numpy / oracle only: the same bytes reach the device and
    the oracle.  K references (class of a particle drawn uniformly), sym = a point-group name (the references carry it),
    scan = dict(nR, nT, rScan[, mS]): a scanned grid of nR random rotations and nT shifts, the particles' true poses are grid points
    (global search); amp_spread: per-image amplitude factors exp(N(0, amp_spread)) (what normCorrection is there to undo)."""
    rng = np.random.default_rng(seed)
    pf, P, nc = 2, 2 * N, N // 2 + 1
    symd = O.symmetry(sym) if sym else None
    refs = np.stack([synth.blob_map(N, seed=seed + 1 + 10 * k, nblob=14, symR=None if symd is None else symd["R"]) for k in range(K)])
    vols = [O.set_projectee(refs[k], pf) for k in range(K)]
    plM = O.pixel_list(N, N // 2 - 2, 0, pf)
    cls_true = rng.integers(0, K, n).astype(np.int32) if K > 1 else np.zeros(n, np.int32)
    grid = None
    if scan is not None:
        gq = synth.random_quats(scan["nR"], rng)
        gt = np.ascontiguousarray(rng.normal(0, 1.5, size=(scan["nT"], 2)))
        grid = (gq, gt)
        r_true, t_true = rng.integers(0, scan["nR"], n), rng.integers(0, scan["nT"], n)
        quat, shift = gq[r_true].copy(), gt[t_true].copy()
    else:
        quat = synth.random_quats(n, rng)
        shift = rng.normal(0, 2.0, size=(n, 2))
    attr = synth.ctf_params(n, rng)
    col0 = np.nonzero((plM["iCol"] == 0) & (plM["iRow"] > 0))[0]
    mirror_dst = (N - plM["iRow"][col0]) * nc
    imgOri = np.zeros((n, N, nc), np.complex64)
    sigs = []
    for l in range(n):
        s = O.project(vols[cls_true[l]], P, pf, O.rotate3D(quat[l]), plM["iCol"], plM["iRow"])
        s = s * O.ctf(pixelSize, *attr[l], N, plM["iCol"], plM["iRow"]) * O.translate(shift[l, 0], shift[l, 1], N, plM["iCol"], plM["iRow"])
        sigs.append(s.astype(np.complex64))
    sigma2 = float(np.mean(np.abs(np.stack(sigs)) ** 2)) / snr / 2.0
    amp = np.exp(rng.normal(0, amp_spread, n)).astype(np.float32) if amp_spread > 0 else np.ones(n, np.float32)
    for l in range(n):
        flat = imgOri[l].reshape(-1)
        flat[plM["iPxl"]] = sigs[l]
        flat[mirror_dst] = np.conj(sigs[l][col0])
        rl = rng.standard_normal((N, N)).astype(np.float32)
        imgOri[l] += (sfft.rfft2(rl) * np.float32(np.sqrt(2.0 * sigma2) / N)).astype(np.complex64)
        imgOri[l] *= amp[l]
    gid = rng.integers(1, nGroup + 1, n).astype(np.int32)
    gid[:nGroup] = np.arange(1, nGroup + 1)
    q0 = np.ascontiguousarray(synth.perturb_quats(quat, mLR, q_spread, rng))
    t0 = np.ascontiguousarray(shift[:, None, :] + rng.normal(0, t_spread, size=(n, mLT, 2)))
    cfg = dict(N=N, pf=pf, nImg=n, nHalfA=(n + 1) // 2, mLR=mLR, mLT=mLT, nPhase=nPhase, mReco=mReco, batch=batch, rL=rL,
               nGroup=nGroup, groupSig=1, pixelSize=pixelSize, maskRadiusPx=float(np.float32(0.45 * N)), sigma2Init=float(np.float32(sigma2 * sigma_scale)),
               transS=2.0, transQ=0.05, pfL=2.0, pfS=0.5, peakFactorR=1e-3, seed=1234567 + seed, coreFSC=1, goldenAverage=1,
               solventFlatten=1, normCorrection=int(norm_correction), nK=K, sym=symd, symName=sym, balanceClass=int(balance),
               pfSGlobal=0.5, peakFactorC=1.0 - 1e-2, maxPhase=int(max_phase))
    if scan is not None:
        from thunder_amd.native import scan_min_spread
        mk, ms = scan_min_spread(scan.get("mS", scan["nR"] * (1 + (symd["n"] if symd else 0))), 0.5)
        cfg.update(rScan=int(scan["rScan"]), scanMinK=mk, scanMinS=ms, nR=scan["nR"], nT=scan["nT"])
    return dict(cfg=cfg, imgOri=imgOri, attr=attr, gid=gid, quat0=q0, tran0=t0, ref=refs[0] if K == 1 else refs, refs=refs, quat=quat,
                shift=shift, cls_true=cls_true, grid=grid, amp=amp)


def oracle_chain(O, inp, cls0=None):
    return O.Iteration(inp["cfg"], inp["imgOri"], inp["attr"], inp["gid"], inp["quat0"], inp["tran0"], inp["refs"], PH, grid=inp.get("grid"),
                       cls0=cls0)


def weight_bar(scaleL):
    """relative bar on exp(L - max L): an absolute error e in L is a relative error e in the weight (test_parity_gpu.py)"""
    return 3 * max(2e-5 * scaleL, 1e-4)


class Follower:
    """resolve-callback for oracle.Iteration.iterate: checks every phase's weights against the device's captured ones and
    applies the tie rule to the discrete decisions.  cap: dict of numpy arrays uR, uT [phase][image][.], r, t (after
    resampling), k123, s01 as thx_refine_set_capture fills them."""

    def __init__(self, O, cap, cfg):
        self.O, self.cap, self.c = O, cap, cfg
        self.symQ = None if not cfg.get("sym") or cfg["sym"]["n"] == 0 else cfg["sym"]["quat"]
        self.n_scan = 0
        self.scan_adopted = []     # (image, what)
        self.q_pure, self.n_counterpart = None, 0
        self.scan_collapsed = []   # (image, distinct support points): spread of a collapsed support set taken from the device
        self.n_checked = 0
        self.adopted = []          # (phase, image, what)
        self.degenerate = []       # (phase, image, distinct incoming rotations, size of the difference)
        self.min_distinct = 8
        self.max_mean_angle = 1e-2
        self.mean_angles = []
        self.prior_err = []
        self.k_in = {}
        self.max_rel = 0.0

    def _match(self, dev_rows, own_rows):
        """index of every device row among the oracle's rows (support points are distinct after a perturbation)"""
        d = np.abs(dev_rows[:, None, :] - own_rows[None, :, :]).max(axis=2)
        idx = d.argmin(axis=1)
        # the perturbation is conjugated by the cloud's mean, which inherits the accuracy of the ACG estimate (the cofactor
        # inverse of a matrix of condition ~1e4, wave-tree vs serial sums: 1e-9, tests/test_pf_gpu.py) -- 1e-7 is far below the
        # spacing of the support points (~1e-3)
        assert d[np.arange(len(idx)), idx].max() <= 1e-7, "a resampled support point of the device is not one of the oracle's"
        return idx

    @staticmethod
    def _on_threshold(src, w, u, rank, u0, tau=1e-5):
        """systematic resampling (src/Particle.cpp:1340-1372): output j takes the first shuffled element whose cumulative
        weight reaches u0 + j / n; True when every device index satisfies that to within tau"""
        n = len(w)
        inv = np.empty(n, np.int64)
        inv[np.asarray(rank)] = np.arange(n)
        ws, us = np.asarray(w, np.float64)[inv], np.asarray(u, np.float64)[inv]
        cdf = np.cumsum(ws * us / (ws * us).sum())
        cdf /= cdf[-1]
        pos = np.asarray(rank)[np.asarray(src)]              # shuffled position of every chosen element
        uj = u0 + np.arange(n) / n
        lower = np.where(pos > 0, cdf[np.maximum(pos - 1, 0)], -1.0)
        return bool(np.all(uj <= cdf[pos] + tau) and np.all(uj > lower - tau))

    # ---- global search: the scan's weights, the class and the support points it leaves ----
    def scan_weights(self, l, own):
        """every class / rotation / shift weight of the scan (src/Optimiser.cpp:834-894) against the device's, at the bar of
        test_expect_global; the filter then continues from the DEVICE's weights (keepHalfHeightPeak and the systematic resampling of
        10 000 points turn a last-digit difference into another draw)"""
        cap = self.cap
        tol = max(6e-5 * abs(own["base"]), 3e-4)
        for name, dev in (("uC", cap["scanUC"][l]), ("uR", cap["scanUR"][l]), ("uT", cap["scanUT"][l])):
            np.testing.assert_allclose(dev, own[name], rtol=tol, atol=1e-30, err_msg="image %d scan %s" % (l, name))
        self.n_scan += 1
        return dict(own, uC=cap["scanUC"][l], uR=cap["scanUR"][l], uT=cap["scanUT"][l])

    def after_scan(self, l, ws):
        """class and support points of image l as pf_class_select / pf_scan_support give them from the device's scan weights and
        the replayed draws: the class must be equal; a support point may be the neighbour in the shuffled order where the draw sits on
        its threshold (the cumulative sum of 10 000 weights; rule of test_scan_support_points)"""
        cap = self.cap
        assert int(cap["cls"][l]) == int(ws["cls"]), "image %d: class %d, oracle %d" % (l, cap["cls"][l], ws["cls"])
        r0, t0 = cap["r0"][l], cap["t0"][l]
        d = np.abs(r0 - ws["q"]).max(axis=1)
        bad = np.nonzero(d > 1e-12)[0]
        if len(bad):
            # (with a point group the device's point is the counterpart of ITS source next to the same anchor: compare sources)
            assert len(bad) <= 2, "image %d: %d support points differ after the scan" % (l, len(bad))
            self.scan_adopted.append((l, "support (%d on a threshold)" % len(bad)))
            ws = dict(ws, q=r0.copy(), k=cap["k0"][l].copy(), topR=ws["topR"])
        assert np.abs(t0 - ws["t"]).max() <= 1e-12, "image %d: support shifts differ after the scan" % l
        # calVari of the new support set: mLR draws from a few of the scanned rotations -- many copies of few points, the regime in
        # which the ACG fixed point is ill-conditioned (tests/test_pf_gpu.py holds it to 5e-2 where the two sides stop in different
        # rounds); the filter continues from the device's spread
        # rounds); with fewer than `min_distinct` distinct points (or one point holding > 15 % of the copies: the collapsed-cloud
        # condition of the mean-frame rule below) the 4 x 4 scatter matrix is near singular, its cofactor inverse is rounding noise on
        # both sides and only the floor max(k, scanMinStdR) is common -- there the spread is required to be finite and at or above
        # the floor, and is counted.  The filter continues from the device's spread either way.
        _, mult = np.unique(np.round(ws["q"], 9), axis=0, return_counts=True)
        if len(mult) < self.min_distinct or mult.max() > 0.15 * len(ws["q"]):
            assert np.all(np.isfinite(cap["k0"][l])) and np.all(cap["k0"][l] >= self.c["scanMinK"] * (1 - 1e-12)), "image %d: spread after the scan" % l
            if not np.allclose(cap["k0"][l], ws["k"], rtol=5e-2):
                self.scan_collapsed.append((l, len(mult)))
        else:
            np.testing.assert_allclose(cap["k0"][l], ws["k"], rtol=5e-2, err_msg="image %d, %d distinct support points, largest multiplicity %d" % (l, len(mult), mult.max()))
        np.testing.assert_allclose(cap["s0"][l], ws["s"], rtol=1e-10)
        return dict(ws, k=cap["k0"][l].copy())

    def after_perturb(self, p, l, q_in, q, t, wR, wT):
        """Particle::perturb conjugates every perturbation by mean = inferACG(mean, _r) of the cloud as resampling left it:
        r_i <- mean * pert_i * conj(mean) * r_i (src/Particle.cpp:1203-1239).  `mean` is the top eigenvector of the last-but-one
        iterate of a fixed point that inverts, every round, a 4 x 4 scatter matrix by cofactors (dmat44::inverse(),
        src/Geometry/DirectionalStat.cpp:93-145).  After resampling the cloud holds many copies of a few tens of points and
        that matrix has condition 1e5 - 1e7: the inverse carries errors of cond^2 * eps and `mean` is determined to 1e-7 ... 1e-3
        rad only -- in the reference as much as here (a cloud collapsed onto < 4 points makes the matrix singular).
        MEAN-FRAME RULE: the device's perturbed cloud must be EXACTLY the oracle's perturbations conjugated by a mean m' that
        lies within `max_mean_angle` of the oracle's: vec(q'_i conj(r_i)) = R vec(q_i conj(r_i)) for ONE rotation R (Kabsch
        fit, residual <= 2e-4) of angle <= 2 max_mean_angle; only a cloud with fewer than `min_distinct` distinct incoming
        rotations may exceed the angle.  The oracle then continues from the device's cloud (priors from the oracle's own
        balanceWeight), so that every later stage is compared on identical inputs."""
        from thunder_amd import synth
        qd, td = self.cap["rP"][p, l], self.cap["tP"][p, l]
        assert np.abs(td - t).max() <= 1e-9, "phase %d image %d: perturbed shifts differ" % (p, l)
        conj = q_in * np.array([1.0, -1, -1, -1])
        qd_dev = qd
        if self.symQ is not None:
            # Particle::perturb ends with symmetrise(&mean) (src/Particle.cpp:1234): every perturbed point is replaced by its symmetry
            # mate nearest the cloud's mean -- and where the mean itself is numerically undetermined (collapsed clouds, below) the device
            # and the oracle can settle on different mates of the same pose.  COUNTERPART RULE: the comparison is made on the
            # perturbations BEFORE that step: the oracle's are known (q_pure: the same call without the point group); of the
            # device's stored point the mate is taken whose rotation from the incoming point has the oracle's angle (a large
            # heavy-tail perturbation can lie nearer another mate's, so "nearest to the incoming point" would not do) -- the Kabsch fit
            # below then holds all of them to ONE frame rotation -- and the stored point must be that mate's counterpart next to
            # the device's mean.  The oracle continues from the device's own cloud.
            q_sym, q = q, self.q_pure
            conjs = np.concatenate([[[1.0, 0, 0, 0]], self.symQ * np.array([1.0, -1, -1, -1])])
            mates = lambda x: np.stack([synth.quat_mul(g[None], x) for g in conjs])                    # [1 + nSym][n][4]
            assert np.abs(np.abs(np.einsum("cni,ni->cn", mates(q), q_sym)).max(axis=0) - 1).max() <= 1e-9   # (the oracle's own step)
            cands = mates(qd)
            po0 = np.abs(synth.quat_mul(q, conj)[:, 0])
            pd0 = np.abs(np.stack([synth.quat_mul(cd, conj)[:, 0] for cd in cands]))
            best = np.abs(pd0 - po0[None]).argmin(axis=0)
            qd = cands[best, np.arange(len(qd))]
            sg = np.sign(synth.quat_mul(qd, conj)[:, 0] * synth.quat_mul(q, conj)[:, 0])
            qd = qd * np.where(sg == 0, 1.0, sg)[:, None]
        po, pd = synth.quat_mul(q, conj), synth.quat_mul(qd, conj)          # mean * pert * conj(mean), both ways
        # (k1..k3 themselves carry the 2e-6 relative accuracy of the ACG estimate, so the perturbations agree to ~1e-6 |pert|)
        # k1..k3 carry the accuracy of the ACG estimate they come from (cofactor inverse at condition ~1e5: 1e-6 ... 1e-3
        # relative), and a heavy-tail draw (|g0| << 1) turns that into up to ~1e-4 of a large perturbation
        assert np.abs(po[:, 0] - pd[:, 0]).max() <= 2e-4, "phase %d image %d: the perturbation angles differ" % (p, l)
        Vo, Vd = po[:, 1:], pd[:, 1:]
        U_, _, Vt = np.linalg.svd(Vo.T @ Vd)
        dsgn = np.sign(np.linalg.det(U_ @ Vt))
        R = (U_ @ np.diag([1, 1, dsgn]) @ Vt).T                              # Vd ~ Vo R^T
        resid = np.abs(Vd - Vo @ R.T).max()
        ang = float(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))) / 2   # angle between the two means (quaternion half-angle)
        nd = len(np.unique(np.round(q_in, 12), axis=0))
        assert resid <= 2e-4, "phase %d image %d: not the same perturbations in another frame (residual %.2g)" % (p, l, resid)
        if self.symQ is not None:
            # the device's mean = the frame rotation applied to the oracle's mean; its stored point must have the largest |<., mean>|
            # among the mates (the reference compares in RFLOAT with a strict >: ties within 1e-6 go either way)
            A, m = np.zeros(16), np.zeros(4)
            L = self.O.lib()
            L.orc_infer_acg(A.ctypes.data_as(C.POINTER(C.c_double)), np.ascontiguousarray(q_in).ctypes.data_as(C.POINTER(C.c_double)), C.c_int(len(q_in)))
            L.orc_sym4_top_eigvec(m.ctypes.data_as(C.POINTER(C.c_double)), A.ctypes.data_as(C.POINTER(C.c_double)))
            w_, V_ = np.linalg.eig(R)
            ax = np.real(V_[:, np.argmin(np.abs(w_ - 1))])
            sn = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) @ ax
            th = np.arctan2(sn / 2, (np.trace(R) - 1) / 2)
            rq = np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * ax])
            md = synth.quat_mul(rq[None], m[None])[0]
            dots = np.abs(np.einsum("cni,i->cn", mates(qd_dev), md))
            short = dots.max(axis=0) - dots[0]
            self.n_counterpart += int(np.count_nonzero(short > 1e-6))
            if ang <= self.max_mean_angle:
                # (md carries the accuracy of the fit: 2e-4)
                assert short.max() <= 2e-3, "phase %d image %d: a stored point is not the counterpart next to the mean (%.2g)" % (p, l, short.max())
        _, mult = np.unique(np.round(q_in, 12), axis=0, return_counts=True)
        self.mean_angles.append((ang, nd, int(mult.max())))
        if ang > self.max_mean_angle:
            # Tyler's fixed point exists only while no support point holds a quarter of the cloud (and no pair a half): beyond
            # that the scatter matrix runs towards a singular one and its cofactor inverse decides the result
            assert nd < self.min_distinct or mult.max() > 0.15 * len(q_in), \
                "phase %d image %d: mean frames %.2g rad apart with %d distinct incoming points, largest multiplicity %d" % (p, l, ang, nd, mult.max())
            self.degenerate.append((p, l, nd, ang))
        # priors of the perturbed cloud: Particle::balanceWeight, w_i = 1 / pdfACG(r_i, A) -- A from the same kind of fixed point
        w = np.zeros(len(qd))
        self.O.lib().orc_balance_weight_R(w.ctypes.data_as(C.POINTER(C.c_double)),
                                          np.ascontiguousarray(qd_dev).ctypes.data_as(C.POINTER(C.c_double)), C.c_int(len(qd)))
        wd, wtd = self.cap["wRP"][p, l], self.cap["wTP"][p, l]
        eR, eT = float(np.abs(wd / w - 1).max()), float(np.abs(wtd / wT - 1).max())
        self.prior_err.append(eR)
        assert eT <= 1e-9, "phase %d image %d: shift priors differ by %.2g" % (p, l, eT)
        if eR > 1e-3:   # the perturbed copies of a collapsed cloud form tight clusters: the same ill-conditioned estimate
            assert nd < self.min_distinct or mult.max() > 0.15 * len(q_in), \
                "phase %d image %d: rotation priors differ by %.2g with %d distinct incoming points" % (p, l, eR, nd)
            if not (self.degenerate and self.degenerate[-1][:2] == (p, l)):
                self.degenerate.append((p, l, nd, ang))
        return qd_dev.copy(), td.copy(), wd.copy(), wtd.copy()

    def __call__(self, p, l, own):
        O, cap, c = self.O, self.cap, self.c
        uR, uT = cap["uR"][p, l], cap["uT"][p, l]
        # every weight of the phase (Particle::setUR / setUT inputs): the device's likelihood sums vs the oracle's
        # bar: the weight bar of tests/test_parity_gpu.py (relative, from the 1e-5 max|L| bar on the log-likelihoods) plus 1e-6
        # of the image's largest weight -- the reference accumulates exp(L - baseline) in RFLOAT and rescales the sums every
        # time the baseline moves (src/Optimiser.cpp:1383-1402), so weights many orders below the largest are already
        # rounding noise in the reference itself (the device forms them in closed form, in double)
        bar = weight_bar(own.get("scaleL", 1.0))
        for dev, mine, name in ((uR, own["uR"], "uR"), (uT, own["uT"], "uT")):
            err = np.abs(dev.astype(np.float64) - mine)
            self.max_rel = max(self.max_rel, float(err.max() / mine.max()))
            assert np.all(err <= bar * np.abs(mine) + 1e-6 * mine.max()), \
                "phase %d image %d %s: %.3g of the largest weight" % (p, l, name, err.max() / mine.max())
        self.n_checked += 1
        if "uD" in own:
            # CTF search: the weights of the defocus factors (Particle::setUD), the factors after initD / perturb(PAR_D) and after
            # resample(mLD, PAR_D) -- the same tie rule: a resampling that differs must be what the oracle's resampler gives for the
            # device's weights
            uD, dP, dR = cap["uD"][p, l], cap["dP"][p, l], cap["dR"][p, l]
            assert np.abs(dP - own["dIn"]).max() <= 1e-12, "phase %d image %d: perturbed defocus factors differ" % (p, l)
            err = np.abs(uD.astype(np.float64) - own["uD"])
            assert np.all(err <= bar * np.abs(own["uD"]) + 1e-6 * own["uD"].max()), "phase %d image %d uD: %.3g of the largest weight" % (p, l, err.max() / own["uD"].max())
            if np.abs(dR - own["d"]).max() > 1e-12:
                mLD = len(dR)
                d2, wD2, sD, topD, srcD = O.pf_update_d(own["dIn"], own["wDIn"], uD, PH.shuffle_ranks(c["seed"], own["li"], own["callU"], 11, mLD),
                                                        PH.draw_u4(c["seed"], own["li"], own["callU"], 12, 0)[0] / mLD)
                assert np.abs(dR - d2).max() <= 1e-12, "phase %d image %d: the device's defocus resampling is not what its own weights give" % (p, l)
                own.update(d=d2, wD=wD2, sD=sD, topD=topD, srcD=srcD)
                self.adopted.append((p, l, "resample D"))
        # discrete decisions: resampled indices
        srcR = self._match(cap["r"][p, l], own["qPre"])
        srcT = self._match(cap["t"][p, l], own["tPre"])
        if not (np.array_equal(srcR, own["srcR"]) and np.array_equal(srcT, own["srcT"])):
            # tie rule: the oracle's resampler on the device's weights must give the device's indices
            seed, mLR, mLT = c["seed"], c["mLR"], c["mLT"]
            li, call = own["li"], own["callU"]
            alt = O.pf_update(own["qIn"], own["tPre"], own["wRIn"], own["wTIn"], uR, uT, c["peakFactorR"],
                              PH.shuffle_ranks(seed, li, call, 2, mLR), PH.draw_u4(seed, li, call, 3, 0)[0] / mLR,
                              PH.shuffle_ranks(seed, li, call, 4, mLT), PH.draw_u4(seed, li, call, 5, 0)[0] / mLT,
                              symQuat=self.symQ, iAnchor=own.get("iAnchor", 0))
            exact = np.array_equal(alt["srcR"], srcR) and np.array_equal(alt["srcT"], srcT)
            if not exact:
                # the device's priors (its own balanceWeight of the perturbed cloud) differ from the oracle's in the 6th digit:
                # every resampled index must still sit on its threshold to within 1e-5 of the (normalised) cumulative weight
                okR = self._on_threshold(srcR, own["wRIn"], alt["uRk"], PH.shuffle_ranks(seed, li, call, 2, mLR),
                                         PH.draw_u4(seed, li, call, 3, 0)[0] / mLR)
                okT = self._on_threshold(srcT, own["wTIn"], alt["uTk"], PH.shuffle_ranks(seed, li, call, 4, mLT),
                                         PH.draw_u4(seed, li, call, 5, 0)[0] / mLT)
                assert okR and okT, "phase %d image %d: the device's resampling is not what its own weights give" % (p, l)
                alt["q"], alt["t"], alt["srcR"], alt["srcT"] = alt["qPre"][srcR].copy(), own["tPre"][srcT].copy(), srcR, srcT
            self.adopted.append((p, l, "resample" if exact else "resample (threshold within 1e-5)"))
            for k in ("q", "t", "wR", "wT", "srcR", "srcT", "topR", "topT", "iTopR", "iTopT"):
                own[k] = alt[k]
        elif "iTopR" in own:
            # same resampled indices, but _topR / _topT = the FIRST LARGEST weight (src/Particle.cpp:1291-1430) is a discrete
            # decision too: two support points whose weights agree to within the weight bar can swap under the device's rounding
            seed, mLR, mLT = c["seed"], c["mLR"], c["mLT"]
            li, call = own["li"], own["callU"]
            alt = O.pf_update(own["qIn"], own["tPre"], own["wRIn"], own["wTIn"], uR, uT, c["peakFactorR"],
                              PH.shuffle_ranks(seed, li, call, 2, mLR), PH.draw_u4(seed, li, call, 3, 0)[0] / mLR,
                              PH.shuffle_ranks(seed, li, call, 4, mLT), PH.draw_u4(seed, li, call, 5, 0)[0] / mLT,
                              symQuat=self.symQ, iAnchor=own.get("iAnchor", 0))
            if alt["iTopR"] != own["iTopR"] or alt["iTopT"] != own["iTopT"]:
                for mine, ia, io, name in ((np.asarray(own["uR"], np.float64), alt["iTopR"], own["iTopR"], "uR"),
                                           (np.asarray(own["uT"], np.float64), alt["iTopT"], own["iTopT"], "uT")):
                    assert abs(mine[ia] - mine[io]) <= 2 * bar * abs(mine[io]) + 2e-6 * mine.max(), \
                        "phase %d image %d: the device's top %s is not a tie in the oracle's weights" % (p, l, name)
                self.adopted.append((p, l, "top"))
                for k in ("topR", "topT", "iTopR", "iTopT"):
                    own[k] = alt[k]
        # the per-image stop rule (maxPhase > nPhase) compares the phase's variances with 0.95 x the smallest seen: a discrete decision
        # on numbers that carry the ACG estimate's accuracy -- the oracle's rule is fed the DEVICE's variances of this phase (held to
        # the oracle's own by the caller), so that the two sides must take the same decision in the same phase
        own["k_stop"], own["s_stop"] = cap["k123"][p, l].copy(), cap["s01"][p, l].copy()
        return own


def fsc_curve(O, a, b, N, n):
    return O.fsc(sfft.rfftn(a).astype(np.complex64), sfft.rfftn(b).astype(np.complex64), N, n)
