"""Seeded inputs of one K-class classification iteration and the oracle's chain of its first three stages (test infrastructure,
numpy + oracle only): scan weights over the K classes (src/Optimiser.cpp:756-894), the class of every image (:925-952) and the
support points of the local search (:953-1079), with the device's Philox draws replayed (tests/_philox.py; call 1 = class
selection, call 2 = support points, image index = the image's index in the rank's shard)."""
import numpy as np

import _philox as PH


def make_case(O, N, K, nImg, nR, nT, seed, noise=0.5):
    """K blob references, images = CTF x slice of a random class at a scanned rotation x ramp of a scanned shift + noise, as rows
    on the rL = 0 list"""
    from thunder_amd import synth
    rng = np.random.default_rng(seed)
    pf, P, rU = 2, 2 * N, N // 2 - 2
    refs = np.stack([synth.blob_map(N, seed=seed + 10 + k, nblob=10) for k in range(K)])
    vols = [O.set_projectee(refs[k], pf) for k in range(K)]
    plM = O.pixel_list(N, rU, 0, pf)
    quat = synth.random_quats(nR, rng)
    shifts = np.ascontiguousarray(rng.normal(0, 1.5, size=(nT, 2)))
    cls_true, r_true, t_true = rng.integers(0, K, nImg), rng.integers(0, nR, nImg), rng.integers(0, nT, nImg)
    attr = synth.ctf_params(nImg, rng)
    ctfM = np.stack([O.ctf(1.32, *attr[l], N, plM["iCol"], plM["iRow"]) for l in range(nImg)])
    datM = np.stack([O.project(vols[cls_true[l]], P, pf, O.rotate3D(quat[r_true[l]]), plM["iCol"], plM["iRow"]) * ctfM[l]
                     * O.translate(np.float32(shifts[t_true[l], 0]), np.float32(shifts[t_true[l], 1]), N, plM["iCol"], plM["iRow"])
                     for l in range(nImg)]).astype(np.complex64)
    sd = noise * float(np.sqrt(np.mean(np.abs(datM) ** 2)))
    datM = (datM + (rng.standard_normal(datM.shape) + 1j * rng.standard_normal(datM.shape)) * (sd / np.sqrt(2))).astype(np.complex64)
    sigM = np.full(datM.shape, np.float32(-0.5 / (sd * sd / 2)), np.float32)
    return dict(refs=refs, vols=vols, plM=plM, quat=quat, shifts=shifts, cls_true=cls_true, r_true=r_true, datM=datM, ctfM=ctfM, sigM=sigM)


def sub_rows(plM, plS):
    posM = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(plM["iCol"], plM["iRow"]))}
    return np.asarray([posM[(int(i), int(j))] for i, j in zip(plS["iCol"], plS["iRow"])])


def stages(O, N, K, vols, quat, shifts, rScan, rL):
    """oracle.ClassifyStages (scan -> classes -> support points) fed with the numpy replica of the device's Philox streams"""
    return O.ClassifyStages(N, K, vols, quat, shifts, rScan, rL, PH)
