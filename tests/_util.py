"""Shared seeded inputs for the parity tests (host side, numpy)."""
import numpy as np

from thunder_amd import synth


def make_case(O, N, seed=7, rU=None, rL=0, pf=2):
    """reference-like test case: blob map -> padded projectee FT (oracle), pixel list"""
    rU = N // 2 - 2 if rU is None else rU
    ref = synth.blob_map(N, seed=seed, nblob=12)
    vol = O.set_projectee(ref, pf)
    pl = O.pixel_list(N, rU, rL, pf)
    return ref, vol, pl


def edge_rotations(rng, n_random=8):
    """identity, 90-degree turns, slight tilts that put y0 / z0 at -1, x<0 folding, random"""
    mats = []
    I = np.eye(3)
    mats.append(I)
    Rz90 = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    Rx90 = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0.0]])
    Ry90 = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0.0]])
    mats += [Rz90, Rx90, Ry90, Rz90 @ Rz90, Rx90 @ Rz90]
    for a in (0.013, -0.013):
        c, s = np.cos(a), np.sin(a)
        mats.append(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]))   # y slightly negative at j = 0
        mats.append(np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]))      # z slightly negative
    q = synth.random_quats(n_random, rng)
    for qq in q:
        mats.append(quat_to_mat(qq))
    # column-major flatten
    return np.stack([m.T.reshape(-1) for m in mats]).astype(np.float64)


def quat_to_mat(q):
    A = np.array([[0, -q[3], q[2]], [q[3], 0, -q[1]], [-q[2], q[1], 0]])
    return np.eye(3) + 2 * q[0] * A + 2 * A @ A


def make_images(O, vol, pl, N, nImg, rng, pf=2, snr_sigma=1.0):
    """nImg noisy images on the pixel list: CTF * slice * ramp + noise; returns dict of host arrays"""
    P = N * pf
    nPxl = pl["nPxl"]
    quat = synth.random_quats(nImg, rng)
    rot = np.stack([O.rotate3D(q) for q in quat])
    shift = rng.normal(0, 1.5, size=(nImg, 2))
    attr = synth.ctf_params(nImg, rng)
    dat = np.zeros((nImg, nPxl), np.complex64)
    ctf = np.zeros((nImg, nPxl), np.float32)
    for l in range(nImg):
        s = O.project(vol, P, pf, rot[l], pl["iCol"], pl["iRow"])
        ctf[l] = O.ctf(1.32, *attr[l], N, pl["iCol"], pl["iRow"])
        ramp = O.translate(shift[l, 0], shift[l, 1], N, pl["iCol"], pl["iRow"])
        sig = s * ctf[l] * ramp
        sd = np.sqrt(np.mean(np.abs(sig) ** 2)) * snr_sigma
        noise = (rng.normal(size=nPxl) + 1j * rng.normal(size=nPxl)) * sd / np.sqrt(2)
        dat[l] = (sig + noise).astype(np.complex64)
    sigma2 = np.mean(np.abs(dat) ** 2, axis=0, keepdims=True).repeat(nImg, 0)
    sigRcp = (-0.5 / sigma2).astype(np.float32)
    return dict(quat=quat, rot=rot, shift=shift, attr=attr, dat=dat, ctf=ctf, sigRcp=sigRcp)
