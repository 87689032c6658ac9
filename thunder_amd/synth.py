"""Seeded synthetic inputs for tests and bench.py: a Gaussian-blob map, per-particle
pose / CTF / noise, and the fixed-work support points fed to the timed iteration.  numpy only (host side);
the per-particle images themselves are produced on the device by the product's own project kernel.
"""
import numpy as np


def blob_map(N, seed=20240601, nblob=40, symR=None):
    """N^3 float32 map in wrapped-index layout (origin at [0,0,0]): sum of isotropic Gaussians inside 0.3 N,
    times a soft spherical mask of radius 0.4 N.  symR [nSym][9] (column-major rotation matrices of a point group's
    non-identity elements): every blob is repeated at its symmetry mates, so the map has that point group."""
    rng = np.random.default_rng(seed)
    ax = (np.fft.fftfreq(N) * N).astype(np.float32)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    m = np.zeros((N, N, N), np.float32)
    mats = [np.eye(3)] + ([] if symR is None else [np.asarray(r, np.float64).reshape(3, 3).T for r in symR])
    for _ in range(nblob):
        c0 = rng.normal(size=3)
        c0 = c0 / np.linalg.norm(c0) * rng.uniform(0, 0.3 * N)
        s = rng.uniform(1.5, 4.0) * N / 256.0 + 1.0
        a = rng.uniform(0.5, 1.0)
        for R in mats:
            c = R @ c0
            m += (a * np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (2 * s * s))).astype(np.float32)
    r = np.sqrt(x * x + y * y + z * z)
    edge = 0.05 * N
    mask = np.clip((0.4 * N + edge - r) / edge, 0, 1)
    mask = (0.5 - 0.5 * np.cos(np.pi * mask)).astype(np.float32)
    return (m * mask).astype(np.float32)


def random_quats(n, rng):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def quat_mul(a, b):
    """Hamilton product, broadcasting over leading dims"""
    a0, a1, a2, a3 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    b0, b1, b2, b3 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                     a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1, a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0], axis=-1)


def perturb_quats(q, n, std, rng):
    """n small rotations (angle ~ N(0, std) about random axes) composed with each base quaternion q [m][4]
    -> [m][n][4]; the first of each set is the unperturbed pose."""
    m = q.shape[0]
    axis = rng.normal(size=(m, n, 3))
    axis /= np.linalg.norm(axis, axis=2, keepdims=True)
    ang = rng.normal(scale=std, size=(m, n))
    ang[:, 0] = 0.0
    dq = np.concatenate([np.cos(ang / 2)[..., None], np.sin(ang / 2)[..., None] * axis], axis=2)
    out = quat_mul(q[:, None, :], dq)
    return out / np.linalg.norm(out, axis=2, keepdims=True)


def cn_symmetry(n):
    """the n-1 non-identity rotations about z of point group Cn as column-major 3x3 (Symmetry::fillLR,
    src/Geometry/Symmetry.cpp:146-171)"""
    mats = []
    for j in range(1, n):
        a = 2 * np.pi * j / n
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float64)
        mats.append(R.T.reshape(-1))  # column-major flatten
    return np.asarray(mats, np.float64).reshape(-1, 9)


def ctf_params(n, rng):
    """[n][7] float32: voltage, defocusU, defocusV, theta, Cs, amplitudeContrast, phaseShift"""
    dU = rng.uniform(1.0e4, 3.0e4, n)
    dV = dU + rng.normal(0, 300, n)
    th = rng.uniform(0, np.pi, n)
    a = np.stack([np.full(n, 3e5), dU, dV, th, np.full(n, 2.7e7), np.full(n, 0.1), np.zeros(n)], axis=1)
    return a.astype(np.float32)
