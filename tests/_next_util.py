"""Seeded full-image inputs for the re-mask / sigma tests (host side, numpy)."""
import numpy as np

from thunder_amd import synth


def full_images(O, vol, N, nImg, rng, projR, pf=2, pixelSize=1.32):
    """nImg full image FTs [N][N/2+1]: CTF * slice * ramp on the disc of radius projR + white noise everywhere.
    Returns dict(img, imgOri, rot, tran, offset, attr)."""
    P = N * pf
    dl = O.disc_list(N, projR)
    quat = synth.random_quats(nImg, rng)
    rot = np.stack([O.rotate3D(q) for q in quat])
    tran = rng.normal(0, 1.5, size=(nImg, 2))
    offset = rng.normal(0, 0.7, size=(nImg, 2))
    attr = synth.ctf_params(nImg, rng)
    img = np.zeros((nImg, N, N // 2 + 1), np.complex64)
    ori = np.zeros_like(img)
    for l in range(nImg):
        s = O.project(vol, P, pf, rot[l], dl["iCol"], dl["iRow"])
        c = O.ctf(pixelSize, *attr[l], N, dl["iCol"], dl["iRow"])
        sigM = s * c * O.translate(tran[l, 0], tran[l, 1], N, dl["iCol"], dl["iRow"])
        sigN = s * c * O.translate(tran[l, 0] - offset[l, 0], tran[l, 1] - offset[l, 1], N, dl["iCol"], dl["iRow"])
        sd = np.sqrt(np.mean(np.abs(sigM) ** 2))
        noise = ((rng.normal(size=img[l].shape) + 1j * rng.normal(size=img[l].shape)) * sd / np.sqrt(2))
        a = noise.astype(np.complex64).copy()
        b = (noise * 1.1).astype(np.complex64)
        a.reshape(-1)[dl["iPxl"]] += sigM.astype(np.complex64)
        b.reshape(-1)[dl["iPxl"]] += sigN.astype(np.complex64)
        img[l], ori[l] = a, b
    return dict(img=img, imgOri=ori, rot=rot, tran=tran, offset=offset, attr=attr, pixelSize=pixelSize)
