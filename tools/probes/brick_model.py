"""What is the COMPULSORY memory traffic of the local-search E-step for the particle filter's real clouds?
Input: gpurun_out/clouds.npz (tools/cloud_dump.py: the support points the kernel sees in each phase of the second iteration of
the bench workload).  For a few images and every phase: the spread of the cloud, the number of distinct 64-byte cells
(cell-packed projector), distinct 128-byte lines and distinct 8-byte voxels (standard layout) touched per sample, and the
share of samples an LDS-staged sub-volume of margin m voxels around the mean rotation's slice would serve."""
import sys
import numpy as np
sys.path.insert(0, ".")
from thunder_amd.refine import pixel_list

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/clouds.npz")
pl = pixel_list(256, 126, 2)
p = np.stack([pl["iCol"] * 2.0, pl["iRow"] * 2.0], 1)
rho = np.hypot(p[:, 0], p[:, 1])


def qmat(q):
    q0, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (q2 * q2 + q3 * q3); R[..., 0, 1] = 2 * (q1 * q2 - q0 * q3); R[..., 0, 2] = 2 * (q1 * q3 + q0 * q2)
    R[..., 1, 0] = 2 * (q1 * q2 + q0 * q3); R[..., 1, 1] = 1 - 2 * (q1 * q1 + q3 * q3); R[..., 1, 2] = 2 * (q2 * q3 - q0 * q1)
    R[..., 2, 0] = 2 * (q1 * q3 - q0 * q2); R[..., 2, 1] = 2 * (q2 * q3 + q0 * q1); R[..., 2, 2] = 1 - 2 * (q1 * q1 + q2 * q2)
    return R


print("samples per image-phase: %d pixels x 125 rotations = %d; algorithmic bytes 64 B each = %.1f MB" % (len(p), len(p) * 125, len(p) * 125 * 64 / 1e6))
for ph in range(3):
    q = d["quat_phase%d" % ph]
    R = qmat(q)
    U, S, Vt = np.linalg.svd(R.mean(1))
    R0 = U @ Vt
    tr = np.trace(np.einsum("nji,nmjk->nmik", R0, R), axis1=2, axis2=3)
    ang = np.degrees(np.arccos(np.clip((tr - 1) / 2, -1, 1)))
    cells, lines, vox, hit = [], [], [], {m: [] for m in (2, 4, 6, 8)}
    for l in range(8):
        pos = np.einsum("mij,pj->mpi", R[l][:, :, :2], p)
        pos0 = np.einsum("ij,pj->pi", R0[l][:, :2], p)
        dev = np.abs(pos - pos0[None]).max(-1)                     # Linf distance of a sample from the mean rotation's slice
        for m in hit:
            hit[m].append((dev <= m).mean())
        neg = pos[..., 0] < 0
        pos[neg] *= -1
        c = np.floor(pos).astype(np.int64)
        key = ((c[..., 2] + 512) << 22) + ((c[..., 1] + 512) << 11) + c[..., 0]
        cells.append(np.unique(key).size / key.size)
        lines.append(np.unique(((c[..., 2] + 512) << 22) + ((c[..., 1] + 512) << 11) + (c[..., 0] >> 1)).size / key.size)
        v = []
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    v.append((((c[..., 2] + dz + 512) << 22) + ((c[..., 1] + dy + 512) << 11) + c[..., 0] + dx).ravel())
        vox.append(np.unique(np.concatenate(v)).size / key.size)
    print("phase %d: angle from the cloud's mean rotation: median %.2f deg, 75 %% %.2f, 90 %% %.2f, 99 %% %.1f" % (ph, *np.percentile(ang, [50, 75, 90, 99])))
    print("   distinct 64-B cells / sample %.3f  (= %.0f MB compulsory with the cell-packed projector)" % (np.mean(cells), np.mean(cells) * len(p) * 125 * 64 / 1e6))
    print("   distinct 128-B lines / sample %.3f ; distinct voxels / sample %.2f (= %.0f MB at 8 B per voxel)" % (np.mean(lines), np.mean(vox), np.mean(vox) * len(p) * 125 * 8 / 1e6))
    print("   share of samples within m voxels of the mean slice (what an LDS sub-volume of that margin serves): " + ", ".join("m=%d: %.2f" % (m, np.mean(h)) for m, h in hit.items()))
