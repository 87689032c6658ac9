"""numpy statement of the Philox4x32-10 draws of thunder_amd/csrc/thx_pf.hip (test infrastructure): lets the particle-filter
tests replay the device's random choices (perturbation normals, shuffle keys, u0) and check everything else exactly."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox(seed, c0, c1, c2, c3):
    """counters broadcastable uint32 arrays -> 4 uint32 arrays"""
    c = [np.asarray(x, np.uint64) & np.uint64(0xFFFFFFFF) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    ka, kb = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(ka)) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(kb)) & mask
        n3 = p0 & mask
        c = [n0, n1, n2, n3]
        ka, kb = (ka + W0) & 0xFFFFFFFF, (kb + W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def draw_u4(seed, img, call, purpose, index):
    return [(x.astype(np.float64) + 0.5) / 4294967296.0 for x in philox(seed, img, call, purpose, index)]


def draw_n4(seed, img, call, purpose, index):
    u = draw_u4(seed, img, call, purpose, index)
    r0, r1 = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
    a0, a1 = 6.283185307179586476925 * u[1], 6.283185307179586476925 * u[3]
    return [r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)]


def shuffle_ranks(seed, img, call, purpose, n):
    """rank[i] = new position of element i (keys compared as in the kernel: by key, ties by index)"""
    keys = philox(seed, img, call, purpose, np.arange(n))[0]
    order = np.lexsort((np.arange(n), keys))
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    return rank
