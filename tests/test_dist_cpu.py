"""world_size 2 and 4 over gloo on CPU: the half-set (hemisphere) groups, the F/T all-reduce within a half, the
half-map exchange and the particle sharding rule of thunder_amd.refine (the N>1 data path of bench.py).
The numeric insertion itself runs through the oracle here (allowed: tests/ only) so that
sum-over-ranks(insert(shard)) == insert(all particles of the half) is checked end to end without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from thunder_amd.refine import HalfGroups, shard_indices
    from _util import make_case, make_images
    g = HalfGroups(rank, world)
    assert g.local_halves() == (rank % 2,)
    # identical global data set on every rank (seeded), sharded by the rule under test
    N, P, nTot, mReco = 16, 32, 12, 2
    rng = np.random.default_rng(123)
    ref, vol, pl = make_case(O, N)
    im = make_images(O, vol, pl, N, nTot, rng)
    mine = shard_indices(nTot, rank, world)
    F = np.zeros((P, P, P // 2 + 1), np.complex64)
    T = np.zeros((P, P, P // 2 + 1), np.float32)
    for l in mine:
        for m in range(mReco):
            O.insertP(F, T, P, im["dat"][l], im["ctf"][l], im["rot"][l], np.float32(1.0 / mReco), pl["iColPad"], pl["iRowPad"])
    Ft, Tt = torch.from_numpy(F), torch.from_numpy(T)
    g.allreduce_half(Tt)
    g.allreduce_half(Ft)
    # half maps: a tag volume per half, exchanged to everyone
    tag = torch.full((4, 4, 4), float(10 + g.half))
    a, b = g.exchange_half_maps({g.half: tag})
    q.put((rank, mine.tolist(), Ft.numpy().copy(), Tt.numpy().copy(), float(a[0, 0, 0]), float(b[0, 0, 0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_half_groups_and_sharded_insert(world, oracle):
    O = oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import make_case, make_images
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda x: x[0])
    N, P, nTot, mReco = 16, 32, 12, 2
    # partition: disjoint, complete, half-consistent
    allidx = sorted(i for r in res for i in r[1])
    assert allidx == list(range(nTot))
    for rank, mine, *_ in res:
        assert all(i % 2 == rank % 2 for i in mine)
    # reference: single-process insertion of each half
    rng = np.random.default_rng(123)
    ref, vol, pl = make_case(O, N)
    im = make_images(O, vol, pl, N, nTot, rng)
    for h in (0, 1):
        F = np.zeros((P, P, P // 2 + 1), np.complex64)
        T = np.zeros((P, P, P // 2 + 1), np.float32)
        for l in range(h, nTot, 2):
            for m in range(mReco):
                O.insertP(F, T, P, im["dat"][l], im["ctf"][l], im["rot"][l], np.float32(1.0 / mReco), pl["iColPad"],
                          pl["iRowPad"])
        for rank, mine, Fr, Tr, a, b in res:
            if rank % 2 == h:
                assert np.abs(Fr - F).max() <= 1e-5 * np.abs(F).max()
                assert np.abs(Tr - T).max() <= 1e-5 * np.abs(T).max()
    for rank, mine, Fr, Tr, a, b in res:
        assert (a, b) == (10.0, 11.0)


def test_shard_indices_world1_and_odd_counts():
    from thunder_amd.refine import shard_indices
    assert shard_indices(7, 0, 1).tolist() == list(range(7))
    got = sorted(i for r in range(8) for i in shard_indices(101, r, 8).tolist())
    assert got == list(range(101))
    sizes = [len(shard_indices(100000, r, 8)) for r in range(8)]
    assert sizes == [12500] * 8


def test_pixel_visit_order_is_a_local_permutation(monkeypatch):
    """the E-step visits the listed pixels along a Morton curve: a permutation of the list in which consecutive pixels
    stay close (that is what keeps the gathered volume cells cached)"""
    from thunder_amd.refine import pixel_list, pixel_visit_order
    N = 64
    pl = pixel_list(N, N // 2 - 2, 2)
    for mode in ("morton", "tile"):
        monkeypatch.setenv("THX_PIXEL_ORDER", mode)
        o = pixel_visit_order(pl, N)
        assert sorted(o.tolist()) == list(range(pl["nPxl"]))
        ic, ir = pl["iCol"][o].astype(int), pl["iRow"][o].astype(int)
        step = np.hypot(np.diff(ic), np.diff(ir))
        assert np.median(step) <= 1.5 and np.mean(step) < 4.0
    monkeypatch.setenv("THX_TILE_ORDER", "0")
    assert pixel_visit_order(pl, N) is None


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_half_part_deals_a_one_rank_job(world):
    """refine.half_part: the contiguous part of the ONE-RANK layout ([0, ceil(n / 2)) = half 0, the rest = half 1) that rank r of
    `world` holds -- rank r is in half r mod 2 (src/Parallel.cpp:26-36) and is rank r // 2 inside it; the parts of a half tile it
    exactly, in rank order, so that a shard's image numbering (thx_refine_set_image_base) is the one-rank run's.  The owner of class
    k of half h (thx_refine.hip: owner_in_half) is world rank 2 (k mod ranks of the half) + h, which must be a rank of that half."""
    sys.path.insert(0, ROOT)
    from thunder_amd.refine import half_part
    for n in (7, 240, 384, 100000):
        nA = (n + 1) // 2
        seen = np.zeros(n, np.int32)
        for h in (0, 1):
            if world == 1 and h == 1:
                continue
            ranks = [r for r in range(world) if r % 2 == h] if world > 1 else [0]
            edge = 0 if h == 0 else nA
            for r in ranks:
                lo, hi = half_part(n, r, world)
                if world == 1:
                    assert (lo, hi) == (0, n)
                    seen[lo:hi] += 1
                    continue
                assert lo == edge and hi >= lo, (n, r, lo, hi, edge)     # contiguous, in rank order
                assert (lo >= nA) == (h == 1) or hi == lo
                edge = hi
                seen[lo:hi] += 1
            if world > 1:
                assert edge == (nA if h == 0 else n)
        if world > 1 or True:
            assert np.all(seen == 1), (n, world)
    for h in (0, 1):
        H = (world - h + 1) // 2 if world > 1 else 1
        for k in range(6):
            owner = 2 * ((k % H) if H > 1 else 0) + h
            if world > 1 and H > 0:
                assert owner < world and owner % 2 == h
