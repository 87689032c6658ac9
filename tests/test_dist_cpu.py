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
