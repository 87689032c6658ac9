"""N ranks == 1 rank, on ONE GPU: 2 and 4 processes share cuda:0 and run the unchanged multi-rank branches of the native driver
(thx_refine_iterate's world / hemi paths, thx_comm.hip) over the TEST-ONLY shared-memory transport (RCCL refuses two ranks per
device), on the particles of a one-rank job dealt to the ranks (refine.take_shard); everything a rank ends with is held to the
one-rank run of the same job (tests/_rank_worker.py runs both).

Reference for what is exchanged: gpu/src/cuthunder.cu:4192-4206 (id bootstrap), :4972-5067 (ncclAllReduce of F, T, O, counter),
src/Parallel.cpp:26-36 (odd / even hemispheres), src/Reconstructor.cpp:2383,2436 (MPI_Allreduce_Large over _hemi),
src/Optimiser.cpp:6362-6367 (norm vector over MPI_COMM_WORLD), :6608-6650 (sigma tables over the hemisphere), :5484-5593 (class
distribution, balanceClass drawn once for everybody), src/Model.cpp:375-391 (half maps to the master).

Bars.  Iteration 1 starts from identical state, the Philox streams are numbered by the image's index in the job, and the F / T
accumulators are 64-bit integers reduced as integers: everything per particle and F / T after the reduce are BITWISE equal, and so
is everything computed from them by deterministic kernels (maps, FSC).  The sigma tables are float sums over the half's ranks
(another order than one rank's): 2e-6 relative.  From iteration 2 on the E-step sees those sigma tables, so the bars are the chain
tests': weights move by 1e-6, a resampling threshold may flip for an image.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_rank_worker.py")


def run_ranks(world, case, extra_env=None, timeout=900, rccl=False):
    d = tempfile.mkdtemp(prefix="thx_ranks_%s_w%d_" % (case, world))
    env = dict(os.environ)
    env["THX_COMM_TRANSPORT"] = "rccl" if rccl else "shm"
    env["THX_COMM_SHM_TIMEOUT_S"] = "240"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    procs = []
    for r in range(world):
        log = open(os.path.join(d, "rank%d.log" % r), "w")
        procs.append((subprocess.Popen([sys.executable, WORKER, "--rank", str(r), "--world", str(world), "--dir", d, "--case", case] +
                                       (["--gpu-per-rank"] if rccl else []),
                                       stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT), log))
    failed = None
    try:
        for r, (p, log) in enumerate(procs):
            try:
                rc = p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                rc = -999
            if rc != 0 and failed is None:
                failed = (r, rc)
    finally:
        for p, log in procs:
            if p.poll() is None:
                p.kill()        # (exactly the processes started here)
                p.wait()
            log.close()
    if failed is not None:
        tail = open(os.path.join(d, "rank%d.log" % failed[0])).read()[-3000:]
        raise AssertionError("rank %d of %d exited with %d:\n%s" % (failed[0], world, failed[1], tail))
    out = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(world)]
    shutil.rmtree(d, ignore_errors=True)
    return out


_one_rank = {}


def one_rank(case):
    if case not in _one_rank:
        _one_rank[case] = run_ranks(1, case)[0]
    return _one_rank[case]


def gather_particles(ranks, key, n):
    out = None
    for g in ranks:
        lo, hi = g["lo_hi"]
        a = g[key]
        if out is None:
            out = np.zeros((n,) + a.shape[1:], a.dtype)
        out[lo:hi] = a
    return out


def owner_rank(world, half, k, replicate=False):
    H = (world - half + 1) // 2
    return 2 * ((k % H) if (H > 1 and not replicate) else 0) + half


def compare(case, world, ranks, ref, nIter, K, replicate=False):
    n = ref["cls0"].shape[0]
    relmax = lambda a, b: float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
    report = []
    for it in range(nIter):
        first = it == 0
        # ---- per particle: class, top pose, support points, re-centring offset
        for key in ("cls", "topR", "topT", "offset", "r"):
            got, want = gather_particles(ranks, "%s%d" % (key, it), n), ref["%s%d" % (key, it)]
            if first:
                assert np.array_equal(got, want), "%s of iteration 1 is not bitwise the one-rank run's" % key
            elif key == "cls":
                assert (got != want).mean() <= 0.02
            elif key in ("topR", "topT"):
                bad = np.abs(got - want).max(axis=1) > 1e-6
                assert bad.mean() <= 0.05, (key, it, bad.mean())      # (a flipped resampling threshold moves one image's top point)
        # ---- F / T after the half-set reduce, from the rank that reconstructs the class
        for hf in (0, 1):
            for k in range(K):
                g = ranks[owner_rank(world, hf, k, replicate)]
                F, T = g["Fraw%d" % it][0][k], g["Traw%d" % it][0][k]
                Fw, Tw = ref["Fraw%d" % it][hf][k], ref["Traw%d" % it][hf][k]
                if first:
                    assert np.array_equal(F, Fw) and np.array_equal(T, Tw), "F / T of half %d class %d are not bitwise the one-rank sums" % (hf, k)
                else:
                    assert relmax(F, Fw) <= 2e-2 and relmax(T, Tw) <= 2e-2, (it, hf, k, relmax(F, Fw), relmax(T, Tw))
                    report.append(("F", it, hf, k, relmax(F, Fw)))
        # ---- sigma tables of the rank's half: float sums over the half's ranks
        for r, g in enumerate(ranks):
            s, sw = g["sig%d" % it][0], ref["sig%d" % it][r % 2]
            tol = 2e-6 if first else 2e-3
            assert np.all(np.abs(s - sw) <= tol * np.abs(sw) + 1e-30), (it, r, float(np.abs(s / sw - 1).max()))
        # ---- every rank holds every map of both halves, the FSC, the balanceClass decision, the class histogram
        for r, g in enumerate(ranks):
            assert np.array_equal(g["balanced%d" % it], ref["balanced%d" % it]), (it, r, g["balanced%d" % it][:K], ref["balanced%d" % it][:K])
            for key in ("mapsFsc", "maps"):
                a, b = g["%s%d" % (key, it)], ref["%s%d" % (key, it)]
                if key == "maps":      # the references Model::refreshProj consumed: a rank flattens and refreshes its OWN half's
                    a, b = a[r % 2], b[r % 2]
                if first and key == "mapsFsc":
                    assert np.array_equal(a, b), "MAP-off half maps of iteration 1 on rank %d are not bitwise the one-rank run's" % r
                elif first:
                    # the MAP-on maps go through the averaging / flattening on the same inputs: still the same bits
                    assert np.array_equal(a, b), "references after iteration 1 on rank %d differ from the one-rank run's" % r
                else:
                    e = relmax(a, b)
                    report.append((key, it, r, e))
                    assert e <= 5e-2, (key, it, r, e)
            f, fw = g["fsc%d" % it], ref["fsc%d" % it]
            if first:
                assert np.array_equal(f, fw)
            else:
                assert np.abs(f - fw)[:, :N_INNER].max() <= 2e-2, (it, r, np.abs(f - fw).max())
        assert sum(int(g["classCount%d" % it][:K].sum()) for g in ranks) == n
        assert np.array_equal(sum(g["classCount%d" % it] for g in ranks), ref["classCount%d" % it])
        # ---- normCorrection: the norms of every image and the median over ALL particles of the job
        if ("norm%d" % it) in ref:
            got, want = gather_particles(ranks, "norm%d" % it, n), ref["norm%d" % it]
            assert np.all(np.abs(got - want) <= 2e-4 * np.abs(want)), float(np.abs(got / want - 1).max())
            for g in ranks:
                assert abs(float(g["normMedian%d" % it]) - float(ref["normMedian%d" % it])) <= 2e-4 * float(ref["normMedian%d" % it])
                assert float(g["normMedian%d" % it]) == float(ranks[0]["normMedian%d" % it])      # one median for everybody
    return report


N_INNER = 8


@pytest.mark.parametrize("world", [2, 4])
def test_n_ranks_equal_one_rank_refinement(dev, world):
    """K = 1, local search, normCorrection ON (the configuration bench.py times), three iterations"""
    ref = one_rank("k1")
    ranks = run_ranks(world, "k1")
    rep = compare("k1", world, ranks, ref, 3, 1)
    print("world %d, k1:" % world, rep)
    # who reconstructs: with two ranks per half the MAP-off pass runs on the half's first rank and the MAP-on pass -- which needs
    # LAST iteration's FSC only -- on its second, at the same time; with one rank per half that rank runs both.  Everybody ends with
    # the maps (compare() above)
    H = world // 2
    for r, g in enumerate(ranks):
        rounds = g["rounds0"]          # [MAP off / on][local half][class]
        if H >= 2:
            assert (rounds[0, 0, 0] > 0) == (r // 2 == 0) and (rounds[1, 0, 0] > 0) == (r // 2 == 1), (r, rounds[:, :, 0])
        else:
            assert (rounds[:, 0, 0] > 0).all(), (r, rounds[:, :, 0])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_n_ranks_equal_one_rank_classification(dev, world):
    """K = 4 with balanceClass (class 3 has no particles and takes over another class's reference -- the same one on every
    rank): global search, then a local search.  With 4 ranks every rank of a half reconstructs two of its four classes; with 8
    -- the shape of BASELINE configs[3] on a node -- every rank reconstructs ONE class."""
    ref = one_rank("k4")
    assert ref["balanced0"][3] >= 0 and ref["balanced0"][3] != 3, ref["balanced0"][:4]
    ranks = run_ranks(world, "k4")
    rep = compare("k4", world, ranks, ref, 2, 4)
    print("world %d, k4:" % world, rep)
    H = world // 2
    for r, g in enumerate(ranks):
        rounds = g["rounds0"][:, 0, :]          # [MAP off / on][class]
        mine = [k for k in range(3) if k % H == r // 2]
        assert all((rounds[:, k] > 0).all() for k in mine), (r, rounds)
        assert all((rounds[:, k] == 0).all() for k in range(3) if k not in mine), (r, rounds)


@pytest.mark.parametrize("world", [2, 4])
def test_n_ranks_equal_one_rank_with_cutoffs(dev, world):
    """Round 6: the frequency cut-offs as per-iteration inputs on every rank -- (r, rU) = (8, 9) -> (10, 12) -> (14, 14) at N = 32 with
    normCorrection on: the quanta, the integer accumulators and their half-set reduce live on the resized (pf size)^3 grids (44^3, 56^3,
    64^3), the reconstructing rank builds its plans at the new size, the FSC has rU shells.  Held to the one-rank run with the same
    cut-offs: iteration 1 bitwise."""
    ref = one_rank("k1cut")
    assert ref["Fraw0"].shape[-2] == 44 and ref["Fraw1"].shape[-2] == 56 and ref["Fraw2"].shape[-2] == 64
    assert np.all(ref["fsc0"][:, 9:] == 0) and np.all(ref["fsc1"][:, 12:] == 0) and ref["fsc1"][0, 10] != 0
    ranks = run_ranks(world, "k1cut")
    rep = compare("k1cut", world, ranks, ref, 3, 1)
    print("world %d, k1cut:" % world, rep)


def test_n_ranks_equal_one_rank_point_group(dev):
    """C4: prepareTF's symmetrisation runs on the reduced sums of the reconstructing rank"""
    ref = one_rank("c4")
    ranks = run_ranks(4, "c4", {"THX_COMM_SHM_SLOT_MB": "1"})     # (1 MiB slots: the accumulators travel in four chunks)
    compare("c4", 4, ranks, ref, 2, 1)


def test_replicated_form_gives_the_same(dev):
    """THX_RECO_OWNERS=0: every rank of a half all-reduces and reconstructs every class, as the reference's ranks do
    (src/Reconstructor.cpp:2383,2436) -- the same maps as the owner form, bit for bit in iteration 1"""
    ref = one_rank("k4")
    ranks = run_ranks(4, "k4", {"THX_RECO_OWNERS": "0"})
    compare("k4", 4, ranks, ref, 2, 4, replicate=True)
    for g in ranks:
        assert (g["rounds0"][:, 0, :3] > 0).all()


@pytest.mark.parametrize("world,case,K,nIter", [(2, "k1", 1, 3), (4, "k1", 1, 3), (4, "k4", 4, 2)])
def test_n_ranks_equal_one_rank_over_rccl(dev, world, case, K, nIter):
    """the same equality with RCCL as the transport and one GPU per rank -- ncclReduce / ncclAllReduce / ncclBroadcast themselves,
    which the 1-GPU boxes of this environment cannot run (RCCL refuses two ranks per device): skipped unless `world` GPUs are visible"""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    ref = one_rank(case)
    ranks = run_ranks(world, case, rccl=True)
    compare(case, world, ranks, ref, nIter, K)
