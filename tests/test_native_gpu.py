"""The native iteration driver (thx_refine_*) and the native RCCL layer (thx_comm_*) on the GPU box.

* the C++ driver reproduces the Python harness (thunder_amd.refine.RefineShard, the sequencing the other GPU tests use)
  on the same particles: same kernels, same Philox streams -> identical particle-filter decisions in the first iteration,
  half maps and FSC that agree to the accuracy the atomically inserted F / T allow;
* RCCL itself: with one GPU a one-rank communicator is the only one RCCL will build; THX_COMM_FORCE=1 makes the library
  issue the real ncclAllReduce / ncclBroadcast calls on it (sum over one rank = identity), which exercises the bootstrap,
  the sphere-row packing and the collective calls of thx_reco_allreduce end to end;
* tests/cpp/iteration.cpp: a torch-free C++ program driving rows -> phases -> sigma -> insert -> reduce -> reconstruct x 2
  -> refresh through the C ABI only (two processes over RCCL when two GPUs are visible).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_native_driver_matches_python_harness(dev):
    from thunder_amd import ops
    from thunder_amd.native import NativeRefine
    from thunder_amd.refine import RefineShard
    N, n = 64, 700
    sh = RefineShard(N, n, dev, mReco=20, batch=256)          # three batches per half: the batched paths are exercised
    nat = NativeRefine(sh)
    nat.reset()
    v = nat.view()
    assert (v.nPxl, v.nPxlM) == (sh.nPxl, sh.nPxlM)
    # identical pixel lists (host integer work on both sides) and identical state after reset
    assert np.array_equal(nat.fetch(v.iCol, np.int32, (sh.nPxl,)), sh.pl["iCol"])
    assert np.array_equal(nat.fetch(v.iRow, np.int32, (sh.nPxl,)), sh.pl["iRow"])
    assert np.array_equal(nat.fetch(v.datP, np.complex64, (n, sh.nPxl)), sh.datP.cpu().numpy())
    assert np.array_equal(nat.fetch(v.ctfP, np.float32, (n, sh.nPxl)), sh.ctfP.cpu().numpy())
    assert np.array_equal(nat.fetch(v.sigRcpP, np.float32, (n, sh.nPxl)), sh.sigRcpP.cpu().numpy())
    assert np.array_equal(nat.fetch(v.vols, np.complex64, tuple(sh.vols.shape)), sh.vols.cpu().numpy())
    fsc_py = sh.run(1)
    fsc_na = nat.iterate(timed=True)
    off, topR, topT = nat.state()
    # first iteration: every particle-filter decision is a function of bit-identical likelihoods and the same Philox
    # streams, so the same support points win; the shifts agree to rounding (the initial spread of the shifts, which
    # scales their perturbation, is torch.std in the harness and k_shift_sd in the driver)
    assert torch.equal(topR, sh.pf_state["topR"])
    assert (topT - sh.pf_state["topT"]).abs().max().item() <= 1e-9
    assert (off - sh.offset).abs().max().item() <= 1e-9 and off.abs().max().item() > 0.1
    # sigma tables: deterministic kernels on identical inputs
    sig = nat.fetch(v.sig, np.float32, tuple(sh.sig.shape))
    np.testing.assert_allclose(sig, sh.sig.cpu().numpy(), rtol=1e-6)
    # maps: the inserted F / T differ in the last bits (atomic order), the gridding loop amplifies that a little
    for h in (0, 1):
        a, b = nat.map(h), sh.last["maps"][h]
        fs = ops.fsc(ops.fft3d_fw(a), ops.fft3d_fw(b), N, N // 2).cpu().numpy()
        assert np.all(fs[:N // 2 - 3] >= 0.999), fs
        assert (a - b).abs().max().item() <= 2e-3 * b.abs().max().item()
    np.testing.assert_allclose(fsc_na[:N // 2 - 3], fsc_py[:N // 2 - 3], atol=2e-3)
    # second iteration keeps tracking
    fsc_py2, fsc_na2 = sh.run(1), nat.iterate(timed=True)
    np.testing.assert_allclose(fsc_na2[:N // 2 - 3], fsc_py2[:N // 2 - 3], atol=3e-2)   # two stochastic filters by now
    same = (nat.state()[1] == sh.pf_state["topR"]).all(dim=1).float().mean().item()
    assert same >= 0.9, same
    st = nat.stats()
    assert st.iterations == 2 and st.expectLaunches == 2 * 2 * sh.nPhase * 2 and st.insertLaunches == 2 * 2 * 2
    assert st.expectImages == 2 * sh.nPhase * n and st.insertImages == 2 * n and st.expectMs > 0 and st.insertMs > 0
    assert st.balancingRounds >= 2 * 4 * 5
    nat.close()


def test_native_rccl_single_rank_forced(dev, knob_env):
    from thunder_amd import capi
    from thunder_amd.capi import ptr, stream_ptr
    from thunder_amd.native import Comm
    knob_env("THX_COMM_FORCE", "1")
    comm = Comm(0, 1, lambda uid: uid)
    assert capi.load().thx_comm_size(comm.handle) == 1 and capi.load().thx_comm_rank(comm.handle) == 0
    g = torch.Generator(device=dev).manual_seed(5)
    a = torch.randn(100003, device=dev, generator=g)
    a0 = a.clone()
    comm.allreduce(a)
    d = torch.randn(77, device=dev, generator=g, dtype=torch.float64)
    d0 = d.clone()
    comm.allreduce(d)
    b = torch.arange(1000, device=dev, dtype=torch.int32)
    comm.broadcast(b, 0)
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(d, d0) and torch.equal(b, torch.arange(1000, device=dev, dtype=torch.int32))
    # thx_reco_allreduce: pack the sphere rows -> ncclAllReduce over the one rank -> unpack: F, T, O, counter unchanged
    P, rU, pf = 128, 30, 2
    F = torch.randn((P, P, P // 2 + 1, 2), device=dev, generator=g)
    T = torch.rand((P, P, P // 2 + 1), device=dev, generator=g)
    O = torch.tensor([1.5, -2.0, 3.25], dtype=torch.float64, device=dev)
    cnt = torch.tensor([42], dtype=torch.int32, device=dev)
    F0, T0 = F.clone(), T.clone()
    need = capi.load().thx_reco_allreduce_workspace(P, rU, pf)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    capi.call("thx_reco_allreduce", comm.handle, ptr(F), ptr(T), ptr(O), ptr(cnt), P, rU, pf, ptr(ws), stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(F, F0) and torch.equal(T, T0) and cnt.item() == 42 and O.tolist() == [1.5, -2.0, 3.25]
    # the packed buffer holds exactly the sphere voxels: pack, wipe the volumes, unpack -> inside restored, outside zero
    nvox = C.c_long(0)
    capi.call("thx_reco_sphere_pack_dev", ptr(F), ptr(T), P, rU, pf, ptr(ws), 0, C.byref(nvox), stream_ptr())
    F.zero_(); T.zero_()
    capi.call("thx_reco_sphere_pack_dev", ptr(F), ptr(T), P, rU, pf, ptr(ws), 1, None, stream_ptr())
    ax = torch.fft.fftfreq(P, d=1.0 / P, device=dev)
    r2 = ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :P // 2 + 1] ** 2
    R = rU * pf + 2
    ii = torch.arange(P // 2 + 1, device=dev, dtype=ax.dtype)
    r2 = ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ii[None, None, :] ** 2
    row_len = torch.floor(torch.sqrt((R * R - ax[:, None] ** 2 - ax[None, :] ** 2).clamp_min(0))) + 1
    inside = (ii[None, None, :] < row_len[:, :, None]) & ((ax[:, None] ** 2 + ax[None, :] ** 2) <= R * R)[:, :, None]
    assert int(inside.sum()) == nvox.value
    assert bool((r2 <= (R - 1) ** 2)[inside.logical_not()].logical_not().all())      # every voxel within R - 1 travels
    assert torch.equal(T[inside], T0[inside]) and torch.equal(F[inside], F0[inside])
    assert T[~inside].abs().max().item() == 0 and F[~inside].abs().max().item() == 0
    comm.close()


def test_cpp_iteration_driver(dev, tmp_path):
    """tests/cpp/iteration.cpp: torch-free C++ over the C ABI -- synthetic particles, thx_refine_create ... iterate x 2,
    half-map FSC and agreement with the generating map; with >= 2 GPUs it forks two ranks that reduce over RCCL"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "iteration")
    libdir = os.path.join(root, "thunder_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "iteration.cpp"), "-o", exe, "-L" + libdir,
                           "-lthunder_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
    print(out.stdout)


def test_cpp_iteration_driver_two_ranks(dev, tmp_path):
    """the same torch-free C++ program as TWO ranks (one half-set each) sharing the one GPU: THX_COMM_TRANSPORT=shm makes the id it
    draws name a shared-memory segment (thx_comm.hip's test-only transport; RCCL refuses two ranks per device), the id travels
    through a pipe as the reference's travels through MPI_Bcast (gpu/src/cuthunder.cu:4192-4206), and every rank runs the unchanged
    multi-rank branches of thx_refine_iterate (sigma tables, norm vector, half-map broadcasts) from C++"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "iteration2")
    libdir = os.path.join(root, "thunder_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "iteration.cpp"), "-o", exe, "-L" + libdir,
                           "-lthunder_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, THX_COMM_TRANSPORT="shm", THX_COMM_SHM_TIMEOUT_S="240")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "two ranks on one GPU" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
    print(out.stdout)


def test_cpp_classification_driver(dev, tmp_path):
    """tests/cpp/classify.cpp: torch-free C++ over the C ABI -- K = 2 synthetic references, images on the scanned grid,
    thx_refine_create (nK = 2, THX_SEARCH_GLOBAL) ... thx_refine_iterate (scan, class, support points, local phases, sigma update,
    multi-reference insertion, two reconstructions per class and half, per-class FSC, averaging, refresh), then a local-search
    iteration: classes recovered, every class map agrees with its own generating map"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "classify")
    libdir = os.path.join(root, "thunder_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "classify.cpp"), "-o", exe, "-L" + libdir,
                           "-lthunder_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
    print(out.stdout)


def test_native_driver_stop_rule(dev):
    """maxPhase > nPhase: the reference's per-image stop rule inside the native driver (src/Optimiser.cpp:1510-1615): every
    image runs at least 5 phases (indices 0-4: the first check, after phase 3, always finds room), stopped images skip the
    later launches through the device mask, and the iteration still converges"""
    from thunder_amd.native import NativeRefine
    from thunder_amd.refine import RefineShard
    N, n = 64, 500
    sh = RefineShard(N, n, dev, mReco=20, batch=128, allocate=False)
    nat = NativeRefine(sh, max_phase=12)
    nat.reset()
    fsc = nat.iterate(timed=True)
    st = nat.stats()
    nP = nat.fetch(nat.view().nP, np.int32, (n,))
    ran = np.where(nP > 0, nP + 1, 12)              # images that never stopped ran all 12 phases
    assert ran.min() >= 5 and ran.max() <= 12
    assert st.imagePhases == int(ran.sum()), (st.imagePhases, int(ran.sum()))
    assert 5 * n <= st.imagePhases < 12 * n and (nP > 0).mean() > 0.5
    assert np.all(fsc[1:4] > 0.9), fsc[:10]     # (the core-mask corrected curve, compareTwoHemispheres with _coreFSC)
    fixed = NativeRefine(sh)                         # the fixed-work iteration on the same particles
    fixed.reset()
    fixed.iterate()
    assert fixed.stats().imagePhases == 3 * n
    nat.close(); fixed.close()


def test_insertion_session_ranks_add_up_bitwise(dev, knob_env):
    """the half-set reduce on the 64-bit fixed-point accumulators (thx_insert_scale_dev -> thx_insert_accumulate_dev ->
    thx_reco_allreduce_acc -> thx_insert_finish_dev): two 'ranks' holding different particles of a half, working in the common
    quanta, sum to bit for bit the F / T one rank accumulates over all of them, in any batching -- checked on one GPU by
    adding the two ranks' accumulators as integers (what ncclSum over ncclInt64 does); the real collective runs on a forced
    one-rank communicator (identity) through the same pack / unpack of the sphere rows."""
    from thunder_amd import capi, ops, synth
    from thunder_amd.capi import ptr, stream_ptr
    from thunder_amd.native import Comm
    from thunder_amd.refine import pixel_list
    rng = np.random.default_rng(31)
    N, P, nImg, mReco = 64, 128, 90, 24
    rU = N // 2 - 2
    pl = pixel_list(N, rU, 0)
    nPxl = pl["nPxl"]
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dat = T_((rng.normal(size=(nImg, nPxl)) + 1j * rng.normal(size=(nImg, nPxl))).astype(np.complex64) * rng.uniform(0.01, 30, size=(nImg, 1)).astype(np.float32))
    ctf = T_(rng.uniform(-1, 1, size=(nImg, nPxl)).astype(np.float32))
    w = T_((rng.uniform(0.2, 1.0, size=nImg) / mReco).astype(np.float32))
    q0 = synth.random_quats(nImg, rng)
    quat = synth.perturb_quats(q0, mReco, 0.03, rng)
    quat[:, 1::2] = quat[:, 0:-1:2]                       # repeated support points, as a resampled filter gives
    rot = ops.rotmat(T_(quat.reshape(-1, 4))).reshape(nImg, mReco, 9)
    trn = T_(rng.normal(0, 1.5, size=(nImg, mReco, 2)))
    iCol, iRow = T_(pl["iCol"]), T_(pl["iRow"])
    L = capi.load()
    nacc = L.thx_insert_acc_bytes(P, 1) // 8
    bounds = torch.empty((nImg, 2), dtype=torch.float32, device=dev)
    gexp = torch.zeros(2, dtype=torch.int32, device=dev)
    capi.call("thx_insert_bounds_dev", ptr(bounds), ptr(dat), ptr(ctf), nPxl, nImg, stream_ptr())
    capi.call("thx_insert_scale_dev", ptr(gexp), ptr(bounds), ptr(w), nImg, mReco, 0, nImg, None, stream_ptr())

    def accumulate(acc, lo, hi):
        capi.call("thx_insert_accumulate_dev", ptr(acc), ptr(gexp), ptr(bounds[lo:hi]), None, None, P, 1, ptr(dat[lo:hi]), ptr(ctf[lo:hi]),
                  ptr(w[lo:hi]), ptr(rot[lo:hi]), ptr(trn[lo:hi]), None, None, None, None, 0, 1.32, ptr(iCol), ptr(iRow), 2, nPxl, mReco, N,
                  hi - lo, stream_ptr())

    def finish(acc):
        F = torch.zeros((P, P, P // 2 + 1), dtype=torch.complex64, device=dev)
        Tt = torch.zeros((P, P, P // 2 + 1), dtype=torch.float32, device=dev)
        capi.call("thx_insert_finish_dev", ptr(F), ptr(Tt), ptr(acc), ptr(gexp), P, 1, stream_ptr())
        return F, Tt
    one = torch.zeros(nacc, dtype=torch.int64, device=dev)
    accumulate(one, 0, nImg)
    F1, T1 = finish(one)
    # two ranks (uneven shares), each in two batches, summed as integers
    a, b = torch.zeros(nacc, dtype=torch.int64, device=dev), torch.zeros(nacc, dtype=torch.int64, device=dev)
    accumulate(a, 0, 20); accumulate(a, 20, 37)
    accumulate(b, 37, 80); accumulate(b, 80, nImg)
    F2, T2 = finish(a + b)
    assert torch.equal(F1, F2) and torch.equal(T1, T2)
    # the one-call form uses the same session
    F3 = torch.zeros_like(F1); T3 = torch.zeros_like(T1)
    ops.insert(F3, T3, P, dat, ctf, w, rot, trn, iCol, iRow, 2, N)
    assert torch.equal(F1, F3) and torch.equal(T1, T3)
    # thx_reco_allreduce_acc on a forced one-rank communicator: pack sphere rows -> ncclAllReduce(ncclInt64) -> unpack = identity
    # inside the sphere; nothing the insertion wrote lies outside it
    knob_env("THX_COMM_FORCE", "1")
    comm = Comm(0, 1, lambda uid: uid)
    ws = torch.empty(L.thx_reco_allreduce_acc_workspace(P, rU, 2), dtype=torch.uint8, device=dev)
    red = one.clone()
    capi.call("thx_reco_allreduce_acc", comm.handle, ptr(red), None, None, P, rU, 2, ptr(ws), stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(red, one)
    comm.close()


def test_iteration_is_bit_reproducible(dev):
    """two runs of the native driver on the same particles give the same bits: support points, sigma tables, inserted F / T
    (fixed-point sums), FSC, half maps, the refreshed projector -- whatever the scheduling.  (The gridding loop's stop rule
    amplifies a last-bit difference into another round count, tests/test_iteration_cpu.py, so nothing before it may jitter.)"""
    from thunder_amd.native import NativeRefine
    from thunder_amd.refine import RefineShard
    N, n = 64, 300
    sh = RefineShard(N, n, dev, mReco=20, batch=64, allocate=False, snr=0.2)
    P = 2 * N
    runs = []
    for rep in range(2):
        nat = NativeRefine(sh)
        nat.reset()
        snap = []
        for it in range(2):
            fsc = nat.iterate()
            v = nat.view()
            off, topR, topT = nat.state()
            snap.append(dict(fsc=fsc.copy(), r=nat.fetch(v.r, np.float64, (n, sh.mLR, 4)), t=nat.fetch(v.t, np.float64, (n, sh.mLT, 2)),
                             sig=nat.fetch(v.sig, np.float32, (2, sh.nGroup, N // 2 - 1)),
                             F=nat.fetch(v.F, np.complex64, (2, P, P, P // 2 + 1)), T=nat.fetch(v.T, np.float32, (2, P, P, P // 2 + 1)),
                             maps=np.stack([nat.map(h).cpu().numpy() for h in (0, 1)]),
                             vols=nat.fetch(v.vols, np.complex64, (2, P, P, P // 2 + 1)), off=off.cpu().numpy(),
                             rounds=list(nat.stats().lastRounds)))
            if rep == 0 and it == 0:   # perturb the scheduling of the second run: other work in flight
                s2 = torch.cuda.Stream()
                with torch.cuda.stream(s2):
                    junk = torch.randn(1 << 26, device=dev).cumsum(0)
                s2.synchronize()
                del junk
        runs.append(snap)
        nat.close()
    for it in range(2):
        a, b = runs[0][it], runs[1][it]
        for k in ("r", "t", "sig", "F", "T", "fsc", "maps", "vols", "off"):
            assert np.array_equal(a[k], b[k]), "iteration %d: %s differs between two runs (%g)" % (it + 1, k, np.abs(a[k] - b[k]).max())
        assert a["rounds"] == b["rounds"]


def test_refine_driver_error_paths(dev):
    """the driver's int-status error convention: bad configurations and out-of-order calls return a status and a
    message, leak nothing and leave the handle destroyable; a failed call does not poison the next one"""
    import ctypes as C
    from thunder_amd import capi
    from thunder_amd.capi import RefineConfig, ThxError, ptr, stream_ptr
    from thunder_amd.native import NativeRefine
    from thunder_amd.refine import RefineShard
    sh = RefineShard(32, 40, dev, mReco=8, batch=16, allocate=False)
    nat = NativeRefine(sh)

    def bad(**kw):
        cfg = RefineConfig.from_buffer_copy(bytes(nat.cfg))
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        with pytest.raises(ThxError) as e:
            capi.call("thx_refine_create", C.byref(h), C.byref(cfg), None, None)
        assert not h.value
        return str(e.value)
    assert "search parameters" in bad(mLT=40) and "search parameters" in bad(mReco=0)
    assert "box" in bad(N=33) and "halfOfRank" in bad(halfOfRank=2) and "nHalfA" in bad(nHalfA=1000)
    # a handle without particles refuses to run, and says why
    h = C.c_void_p()
    capi.call("thx_refine_create", C.byref(h), C.byref(nat.cfg), None, None)
    with pytest.raises(ThxError) as e:
        capi.call("thx_refine_iterate", h, None, 0, stream_ptr())
    assert "set_particles" in str(e.value)
    with pytest.raises(ThxError) as e:
        capi.call("thx_refine_reset", h, stream_ptr())
    assert "set_particles" in str(e.value)
    # group ids out of range are rejected before anything is launched; the handle is still usable afterwards
    gid = np.full(40, 99, np.int32)
    with pytest.raises(ThxError) as e:
        capi.call("thx_refine_set_particles", h, ptr(sh.imgOri), ptr(sh.attr), gid.ctypes.data, ptr(sh.pf0["r"]), ptr(sh.pf0["t"]), stream_ptr())
    assert "groupID" in str(e.value)
    gid = np.ascontiguousarray(sh.gid.astype(np.int32))
    capi.call("thx_refine_set_particles", h, ptr(sh.imgOri), ptr(sh.attr), gid.ctypes.data, ptr(sh.pf0["r"]), ptr(sh.pf0["t"]), stream_ptr())
    capi.call("thx_refine_set_reference", h, ptr(sh.ref), stream_ptr())
    capi.call("thx_refine_reset", h, stream_ptr())
    fsc = np.zeros(16, np.float32)
    capi.call("thx_refine_iterate", h, fsc.ctypes.data, 0, stream_ptr())
    assert fsc[0] > 0.99
    capi.call("thx_refine_destroy", h)
    capi.call("thx_refine_destroy", None)      # NULL is a no-op
    nat.close()


def test_release_stream_evicts_scratch_and_plans(dev):
    """thx_release_stream: work on a side stream (scratch buffers, 2-D and 3-D hipFFT plans are cached per stream), release it,
    destroy it; a new stream -- possibly with the recycled handle value -- works and gives the same results"""
    from thunder_amd import capi, ops
    N = 32
    g = torch.Generator(device=dev).manual_seed(3)
    rl = torch.randn((N, N, N), device=dev, generator=g)
    img = torch.view_as_complex(torch.randn((6, N, N // 2 + 1, 2), device=dev, generator=g)).contiguous()
    want_ft = ops.fft3d_fw(rl).clone()
    want_img = img.clone()
    ops.remask(want_img, 12.0)
    torch.cuda.synchronize()
    for rep in range(3):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ft = ops.fft3d_fw(rl)
            im = img.clone()
            ops.remask(im, 12.0)
        s.synchronize()
        assert torch.equal(ft, want_ft) and torch.equal(im, want_img)
        capi.call("thx_release_stream", s.cuda_stream)
        del s
    capi.call("thx_release_stream", capi.stream_ptr())     # releasing the default stream's caches is harmless: they are rebuilt
    assert torch.equal(ops.fft3d_fw(rl), want_ft)

def test_native_norm_correction(dev):
    """Optimiser::normCorrection inside the native iteration (thx_refine_config.normCorrection): nothing in the first iteration; in
    the second the caller's _imgOri stack and the driver's _img are multiplied image by image by sqrt(median / norm_l) -- checked
    bit for bit from the norms the driver exposes -- with rNorm = min(rU, resP(previous FSC, 0.75)), and the refinement still
    converges.  (The kernels have their own oracle parity test, tests/test_next_gpu.py::test_norm_correction.)"""
    from thunder_amd.native import NativeRefine
    from thunder_amd.refine import RefineShard
    N, n = 64, 96
    sh = RefineShard(N, n, dev, mLR=40, mLT=5, nPhase=2, mReco=16, batch=64, particle_filter=True, allocate=False, snr=0.5)
    nat = NativeRefine(sh, norm_correction=True)
    nat.reset()
    ori0 = sh.imgOri.clone()
    fsc1 = nat.iterate()
    st = nat.stats()
    assert st.normMedian == 0.0 and torch.equal(sh.imgOri, ori0)            # (_iter != 0)
    fsc2 = nat.iterate()
    st = nat.stats()
    v = nat.view()
    norm = nat.fetch(v.norm, np.float32, (n,))
    assert norm.min() > 0 and st.normMedian > 0
    # resP(fsc, 0.75, 1, 1, false) on the FSC of the first iteration, capped at rU = N / 2 - 2
    res = 1
    while res < N // 2 - 2 and fsc1[res] >= 0.75:
        res += 1
    assert st.normRadius == min(N // 2 - 2, res - 1)
    srt = np.sort(norm)
    idx = 0.5 * (n - 1)
    lhs = int(idx)
    med = np.float32((1 - (idx - lhs)) * np.float64(srt[lhs]) + (idx - lhs) * np.float64(srt[lhs + 1]))
    assert np.float32(st.normMedian) == med
    f = np.sqrt(med / norm).astype(np.float32)
    want = (ori0.cpu().numpy() * f[:, None, None]).astype(np.complex64)
    assert np.array_equal(sh.imgOri.cpu().numpy(), want)
    assert 0.2 < f.min() and f.max() < 5.0
    assert np.all(np.isfinite(fsc2)) and fsc2[1] >= 0.95 and fsc2[2] >= 0.9      # (96 noisy particles: the half maps agree to shell 3)
    nat.close()
