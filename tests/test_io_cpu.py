"""MRC / .thu host-side I/O of the C ABI against an independent numpy statement of the
formats (oracle.mrc_read / mrc_images / mrc_volume).  No GPU needed: these entry points are host code."""
import ctypes as C

import numpy as np
import pytest


def _write_mrc(path, data, mode, nsymbt=0):
    """independent MRC writer (numpy): header words per include/Image/MRCHeader.h"""
    nz, ny, nx = data.shape
    head = np.zeros(256, np.int32)
    head[0:4] = (nx, ny, nz, mode)
    head[7:10] = (nx, ny, nz)
    head[23] = nsymbt
    raw = head.tobytes()
    raw = raw[:208] + b"MAP " + raw[212:]
    with open(path, "wb") as f:
        f.write(raw)
        f.write(b"\x07" * nsymbt)
        f.write(data.astype({0: np.int8, 1: np.int16, 2: np.float32}[mode]).tobytes())


@pytest.fixture(scope="module")
def lib():
    from thunder_amd import capi
    return capi


@pytest.mark.parametrize("mode,nsymbt", [(0, 0), (1, 80), (2, 0), (2, 160)])
def test_mrc_read_images_and_volume(tmp_path, oracle, lib, mode, nsymbt):
    rng = np.random.default_rng(mode + nsymbt)
    nz, ny, nx = 6, 8, 10
    data = rng.integers(-100, 100, size=(nz, ny, nx)).astype(np.float32)
    if mode == 2:
        data += rng.standard_normal(data.shape).astype(np.float32)
    p = str(tmp_path / "s.mrcs")
    _write_mrc(p, data, mode, nsymbt)
    info = [C.c_int() for _ in range(5)]
    lib.call("thx_mrc_info", p.encode(), *[C.byref(x) for x in info])
    assert [x.value for x in info] == [nx, ny, nz, mode, nsymbt]
    want = oracle.mrc_images(p)
    got = np.zeros((nz, ny, nx), np.float32)
    lib.call("thx_mrc_read_images", p.encode(), 0, nz, got.ctypes.data)
    assert np.array_equal(got, want)
    part = np.zeros((2, ny, nx), np.float32)
    lib.call("thx_mrc_read_images", p.encode(), 3, 2, part.ctypes.data)   # "000004@s.mrcs" -> slice index 3
    assert np.array_equal(part, want[3:5])
    vol = np.zeros((nz, ny, nx), np.float32)
    lib.call("thx_mrc_read_volume", p.encode(), vol.ctypes.data)
    assert np.array_equal(vol, oracle.mrc_volume(p))
    # the centre sample of the file is the origin sample in memory
    assert vol[0, 0, 0] == data.astype({0: np.int8, 1: np.int16, 2: np.float32}[mode])[nz // 2, ny // 2, nx // 2]
    with pytest.raises(lib.ThxError):
        lib.call("thx_mrc_read_images", p.encode(), nz - 1, 2, part.ctypes.data)
    with pytest.raises(lib.ThxError):
        lib.call("thx_mrc_info", (p + ".missing").encode(), *[C.byref(x) for x in info])


def test_mrc_write_roundtrip(tmp_path, oracle, lib):
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((8, 8, 8)).astype(np.float32)
    p = str(tmp_path / "v.mrc")
    lib.call("thx_mrc_write_volume", p.encode(), vol.ctypes.data, 8, 8, 8, 1.32)
    raw, (nx, ny, nz, mode, nsymbt) = oracle.mrc_read(p)
    assert (nx, ny, nz, mode, nsymbt) == (8, 8, 8, 2, 0)
    assert np.array_equal(oracle.mrc_volume(p), vol)
    with open(p, "rb") as f:
        head = f.read(1024)
    assert np.allclose(np.frombuffer(head, np.float32, 3, 40), 8 * 1.32) and head[208:212] == b"MAP "
    assert list(np.frombuffer(head, np.int32, 3, 64)) == [1, 2, 3]
    st = rng.standard_normal((5, 6, 6)).astype(np.float32)
    p2 = str(tmp_path / "st.mrcs")
    lib.call("thx_mrc_write_stack", p2.encode(), st.ctypes.data, 6, 5, 2.0)
    assert np.array_equal(oracle.mrc_images(p2), st)
    back = np.zeros_like(st)
    lib.call("thx_mrc_read_images", p2.encode(), 0, 5, back.ctypes.data)
    assert np.array_equal(back, st)


def test_thu_table(tmp_path, lib):
    from thunder_amd.capi import CtfAttr
    rows = []
    rng = np.random.default_rng(9)
    n = 7
    vals = rng.uniform(0.1, 2.0, size=(n, 27))
    for l in range(n):
        v = vals[l]
        cols = ["%18.9f" % x for x in (300000.0, 15000 + 100 * l, 15100 + 100 * l, v[3], 2.7e7, 0.1, 0.0)]
        cols += ["%06d@stack_%d.mrcs" % (l + 1, l % 2), "mic_%d.mrc" % (l % 3), "%18.9f" % v[9], "%18.9f" % v[10]]
        cols += ["%6d" % (l % 3 + 1), "%6d" % 0] + ["%18.9f" % x for x in v[13:27]]
        rows.append(" ".join(cols))
    text = "# a comment line\n\n" + "\n".join(rows[:3]) + "\n   \n  # indented comment\n" + "\n".join(rows[3:]) + "\n"
    p = str(tmp_path / "particles.thu")
    open(p, "w").write(text)
    cnt, grp = C.c_int(), C.c_int()
    lib.call("thx_thu_count", p.encode(), C.byref(cnt), C.byref(grp))
    assert (cnt.value, grp.value) == (n, 3)
    ctf = (CtfAttr * n)()
    paths = C.create_string_buffer(n * 64)
    gid = np.zeros(n, np.int32); cid = np.zeros(n, np.int32)
    quat = np.zeros((n, 4)); tran = np.zeros((n, 2)); stdT = np.zeros((n, 2)); dfac = np.zeros(n); score = np.zeros(n)
    lib.call("thx_thu_load", p.encode(), n, C.cast(ctf, C.c_void_p), C.cast(paths, C.c_void_p), 64, gid.ctypes.data,
             cid.ctypes.data, quat.ctypes.data, tran.ctypes.data, stdT.ctypes.data, dfac.ctypes.data, score.ctypes.data)
    assert [ctf[l].defocusU for l in range(n)] == [np.float32(15000 + 100 * l) for l in range(n)]
    assert ctf[0].voltage == 300000.0 and abs(ctf[2].defocusTheta - np.float32(vals[2, 3])) < 1e-6
    got_paths = [paths.raw[l * 64:(l + 1) * 64].split(b"\0")[0].decode() for l in range(n)]
    assert got_paths == ["%06d@stack_%d.mrcs" % (l + 1, l % 2) for l in range(n)]
    assert gid.tolist() == [l % 3 + 1 for l in range(n)] and cid.tolist() == [0] * n
    r9 = lambda a: np.round(a, 9)
    assert np.allclose(quat, r9(vals[:, 13:17]), atol=1e-9) and np.allclose(tran, r9(vals[:, 20:22]), atol=1e-9)
    assert np.allclose(stdT, r9(vals[:, 22:24]), atol=1e-9) and np.allclose(dfac, r9(vals[:, 24]), atol=1e-9)
    assert np.allclose(score, r9(vals[:, 26]), atol=1e-9)
    # short table (CTF + paths only): defaults for the pose columns
    open(p, "w").write("\n".join(" ".join(r.split()[:9]) for r in rows) + "\n")
    lib.call("thx_thu_load", p.encode(), n, None, None, 0, gid.ctypes.data, None, quat.ctypes.data, None, None,
             dfac.ctypes.data, None)
    assert np.array_equal(quat, np.tile([1.0, 0, 0, 0], (n, 1))) and np.array_equal(dfac, np.ones(n))
    with pytest.raises(lib.ThxError):
        lib.call("thx_thu_load", p.encode(), n + 1, None, None, 0, gid.ctypes.data, None, None, None, None, None, None)


def test_thu_write_roundtrip(tmp_path, lib):
    """Optimiser::saveDatabase's table (src/Optimiser.cpp:8217-8416): written by thx_thu_write (two 'ranks': write then
    append), read back by the library's own loaders and by a plain whitespace parser the way Database::reGenDatabase
    (src/Database.cpp:40-100) sees it; the line layout equals the reference's fprintf format"""
    from thunder_amd.capi import CtfAttr
    rng = np.random.default_rng(11)
    n = 9
    ctf = (CtfAttr * n)()
    for l in range(n):
        ctf[l].voltage, ctf[l].defocusU, ctf[l].defocusV = 300000.0, 15000.5 + 10 * l, 15100.25 + 10 * l
        ctf[l].defocusTheta, ctf[l].Cs, ctf[l].amplitudeContrast, ctf[l].phaseShift = 0.1 * l, 2.7e7, 0.1, 0.0
    paths = C.create_string_buffer(n * 64)
    mics = C.create_string_buffer(n * 32)
    for l in range(n):
        paths[l * 64:l * 64 + 64] = ("%06d@stack_%d.mrcs" % (l + 1, l % 2)).encode().ljust(64, b"\0")
        mics[l * 32:l * 32 + 32] = ("mic_%03d.mrc" % (l % 4)).encode().ljust(32, b"\0")
    coord = rng.uniform(0, 4000, (n, 2)); gid = (np.arange(n) % 3 + 1).astype(np.int32); cid = (np.arange(n) % 2).astype(np.int32)
    quat = rng.normal(size=(n, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    k123 = rng.uniform(1e-5, 1e-3, (n, 3)); tran = rng.normal(0, 2, (n, 2)); stdT = rng.uniform(0.1, 1, (n, 2))
    dfac = rng.normal(1, 0.01, n); sdf = rng.uniform(0, 0.01, n); score = rng.uniform(0, 1, n)
    p = str(tmp_path / "Meta_Round_001.thu")
    cut = 5

    def part(a, lo, hi):
        return np.ascontiguousarray(a[lo:hi])
    for lo, hi, app in ((0, cut, 0), (cut, n, 1)):
        arrs = [part(x, lo, hi) for x in (coord, gid, cid, quat, k123, tran, stdT, dfac, sdf, score)]
        lib.call("thx_thu_write", p.encode(), app, hi - lo, C.cast(C.byref(ctf, lo * C.sizeof(CtfAttr)), C.c_void_p),
                 C.cast(C.byref(paths, lo * 64), C.c_void_p), 64, C.cast(C.byref(mics, lo * 32), C.c_void_p), 32,
                 *[a.ctypes.data for a in arrs])
    text = open(p).read().splitlines()
    assert text[0] == "#0:VOLTAGE\tFLOAT\t18.9f" and text[13] == "#13QUATERNION_0\tFLOAT\t18.9f" and text[27] == ""
    data = [t for t in text if t.strip() and not t.lstrip().startswith("#")]
    assert len(data) == n and sum(t.startswith("#0:") for t in text) == 2      # every rank writes the description block
    # the reference's line: its format string continues lines inside the literal (21 blanks before columns 7, 11, 13, 17, 20, 24, 26)
    l = 3
    want = ("%18.9f %18.9f %18.9f %18.9f %18.9f %18.9f %18.9f " + " " * 21 + "%s %s %18.9f %18.9f " + " " * 21 + "%6d %6d " + " " * 21
            + "%18.9f %18.9f %18.9f %18.9f " + " " * 21 + "%18.9f %18.9f %18.9f " + " " * 21 + "%18.9f %18.9f %18.9f %18.9f " + " " * 21
            + "%18.9f %18.9f " + " " * 21 + "%18.9f") % (
        ctf[l].voltage, ctf[l].defocusU, ctf[l].defocusV, ctf[l].defocusTheta, ctf[l].Cs, ctf[l].amplitudeContrast, ctf[l].phaseShift,
        "%06d@stack_%d.mrcs" % (l + 1, l % 2), "mic_%03d.mrc" % (l % 4), coord[l, 0], coord[l, 1], gid[l], cid[l], *quat[l], *k123[l],
        *tran[l], *stdT[l], dfac[l], sdf[l], score[l])
    assert data[l] == want
    cols = [t.split() for t in data]
    assert all(len(c) == 27 for c in cols)
    # read back through the library
    cnt, grp = C.c_int(), C.c_int()
    lib.call("thx_thu_count", p.encode(), C.byref(cnt), C.byref(grp))
    assert (cnt.value, grp.value) == (n, 3)
    ctf2 = (CtfAttr * n)(); paths2 = C.create_string_buffer(n * 64); mics2 = C.create_string_buffer(n * 32)
    gid2 = np.zeros(n, np.int32); cid2 = np.zeros(n, np.int32)
    quat2 = np.zeros((n, 4)); tran2 = np.zeros((n, 2)); stdT2 = np.zeros((n, 2)); dfac2 = np.zeros(n); score2 = np.zeros(n)
    coord2 = np.zeros((n, 2)); k2 = np.zeros((n, 3)); sdf2 = np.zeros(n)
    lib.call("thx_thu_load", p.encode(), n, C.cast(ctf2, C.c_void_p), C.cast(paths2, C.c_void_p), 64, gid2.ctypes.data,
             cid2.ctypes.data, quat2.ctypes.data, tran2.ctypes.data, stdT2.ctypes.data, dfac2.ctypes.data, score2.ctypes.data)
    lib.call("thx_thu_load_extra", p.encode(), n, C.cast(mics2, C.c_void_p), 32, coord2.ctypes.data, k2.ctypes.data, sdf2.ctypes.data)
    assert paths2.raw == paths.raw and mics2.raw == mics.raw
    assert np.array_equal(gid2, gid) and np.array_equal(cid2, cid)
    for a, b in ((quat2, quat), (tran2, tran), (stdT2, stdT), (dfac2, dfac), (score2, score), (coord2, coord), (k2, k123), (sdf2, sdf)):
        assert np.allclose(a, b, atol=6e-10, rtol=0)
    assert all(abs(ctf2[l].defocusU - ctf[l].defocusU) < 1e-2 and ctf2[l].Cs == ctf[l].Cs for l in range(n))
    # only the required columns given: neutral values for the rest
    lib.call("thx_thu_write", p.encode(), 0, 2, C.cast(ctf, C.c_void_p), C.cast(paths, C.c_void_p), 64, None, 0, None, None, None, None,
             None, None, None, None, None, None)
    c = [t.split() for t in open(p).read().splitlines() if t.strip() and not t.startswith("#")]
    assert len(c) == 2 and c[0][8] == "mic.mrc" and [float(x) for x in c[0][13:17]] == [1, 0, 0, 0] and float(c[0][24]) == 1


def test_ingestion_oracle_statistics(oracle):
    """the oracle's GSL-style running statistics against numpy on a noisy disc image"""
    O = oracle
    rng = np.random.default_rng(11)
    N, r = 32, 11.0
    jj, ii = np.meshgrid(np.fft.fftfreq(N, 1.0 / N), np.fft.fftfreq(N, 1.0 / N), indexing="ij")
    img = (5.0 + 2.0 * rng.standard_normal((N, N)) + 10.0 * (np.hypot(ii, jj) < 6)).astype(np.float32)
    x = img.copy()
    O.lib().orc_subtract_bg(x.ctypes.data_as(O.c_f), N, C.c_float(r))
    bg = (ii * ii + jj * jj) > r * r
    want = (img - img[bg].mean()) / img[bg].std(ddof=1)
    assert np.abs(x - want).max() < 2e-5
    st = np.zeros(4)
    O.lib().orc_stat_img(st.ctypes.data_as(O.c_d), x.ctypes.data_as(O.c_f), N, C.c_float(r))
    u = np.hypot(ii, jj).astype(np.float32)
    assert abs(st[0] - x[u < r].mean()) < 1e-5
    assert abs(st[1] - np.sqrt((x[bg].astype(np.float64) ** 2).sum() / (bg.sum() - 1))) < 1e-5
    assert abs(st[2] - np.sqrt((x.astype(np.float64) ** 2).sum() / (N * N - 1))) < 1e-5 and abs(st[3] - st[1] ** 2) < 1e-6
    m = np.zeros_like(x)
    O.lib().orc_soft_mask_bg(m.ctypes.data_as(O.c_f), x.ctypes.data_as(O.c_f), N, C.c_float(r), C.c_float(6.0), C.c_float(0))
    assert np.array_equal(m, x * O.soft_mask(N, r, 6.0)) or np.abs(m - x * O.soft_mask(N, r, 6.0)).max() < 1e-6


def test_cpp_io_mirrors(tmp_path, lib):
    """ImageFile / Database mirrors (include/thunder_amd/ImageFile.hpp) compiled with g++ against the C ABI: host only"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "io_rt")
    libdir = os.path.join(root, "thunder_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "io_roundtrip.cpp"), "-o", exe, "-L" + libdir, "-lthunder_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK"), (out.stdout, out.stderr[-2000:])
