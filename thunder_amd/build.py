"""Builds libthunder_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libthunder_amd.so")
SOURCES = ["thx_estep.hip", "thx_mstep.hip", "thx_insert_sort.hip", "thx_reco.hip", "thx_next.hip", "thx_io.hip", "thx_pf.hip", "thx_iface.hip", "thx_host.hip", "thx_comm.hip", "thx_refine.hip", "thx_model.hip"]
HEADERS = ["thx_common.h", "thx_insert.h", "thx_fft8.h", "thx_philox.h", os.path.join("..", "..", "include", "thunder_amd.h")]

# -ffp-contract=off: see thx_common.h (bit-identical trilinear arithmetic); fused ops are written out as fmaf().
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + os.environ.get("THX_EXTRA_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(o)
        # one object per source: only what changed (or sits below a changed header) is compiled again
        if not force and not os.environ.get("THX_EXTRA_FLAGS") and os.path.exists(o) and \
                os.path.getmtime(o) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, s))):
            continue
        cmd = [_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on %s" % s)
        if verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lhipfft", "-lrccl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


HARNESS_SRC = os.path.join(HERE, "..", "integration", "expectationG_harness.cpp")
HARNESS = os.path.join(LIBDIR, "libthx_harness.so")


def build_harness(force=False):
    """integration/expectationG_harness.cpp -> thunder_amd/lib/libthx_harness.so: the CALLER side of the reference's plug-in surface
    (Optimiser::expectationG's OpenMP loop) over the C ABI -- test / bench infrastructure, plain g++ (no device code), linked against
    libthunder_amd.so, never part of it."""
    lib = build()
    src = os.path.abspath(HARNESS_SRC)
    if not force and os.path.exists(HARNESS) and os.path.getmtime(HARNESS) > max(os.path.getmtime(src), os.path.getmtime(lib)):
        return HARNESS
    cmd = ["g++", "-O2", "-std=c++14", "-fopenmp", "-shared", "-fPIC", "-Wall", "-I" + os.path.join(HERE, "..", "include"), src, "-o", HARNESS,
           "-L" + LIBDIR, "-lthunder_amd", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return HARNESS


def build_tools(force=False):
    """the two micro-benchmarks bench.py's in-run PMC pass needs (tools/pmc_traffic.sh: the calibration dispatches of known byte count and
    the LDS atomic rates): built here so that they travel with the tree; the script builds them itself where they are missing"""
    tools = os.path.join(HERE, "..", "tools")
    out = []
    for exe, src in (("pmc_calib", "pmc_calib.hip"), ("lds_atomic_bench", os.path.join("probes", "lds_atomic_bench.hip"))):
        e, s_ = os.path.abspath(os.path.join(tools, exe)), os.path.abspath(os.path.join(tools, src))
        if force or not os.path.exists(e) or os.path.getmtime(e) < os.path.getmtime(s_):
            subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-o", e, s_])
        out.append(e)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_harness(force="--force" in sys.argv))
