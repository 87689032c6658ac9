#!/bin/bash
# HBM traffic of the hot kernels from the TCC counters, collected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes with --kernel-trace only.  The unit of FETCH_SIZE depends on the
# access shape on gfx950, so every pass also profiles tools/pmc_calib: 1 GiB dispatches of known byte count in three
# shapes (wide streaming copy, scattered 64-byte cells = the E-step's packed gathers, scattered 16-byte pairs).  A third
# pass collects the L2's request / hit / miss counts for a cross-check (misses x 128-byte lines).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$(readlink -f "$0")")/..}"
OUT=gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
[ -x tools/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip
[ -x tools/lds_atomic_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/lds_atomic_bench tools/probes/lds_atomic_bench.hip
tools/lds_atomic_bench > $OUT/lds_atomic_bench.txt 2>&1
tools/pmc_calib > $OUT/pmc_calib_unprofiled.txt 2>&1
# PMC_CMD (optional): the command whose kernels are counted -- default the 1 024-image probe; round 3 counts the bench's own
# view-ordered 100 k-particle iteration: PMC_CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline" THX_PROBE_PARTICLES=20000
# (THX_PROBE_PARTICLES / 2 = images per launch of the counted kernels, for the per-image figures)
CMD="${PMC_CMD:-python tools/probes/traffic_probe.py}"
for pass in fetch:FETCH_SIZE write:WRITE_SIZE l2:"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
    name=${pass%%:*}; ctr=${pass#*:}
    [ "$name" = l2 ] && [ "${PMC_SKIP_L2:-0}" = 1 ] && continue      # (bench.py's in-run pass: FETCH_SIZE and WRITE_SIZE only)
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/cal_$name -o p -- tools/pmc_calib > $OUT/cal_$name.log 2>&1
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1
done
python tools/pmc_traffic_parse.py $OUT
