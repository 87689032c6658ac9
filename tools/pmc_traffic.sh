#!/bin/bash
# HBM traffic of the hot kernels from the TCC counters, collected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes with --kernel-trace only, plus a calibration dispatch of known
# byte count (a 1 GiB float4 device copy) in the same passes to fix the unit / gfx950 factor for this access width.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic
mkdir -p $OUT
CMD="python tools/traffic_probe.py"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
python tools/pmc_traffic_parse.py $OUT
