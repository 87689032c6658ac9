#!/bin/bash
# profiling aid: PMC counters of the insertion / expectation kernels (separate passes, kernel-trace only)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_r01
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o p -- python tools/insert_probe.py 1024 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAVES --output-format csv -d $OUT/p2 -o p -- python tools/insert_probe.py 1024 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $OUT/p3 -o p -- python tools/insert_probe.py 1024 > $OUT/p3.log 2>&1
python - <<'PY'
import csv, glob, collections
for p in ("p1","p2","p3"):
    fs = glob.glob("gpurun_out/pmc_r01/%s/**/*counter_collection.csv" % p, recursive=True)
    print(p, fs)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    rows = list(csv.DictReader(open(fs[0])))
    for r in rows:
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    seen=set()
    for r in rows:
        key=(r["Kernel_Name"][:40], r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[r["Kernel_Name"][:40]] += 1
    for k in agg:
        if "insert" in k or "expect_local" in k:
            print(k, "dispatches", cnt[k], {c: "%.3g" % (v / cnt[k]) for c, v in agg[k].items()})
PY
