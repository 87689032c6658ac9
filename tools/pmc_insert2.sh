#!/bin/bash
# profiling aid: second set of PMC counters for k_insert_win (LDS queues, instruction fetch, vector-memory mix)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ins2
rm -rf $OUT; mkdir -p $OUT
DBGS=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -o p -- python tools/insert_probe.py 1024 > $OUT/p1.log 2>&1
DBGS=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/p2 -o p -- python tools/insert_probe.py 1024 > $OUT/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for p in ("p1","p2"):
    fs = glob.glob("gpurun_out/pmc_ins2/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs: print(p, "no output"); continue
    agg = collections.defaultdict(float); n=set()
    for r in csv.DictReader(open(fs[0])):
        if "k_insert_win" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    print(p, "dispatches", len(n), {c: "%.3g" % (v / max(1,len(n))) for c, v in agg.items()})
PY
