#!/bin/bash
# Round 6, DESIGN Appendix A E13: the local-search kernel's NON-MEMORY floor on its own.  The coherence probe (tools/probes/
# coherence_probe.py: every image on image 0's cloud -- every gather an L2 hit) is run with the default kernel and with the look-ahead
# form (THX_EXPECT_SPLIT = 1000: every sample requested one pixel ahead, two gathers in flight per lane) at 2 and at 1 workgroups per CU,
# (a) timed, (b) under SQ counters (kernel-trace only, one pass): SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
# SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU, SQ_INSTS_VMEM_RD.  Output: gpurun_out/r06_estep_floor.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_floor; rm -rf $OUT; mkdir -p $OUT
N=${N:-4096}
{
for v in "default:0:2" "default:0:1" "lookahead:1000:2" "lookahead:1000:1" "lookahead_near:4:2"; do
    name=${v%%:*}; rest=${v#*:}; split=${rest%%:*}; wg=${rest#*:}
    echo "== $name  THX_EXPECT_SPLIT=$split  workgroups per CU $wg"
    E13=1 WG=$wg THX_EXPECT_SPLIT=$split python tools/probes/coherence_probe.py $N 2>/dev/null | grep -E "^A |^B "
    E13=1 WG=$wg THX_EXPECT_SPLIT=$split rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD \
        --output-format csv -d $OUT/$name$wg -- python tools/probes/coherence_probe.py $N > /dev/null 2> $OUT/$name$wg.err
    python - $OUT/$name$wg <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')
if not fs:
    print('   (no counter file)'); sys.exit(0)
rows = [r for r in csv.DictReader(open(fs[0])) if 'k_expect_local' in r['Kernel_Name']]
# dispatches in order: warm-up run(s) of the shard, then 3 x A, 3 x B -- the LAST three dispatches are B
ids = sorted({int(r['Dispatch_Id']) for r in rows})
for label, sel in (('A (own clouds)', ids[-6:-3]), ('B (one cloud: every gather an L2 hit)', ids[-3:])):
    acc = collections.defaultdict(float)
    for r in rows:
        if int(r['Dispatch_Id']) in sel: acc[r['Counter_Name']] += float(r['Counter_Value']) / len(sel)
    wc = acc.get('SQ_WAVE_CYCLES', 0) or 1
    print('   %-40s' % label + '  '.join('%s %.3g' % (k.replace('SQ_', ''), v) for k, v in sorted(acc.items())))
    print('   %-40s wait_any/wave %.2f  wait_inst/wave %.2f  active/wave %.2f  VALU insts per VMEM_RD %.1f' % ('', acc.get('SQ_WAIT_ANY', 0) / wc, acc.get('SQ_WAIT_INST_ANY', 0) / wc,
          acc.get('SQ_ACTIVE_INST_ANY', 0) / wc, acc.get('SQ_INSTS_VALU', 0) / max(1.0, acc.get('SQ_INSTS_VMEM_RD', 0))))
PY
done
} 2>&1 | tee gpurun_out/r06_estep_floor.txt
rm -rf $OUT
