# SQ counters of the brick-sorted insertion's two kernels (kernel-trace only, as the pool requires); two passes of counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/insertpmc; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --particles 20000 --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- $CMD > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/b -- $CMD > /dev/null 2> $OUT/b.err
python - <<'PY'
import csv, glob, collections
for tag in ('a', 'b'):
    fs = glob.glob('gpurun_out/insertpmc/%s/*/*counter_collection.csv' % tag)
    if not fs:
        print('no counters in pass', tag); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = 'k_bin' if 'k_bin' in r['Kernel_Name'] else ('k_acc' if 'k_acc' in r['Kernel_Name'] else None)
        if k is None: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, d in sorted(acc.items()):
        print(k, '(pass %s; sums over all dispatches / dispatches = per chunk)' % tag)
        for c, v in sorted(d.items()): print('    %-24s %16.0f per dispatch over %d dispatches' % (c, v / n[(k, c)], n[(k, c)]))
PY
