# SQ / GRBM counters of the scan contraction (one pass; kernel-trace only, as the pool requires)
rm -rf /root/repo/gpurun_out/scanpmc; cd /tmp && export TMPDIR=/tmp
THX_SCAN=${1:-t22} rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /root/repo/gpurun_out/scanpmc -- python /root/repo/bench.py --classification --steps 1 --warmup 1 > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/scanpmc/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:40]
    if 'k_scan_gemm<true' not in r['Kernel_Name']: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print('   ', c, v / cnt[(k, c)])
PY
