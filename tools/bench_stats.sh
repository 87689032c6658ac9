# kernel statistics of a short bench run under rocprofv3 (GPU box): bash tools/bench_stats.sh [particles] [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/stats
P=${1:-20000}; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -- python bench.py --particles $P --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_stats.json 2> gpurun_out/stats.err
cp $(find gpurun_out/stats -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats.csv; rm -rf gpurun_out/stats
python - <<PY
import csv, json
d = json.loads(open('gpurun_out/bench_stats.json').read().strip().splitlines()[-1]); print(round(d['value'], 1), d['stages_ms_per_step'])
for r in list(csv.DictReader(open('gpurun_out/kernel_stats.csv')))[:10]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), "%9.1f ms total %8.3f avg" % (float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6))
PY
