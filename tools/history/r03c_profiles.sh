# round-3c measurement suite (HEAD: normCorrection in the iteration, THX_BIN_KEEP, native classification driver) (run on the GPU box; outputs under gpurun_out/r03c, copied into profiles/ by hand)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c
mkdir -p $OUT
python bench.py > $OUT/bench_100k.json 2> $OUT/bench_100k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_100k_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_100k.csv
rm -rf $OUT/stats
# HBM traffic of the hot kernels in the bench's OWN view-ordered run (separate --pmc passes, calibrated in the same passes)
PMC_CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline" THX_PROBE_PARTICLES=20000 bash tools/pmc_traffic.sh > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_traffic/summary.json $OUT/pmc_traffic_summary.json
cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
cp gpurun_out/pmc_traffic/lds_atomic_bench.txt $OUT/lds_atomic_bench.txt
rm -rf gpurun_out/pmc_traffic
tail -c 1500 $OUT/bench_100k.json; echo; head -8 $OUT/kernel_stats_100k.csv | cut -c1-170; cat $OUT/pmc_traffic.json
# BASELINE configs[3] through the native driver, with its kernel statistics
OUT=gpurun_out/r03c
timeout 900 python bench.py --classification > $OUT/bench_classification_k4.json 2> $OUT/bench_classification_k4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statsK -- timeout 600 python bench.py --classification --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_classification_k4_under_rocprof.json 2> $OUT/statsK.err
cp $(find $OUT/statsK -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_classification_k4.csv; rm -rf $OUT/statsK
tail -c 700 $OUT/bench_classification_k4.json; echo; head -6 $OUT/kernel_stats_classification_k4.csv | cut -c1-150
