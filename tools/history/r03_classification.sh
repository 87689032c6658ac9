# BASELINE configs[3] on one GPU: one whole K = 4 classification iteration (GPU box; outputs under gpurun_out/r03)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03; mkdir -p $OUT
timeout 900 python bench.py --classification > $OUT/bench_classification_k4.json 2> $OUT/bench_classification_k4.err
tail -c 1800 $OUT/bench_classification_k4.json; tail -3 $OUT/bench_classification_k4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statsK -- timeout 600 python bench.py --classification --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_classification_k4_under_rocprof.json 2> $OUT/statsK.err
cp $(find $OUT/statsK -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_classification_k4.csv; rm -rf $OUT/statsK
timeout 900 python bench.py --classification --scan-images 50000 --no-cpu-baseline > $OUT/bench_classification_k4_50k.json 2> $OUT/bench_classification_k4_50k.err
tail -c 900 $OUT/bench_classification_k4_50k.json; tail -3 $OUT/bench_classification_k4_50k.err
