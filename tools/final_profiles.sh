# round-end measurement suite: bench line, kernel statistics, PMC traffic (run on the GPU box; outputs under gpurun_out/final)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --fixed-support --no-cpu-baseline > $OUT/bench_fixed_support.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
bash tools/pmc_traffic.sh > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_traffic/summary.json $OUT/pmc_traffic_summary.json
cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
tail -c 600 $OUT/bench.json; echo; head -8 $OUT/kernel_stats.csv | cut -c1-160; cat $OUT/pmc_traffic.json
