# round-end measurement suite (run on the GPU box; outputs under gpurun_out/final, copied into profiles/ by hand):
# bench line of the metric's workload, kernel statistics of the same command, configs[1] line, global-scan line, PMC traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_100k.json 2> $OUT/bench_100k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_100k_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_100k.csv
rm -rf $OUT/stats
python bench.py --particles 10000 --no-cpu-baseline > $OUT/bench_10k.json 2>> $OUT/bench_100k.err
python bench.py --classification --steps 2 --warmup 1 > $OUT/bench_global_scan.json 2>> $OUT/bench_100k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scanstats -- python bench.py --classification --steps 2 --warmup 1 > /dev/null 2>> $OUT/stats.err
cp $(find $OUT/scanstats -name "*kernel_stats.csv" | head -1) $OUT/global_scan_kernel_stats.csv
rm -rf $OUT/scanstats
bash tools/pmc_traffic.sh > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_traffic/summary.json $OUT/pmc_traffic_summary.json
cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic.json
cp gpurun_out/pmc_traffic/pmc_calib_unprofiled.txt $OUT/pmc_calib_timings.txt
cp gpurun_out/pmc_traffic/lds_atomic_bench.txt $OUT/lds_atomic_bench.txt
if [ "${LONG_CPU:-0}" = "1" ]; then python bench.py --steps 1 --warmup 1 --cpu-particles 2048 > $OUT/bench_100k_cpu2048.json 2>> $OUT/bench_100k.err; fi
tail -c 700 $OUT/bench_100k.json; echo; head -8 $OUT/kernel_stats_100k.csv | cut -c1-170; cat $OUT/pmc_traffic.json
