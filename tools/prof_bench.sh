# rocprofv3 kernel statistics of one bench run; prints the top kernels and leaves the csv under gpurun_out/prof_bench
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
cut -d, -f1-5 $f | cut -c1-150 | head -${TOP:-16}
tail -1 $OUT/bench.json | cut -c1-300
