# BASELINE configs[4] on one GPU: 20 000 synthetic 512^3 particles (P = 1024: 4.3 GB projector, 34 GB cell-packed per half)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03
mkdir -p $OUT
N=${PARTICLES:-20000}
timeout 1500 python bench.py --box 512 --particles $N --batch 2500 --steps 1 --warmup 1 --cpu-particles 64 > $OUT/bench_box512.json 2> $OUT/bench_box512.err
tail -c 2500 $OUT/bench_box512.json; tail -5 $OUT/bench_box512.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats512 -- timeout 900 python bench.py --box 512 --particles $N --batch 2500 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_box512_under_rocprof.json 2> $OUT/stats512.err
cp $(find $OUT/stats512 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_box512.csv
rm -rf $OUT/stats512
head -6 $OUT/kernel_stats_box512.csv | cut -c1-170
