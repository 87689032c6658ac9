# final HEAD record: insertion / classification / native tests, smoke, kernel statistics of the bench command, the driver's own command line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c_final; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -k "insert or classif or native or smoke or abi" --tb=short 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_100k_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_100k.csv; rm -rf $OUT/stats
head -5 $OUT/kernel_stats_100k.csv | cut -c1-120
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_100k_steps20_warmup5.json 2> $OUT/bench_100k_steps20.err; head -c 300 $OUT/bench_100k_steps20_warmup5.json; echo; echo "wall seconds of the whole command: $SECONDS"
