# round-end check on the GPU box: full -m gpu suite, smoke(), BASELINE configs[1], the driver's own command line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -q -m gpu --tb=line 2>&1 | grep -a "passed\|failed\|error" | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --particles 10000 --steps 2 --warmup 1 > gpurun_out/r03/bench_10k.json 2> gpurun_out/r03/bench_10k.err; head -c 400 gpurun_out/r03/bench_10k.json; echo
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_100k_steps20_warmup5.json 2> gpurun_out/r03/bench_100k_steps20.err; head -c 300 gpurun_out/r03/bench_100k_steps20_warmup5.json; echo; echo "wall seconds of the whole command: $SECONDS"
