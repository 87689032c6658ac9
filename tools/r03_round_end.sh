# round-end check on the GPU box: full -m gpu suite, the driver's command line, the global-scan line with its CPU baseline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -15
python bench.py --classification --steps 2 --warmup 1 > gpurun_out/r03/bench_global_scan.json 2> gpurun_out/r03/bench_global_scan.err; tail -c 900 gpurun_out/r03/bench_global_scan.json; tail -2 gpurun_out/r03/bench_global_scan.err
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_100k_steps20_warmup5.json 2> gpurun_out/r03/bench_100k_steps20.err; head -c 300 gpurun_out/r03/bench_100k_steps20_warmup5.json; grep -i "elapsed" gpurun_out/r03/bench_100k_steps20.err
