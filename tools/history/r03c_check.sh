# HEAD check on the GPU box: full -m gpu suite with durations, smoke(), default bench line (normCorrection on)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
SECONDS=0
python -m pytest tests -q -m gpu --tb=short --durations=12 2>&1 | tail -40 > gpurun_out/r03c/pytest_gpu.txt; tail -25 gpurun_out/r03c/pytest_gpu.txt
echo "pytest wall: $SECONDS"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
SECONDS=0
python bench.py > gpurun_out/r03c/bench_100k.json 2> gpurun_out/r03c/bench_100k.err; head -c 600 gpurun_out/r03c/bench_100k.json; echo; tail -3 gpurun_out/r03c/bench_100k.err
echo "bench wall: $SECONDS"
