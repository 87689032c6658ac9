# native classification driver: parity tests, then the K = 4 iteration native vs Python-sequenced
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_next_gpu.py -q -m gpu -k "classification" --tb=short 2>&1 | tail -30
timeout 600 python bench.py --classification --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03c/classify_native.json 2> gpurun_out/r03c/classify_native.err; tail -3 gpurun_out/r03c/classify_native.err; head -c 1500 gpurun_out/r03c/classify_native.json; echo
timeout 600 python bench.py --classification --steps 2 --warmup 1 --no-cpu-baseline --python-sequencing > gpurun_out/r03c/classify_python.json 2> gpurun_out/r03c/classify_python.err; tail -3 gpurun_out/r03c/classify_python.err; head -c 300 gpurun_out/r03c/classify_python.json; echo
