cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 512 256 128 64; do python tools/probes/reco_time.py $n 2>&1 | grep "^N ="; done | tee gpurun_out/r04_reco_time.txt
timeout 600 python -m pytest tests -q -m gpu -k "hand_fft or reconstruct" 2>&1 | tail -3
for b in 10240 25600 51200; do python bench.py --no-cpu-baseline --other-configs off --batch $b > gpurun_out/r04_batch_$b.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r04_batch_$b.json").read().strip().splitlines()[-1])
print("batch $b", d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["images_per_launch"], d["roofline"]["frac"], d["stages_ms_per_step"]["expectation"], d["stages_ms_per_step"]["insertion"])
PY
done | tee gpurun_out/r04_batch_ab.txt
