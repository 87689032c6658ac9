# round 4: the gridding loop with W / T tiled by z column -- tests, per-round timings at 1024^3 / 512^3 / 256^3, and the driver's
# other-configs sequence (a 10 k headline first, so that the 512^3 job meets the 32 GiB insertion scratch of an earlier configuration)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -k "hand_fft or reconstruct or reco or insert or classification or config0" 2>&1 | tail -6
for n in 512 256 128; do python tools/probes/reco_time.py $n 2>&1 | tail -3; done | tee gpurun_out/r04_reco_time.txt
timeout 1200 python bench.py --particles 10000 --steps 1 --warmup 0 --no-cpu-baseline --other-configs on > gpurun_out/r04_others_check.json 2> gpurun_out/r04_others_check.err
tail -c 600 gpurun_out/r04_others_check.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_others_check.json").read().strip().splitlines()[-1])
for k,v in d.get("other_configs",{}).items(): print(k, v.get("value"), v.get("error"), v.get("stages_ms_per_step"))
PY
