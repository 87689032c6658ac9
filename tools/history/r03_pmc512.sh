# E-step / insertion HBM traffic at the 512^3 box (P = 1024: the 4 MiB L2s and the 256 MiB Infinity Cache stop helping)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
PMC_CMD="python bench.py --box 512 --particles 10000 --batch 2500 --steps 1 --warmup 0 --no-cpu-baseline" THX_PROBE_PARTICLES=5000 bash tools/pmc_traffic.sh > gpurun_out/r03/pmc512.log 2>&1
cp gpurun_out/pmc_traffic/summary.json gpurun_out/r03/pmc_traffic_summary_box512.json
rm -rf gpurun_out/pmc_traffic
python - <<'PY'
import json
s = json.load(open("gpurun_out/r03/pmc_traffic_summary_box512.json"))
for k in ("k_expect_local", "k_insert_win"):
    e = s.get(k, {})
    print(k, {a: e.get(a) for a in ("hbm_bytes_per_launch", "TCC_EA0_RDREQ_sum_per_launch", "TCC_HIT_sum_per_launch", "TCC_MISS_sum_per_launch", "images_per_launch")})
PY
