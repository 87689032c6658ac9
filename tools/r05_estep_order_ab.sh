#!/bin/bash
# Round 5, E-step A/B on one box: lane <-> rotation of k_expect_local following a ranking of every image's cloud
# (THX_EXPECT_ORDER = 1 in-plane angle, 2 / 3 the two tilt components; thx_estep.hip:k_cloud_order) against the storage order, on
# 20 000 particles of the refinement workload and on one GPU's share of the classification workload (wide clouds after a scan).
# Prints one line per variant; full JSON under gpurun_out/r05_estep_order_ab/.
set -u
out=gpurun_out/r05_estep_order_ab; mkdir -p $out
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --other-configs off > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; return; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("rooflines", {}).get("local_phases", d["roofline"])
st = d["stages_ms_per_step"]
print("%-28s value %9.1f %s  E-step launch %.2f ms (%.0f img) frac %.3f  stage expectation %.1f ms" % (sys.argv[2], d["value"], d["unit"], r["avg_launch_ms"], r["images_per_launch"], r["frac"], st["expectation"]))
PY
}
R="--particles 20000 --steps 2 --warmup 1"
run refine_default THX_X=0 -- $R
run refine_order1 THX_EXPECT_ORDER=1 -- $R
run refine_order2 THX_EXPECT_ORDER=2 -- $R
run refine_order3 THX_EXPECT_ORDER=3 -- $R
run refine_order1_wg3 THX_EXPECT_ORDER=1 THX_EXPECT_WG_PER_CU=3 -- $R
run refine_order1_wg4 THX_EXPECT_ORDER=1 THX_EXPECT_WG_PER_CU=4 -- $R
run refine_default_again THX_X=0 -- $R
C="--classification --steps 2 --warmup 1"
run classify_default THX_X=0 -- $C
run classify_order1 THX_EXPECT_ORDER=1 -- $C
run classify_order2 THX_EXPECT_ORDER=2 -- $C
run classify_order1_wg3 THX_EXPECT_ORDER=1 THX_EXPECT_WG_PER_CU=3 -- $C
