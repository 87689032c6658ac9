#!/bin/bash
# Round 4, E-step A/B on one box: the near-slab / tail form (THX_EXPECT_SPLIT = margin in voxels) and a per-phase occupancy cap
# (THX_EXPECT_WG_LATER) against the default kernel, on 20 000 particles of the refinement workload and on one GPU's share of the
# classification workload (wide clouds after a scan).  Prints one line per variant; full JSON under gpurun_out/r04_estep_ab/.
set -u
out=gpurun_out/r04_estep_ab; mkdir -p $out
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
run refine_split4 THX_EXPECT_SPLIT=4 -- $R
run refine_split8 THX_EXPECT_SPLIT=8 -- $R
run refine_split1000 THX_EXPECT_SPLIT=1000 -- $R
run refine_wg_later3 THX_EXPECT_WG_LATER=3 -- $R
run refine_split4_wg3 THX_EXPECT_SPLIT=4 THX_EXPECT_WG_PER_CU=3 -- $R
run refine_default_again THX_X=0 -- $R
C="--classification --steps 2 --warmup 1"
run classify_default THX_X=0 -- $C
run classify_split8 THX_EXPECT_SPLIT=8 -- $C
run classify_split16 THX_EXPECT_SPLIT=16 -- $C
run classify_split1000 THX_EXPECT_SPLIT=1000 -- $C
