#!/bin/bash
# Round 4, insertion A/B on one box (20 000 particles of the refinement workload): record scratch of the brick-sorted insertion
# (THX_INSERT_SCRATCH_MB: larger chunks flush each brick fewer times) on the pipelined chunk loop.  One line per variant; full JSON
# under gpurun_out/r04_insert_ab/.
set -u
out=gpurun_out/r04_insert_ab; mkdir -p $out
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --other-configs off > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; return; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d["stages_ms_per_step"]
k = next((v for n_, v in d.get("kernels", {}).items() if n_.startswith("insertion")), {})
print("%-24s value %9.1f %s  stage insertion %.1f ms  expectation %.1f ms  %s" % (sys.argv[2], d["value"], d["unit"], st["insertion"], st["expectation"],
      " ".join("%s=%.4g" % (a, b) for a, b in k.items() if isinstance(b, (int, float)))))
PY
}
R="--particles 20000 --steps 2 --warmup 1"
run default THX_X=0 -- $R
run scratch_4g THX_INSERT_SCRATCH_MB=4096 -- $R
run scratch_16g THX_INSERT_SCRATCH_MB=16384 -- $R
run scratch_32g THX_INSERT_SCRATCH_MB=32768 -- $R
run scratch_64g THX_INSERT_SCRATCH_MB=65536 -- $R
run default_again THX_X=0 -- $R
