# brick geometry of the sorted insertion (tools/build_alt.sh "-DTHX_BRICK_LZ=2 -DTHX_ACC_WGS=4" z4, ...): insertion tests on one variant,
# then the insertion stage of the bench's iteration for every build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
THX_LIB=$GRAFT_REPO_ROOT/thunder_amd/lib/libthunder_amd_${TESTLIB:-z4}.so timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "insert" --tb=short 2>&1 | tail -3
for name in ${VARIANTS:-"" z4 y8 y8z4 ""}; do
  lib=""; [ -n "$name" ] && [ "$name" != "_" ] && lib=$GRAFT_REPO_ROOT/thunder_amd/lib/libthunder_amd_$name.so
  THX_LIB=$lib python bench.py --particles ${1:-20000} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=%s' % ('$name' or 'default'), round(d['value'],1), 'insertion', d['stages_ms_per_step']['insertion'], 'expect', d['stages_ms_per_step']['expectation'])"
done
