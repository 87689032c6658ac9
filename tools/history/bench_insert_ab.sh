# A/B of the insertion stage inside the bench's own iteration: default library against THX_LIB=...alt.so (tools/build_alt.sh)
cd $GRAFT_REPO_ROOT
for lib in "" "$GRAFT_REPO_ROOT/thunder_amd/lib/libthunder_amd_alt.so"; do
  THX_LIB=$lib python bench.py --particles ${1:-20000} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=%s' % ('$lib'[-12:] or 'default'), round(d['value'],1), d['stages_ms_per_step'])"
done
