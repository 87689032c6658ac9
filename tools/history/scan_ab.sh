mkdir -p gpurun_out/scan
timeout 600 python -m pytest tests -m gpu -x -q -k "global or classification or iface" 2>&1 | tail -3
for t in t22 t42 t24 t44; do
  THX_SCAN=$t timeout 300 python bench.py --classification --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', j['roofline']['avg_launch_ms'], j['roofline']['achieved'], j['config']['classes_recovered'])"
done
