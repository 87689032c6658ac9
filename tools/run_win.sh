export THX_INSERT_KERNEL=win
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "insert" 2>&1 | tail -5
for v in tiles win; do
  export THX_INSERT_KERNEL=$v
  echo "== insert kernel $v"
  timeout 300 python tools/pf_probe2.py 256 2048 2>&1 | grep "insertion"
done
