# window-geometry sweep of k_insert_win: times tools/insert_probe.py with each variant library under thunder_amd/lib/variants
# (built in the container: thx_mstep.hip with -DTHX_KWD / -DTHX_KWZ / -DTHX_KIPIX / -DTHX_KWINTHREADS, linked with the other objects)
for so in thunder_amd/lib/variants/lib_*.so; do
  echo "$(basename $so): $(THX_LIB=$PWD/$so DBGS=0 timeout 300 python tools/insert_probe.py 2048 2>&1 | grep 'debug=0')"
done
