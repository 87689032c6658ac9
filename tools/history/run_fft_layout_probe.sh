cd /tmp && export TMPDIR=/tmp
timeout 200 $GRAFT_REPO_ROOT/tools/fft_layout_probe 512 2>&1 | tail -8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fl -- $GRAFT_REPO_ROOT/tools/fft_layout_probe 512 > /dev/null 2>&1
f=$(find /tmp/prof_fl -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | head -30
