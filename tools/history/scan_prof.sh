# per-kernel durations of the global scanning stage (bench.py --classification) under rocprofv3, for one tile shape
t=${1:-t22}
rm -rf /root/repo/gpurun_out/scanprof_$t; cd /tmp && export TMPDIR=/tmp
THX_SCAN=$t rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/scanprof_$t -- python /root/repo/bench.py --classification --steps 2 --warmup 1 > /root/repo/gpurun_out/scanprof_$t.json 2>/dev/null
cd /root/repo
f=$(ls gpurun_out/scanprof_$t/*/*kernel_stats.csv | head -1)
head -8 $f | cut -c1-60,150-260
