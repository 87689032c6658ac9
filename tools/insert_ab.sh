# kernel split of the insertion probe under rocprofv3 (run on the GPU box): bash tools/insert_ab.sh [particles] [box]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -- python tools/insert_ab.py ${1:-8192} ${2:-256} > gpurun_out/ab.txt 2>&1
cp $(find gpurun_out/stats -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_ab.csv; rm -rf gpurun_out/stats
grep -v "^[EWI]2026" gpurun_out/ab.txt | tail -4
grep "k_bin\|k_acc\|radix\|scan\|k_seg\|k_insert_win" gpurun_out/kernel_stats_ab.csv | cut -c1-60,200-330 | sed 's/rocprim::ROCPRIM_400200_NS:://g'
