# round-6 measurement suite (run on the GPU box; outputs under gpurun_out/r06, copied into profiles/r06_* by hand).
#   PART=a  the metric's workload: bench line, kernel statistics of the same command, PMC traffic of its first iteration
#   PART=b  BASELINE configs[3] (K = 4 classification through thx_refine_iterate): line, kernel statistics, PMC traffic of its
#           local-search launches; configs[1] (10 k particles)
#   PART=c  BASELINE configs[4] (512^3, 20 000 particles on one GPU): line, kernel statistics, PMC traffic
#   PART=d  the driver's own command line at this HEAD (+ its wall time) and the GPU suite's tail
#   PART=e  round 6: an iteration BELOW Nyquist (configs[1] at r = rU = 48: resized reconstruction grid) -- line + kernel statistics;
#           the drop-in path (bench.py --staged: C++ caller loop with / without the reference's per-GPU lock) -- line + kernel statistics
#   PART=f  the drop-in path alone, + the kernel-by-kernel chain of a locked image-phase (tools/probes/chain_trace.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
stats() {   # stats <name> <bench args...>
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$name -- python bench.py "$@" --no-cpu-baseline --other-configs off > $OUT/bench_${name}_under_rocprof.json 2> $OUT/st_$name.err
    cp $(find $OUT/st_$name -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$name.csv; rm -rf $OUT/st_$name
    head -6 $OUT/kernel_stats_$name.csv | cut -c1-150
}
pmc() {     # pmc <name> <images per E-step launch x 2> <box> <bench args...>
    local name=$1 n2=$2 box=$3; shift 3
    PMC_CMD="python bench.py $* --steps 1 --warmup 0 --no-cpu-baseline --other-configs off" THX_PROBE_PARTICLES=$n2 THX_PROBE_BOX=$box bash tools/pmc_traffic.sh > $OUT/pmc_$name.log 2>&1
    cp gpurun_out/pmc_traffic/summary.json $OUT/pmc_traffic_summary_$name.json
    cp gpurun_out/pmc_traffic/pmc_traffic.json $OUT/pmc_traffic_$name.json
    cp gpurun_out/pmc_traffic/lds_atomic_bench.txt $OUT/lds_atomic_bench.txt
    rm -rf gpurun_out/pmc_traffic
    cat $OUT/pmc_traffic_$name.json; echo
}
case "${PART:-a}" in
a)
    python bench.py --other-configs off > $OUT/bench_100k.json 2> $OUT/bench_100k.err; tail -c 1200 $OUT/bench_100k.json; echo
    stats 100k --steps 2 --warmup 1
    pmc 100k 20000 256 ;;
b)
    timeout 900 python bench.py --classification --other-configs off > $OUT/bench_classification_k4.json 2> $OUT/bench_classification_k4.err; tail -c 900 $OUT/bench_classification_k4.json; echo
    stats classification_k4 --classification --steps 2 --warmup 1
    pmc classification_k4 6250 256 --classification
    python bench.py --particles 10000 --steps 2 --warmup 1 --other-configs off > $OUT/bench_10k.json 2> $OUT/bench_10k.err; head -c 400 $OUT/bench_10k.json; echo ;;
c)
    timeout 1500 python bench.py --box 512 --particles 20000 --steps 2 --warmup 1 --other-configs off > $OUT/bench_box512_20k.json 2> $OUT/bench_box512_20k.err; tail -c 900 $OUT/bench_box512_20k.json; echo
    stats box512 --box 512 --particles 20000 --steps 1 --warmup 1
    pmc box512 5000 512 --box 512 --particles 20000 ;;
d)
    SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_100k_steps20_warmup5.json 2> $OUT/bench_100k_steps20.err
    head -c 400 $OUT/bench_100k_steps20_warmup5.json; echo; echo "wall seconds of the whole command: $SECONDS" | tee $OUT/driver_command_wall.txt
    timeout 1500 python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" > $OUT/pytest_gpu_full.txt; tail -4 $OUT/pytest_gpu_full.txt > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
    python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 ;;
e)
    python bench.py --particles 10000 --steps 2 --warmup 1 --cutoff 48 48 --other-configs off > $OUT/bench_10k_cutoff48.json 2> $OUT/bench_10k_cutoff48.err; head -c 600 $OUT/bench_10k_cutoff48.json; echo
    stats 10k_cutoff48 --particles 10000 --steps 2 --warmup 1 --cutoff 48 48
    python bench.py --staged > $OUT/bench_staged.json 2> $OUT/bench_staged.err; head -c 500 $OUT/bench_staged.json; echo
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_staged -- python bench.py --staged --staged-images 512 > $OUT/bench_staged_under_rocprof.json 2> $OUT/st_staged.err
    cp $(find $OUT/st_staged -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_staged.csv; rm -rf $OUT/st_staged
    head -8 $OUT/kernel_stats_staged.csv | cut -c1-150 ;;
f)  # the drop-in path alone (after the zero-copy outputs + completion word): line, kernel statistics, the chain of a locked image-phase
    python bench.py --staged > $OUT/bench_staged.json 2> $OUT/bench_staged.err; head -c 500 $OUT/bench_staged.json; echo
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_staged -- python bench.py --staged --staged-images 512 > $OUT/bench_staged_under_rocprof.json 2> $OUT/st_staged.err
    cp $(find $OUT/st_staged -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_staged.csv
    python tools/probes/chain_trace.py $(find $OUT/st_staged -name "*kernel_trace.csv" | head -1) 150 1500 | tee $OUT/dropin_chain_trace.txt
    rm -rf $OUT/st_staged ;;
esac
