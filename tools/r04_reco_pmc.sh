# SQ counters of the gridding loop's hand-written FFT passes at P = 512 and P = 1024 (kernel-trace only; two passes of counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/recopmc; rm -rf $OUT; mkdir -p $OUT
for N in 256 512; do
CMD="python tools/probes/reco_time.py $N"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a$N -- $CMD > /dev/null 2> $OUT/a$N.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/b$N -- $CMD > /dev/null 2> $OUT/b$N.err
done
python - <<'PY' | tee gpurun_out/r04_reco_pmc.txt
import csv, glob, collections
for N in (256, 512):
  for tag in ('a', 'b'):
    fs = glob.glob('gpurun_out/recopmc/%s%d/*/*counter_collection.csv' % (tag, N))
    if not fs:
        print('no counters in pass', tag, N); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        kn = r['Kernel_Name']
        k = None
        for key in ('k_fft_z_update', 'k_fft_x_conv', 'k_fft_strided'):
            if key in kn: k = key + ('<first>' if key == 'k_fft_z_update' and ', true, ' in kn.split('(')[0][-30:] else '')
        if k is None or 'first' in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, d in sorted(acc.items()):
        print('P = %d  %s (pass %s)' % (2 * N, k, tag))
        for c, v in sorted(d.items()): print('    %-24s %16.0f per dispatch over %d dispatches' % (c, v / n[(k, c)], n[(k, c)]))
PY
