# A/B of k_bin with its samples kept in registers between count and scatter (-DTHX_BIN_KEEP=1, tools/build_alt.sh): insertion tests on the
# alternative build, then the insertion stage inside the bench's iteration for both builds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
THX_LIB=$GRAFT_REPO_ROOT/thunder_amd/lib/libthunder_amd_alt.so timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "insert" --tb=short 2>&1 | tail -5
bash tools/bench_insert_ab.sh 20000
bash tools/bench_insert_ab.sh 20000
