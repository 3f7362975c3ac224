cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6l
SPY_FUZZ_OFFSET=1700000 python tools/fuzz_diag.py toi 187 2>&1 | grep -v "Warn\|amdgpu" | cut -c1-900 | tee gpurun_out/r6l/diag_toi187.log
SPY_FUZZ_OFFSET=1300000 python tools/fuzz_diag.py conn 790 2>&1 | grep -v "Warn\|amdgpu" | cut -c1-600 | tee gpurun_out/r6l/diag_conn790.log
