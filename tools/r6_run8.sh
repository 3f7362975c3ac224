cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
SPY_FUZZ_OFFSET=500000 python tools/granger_diag.py 94 2>&1 | grep -v Warn | tee gpurun_out/r6h/granger94.log
