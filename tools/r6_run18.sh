cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s
for off in ${FAMILIES:-0 2100000 2500000 2900000 3300000}; do
SPY_FUZZ_SCALE=40 SPY_FUZZ_OFFSET=$off timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q --tb=line -n 4 2>&1 | grep -E "Error|FAILED|passed|failed" | cut -c1-600 | tee gpurun_out/r6s/fuzz_offset$off.log
done
