set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r6g/gpu_tests.log
SPY_FUZZ_SCALE=40 SPY_FUZZ_OFFSET=500000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q --tb=line -n 4 2>&1 | grep -E "Error|FAILED|passed|failed" | cut -c1-500 | tee gpurun_out/r6g/fuzz_offset500000.log
SPY_FUZZ_SCALE=40 SPY_FUZZ_OFFSET=900000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q --tb=line -n 4 2>&1 | grep -E "Error|FAILED|passed|failed" | cut -c1-500 | tee gpurun_out/r6g/fuzz_offset900000.log
