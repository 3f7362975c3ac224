set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 4 2>&1 | tail -4 > gpurun_out/r4_gpu_tests_final.log
cat gpurun_out/r4_gpu_tests_final.log
SPY_FUZZ_SCALE=20 python -m pytest tests/test_gpu_fuzz.py -q --tb=line -n 4 2>&1 | tail -3 > gpurun_out/r4_fuzz_scale20.log
cat gpurun_out/r4_fuzz_scale20.log
SPY_FUZZ_SCALE=10 SPY_FUZZ_OFFSET=100000 python -m pytest tests/test_gpu_fuzz.py -q --tb=line -n 4 2>&1 | tail -3 > gpurun_out/r4_fuzz_offset100000.log
cat gpurun_out/r4_fuzz_offset100000.log
