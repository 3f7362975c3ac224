set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
SPY_FUZZ_SCALE=40 SPY_FUZZ_OFFSET=500000 timeout 600 python -m pytest "tests/test_gpu_fuzz.py::test_granger_random_networks[94]" "tests/test_gpu_fuzz.py::test_connectivity_random_options[23]" "tests/test_gpu_fuzz.py::test_connectivity_random_options[284]" -q 2>&1 | grep -v Warn | tail -40 | cut -c1-600 | tee gpurun_out/r6f/cases_500000.log
SPY_FUZZ_SCALE=40 SPY_FUZZ_OFFSET=900000 timeout 600 python -m pytest "tests/test_gpu_fuzz.py::test_mtmfft_random_options[1624]" "tests/test_gpu_fuzz.py::test_connectivity_random_options[668]" -q 2>&1 | grep -v Warn | tail -20 | cut -c1-600 | tee gpurun_out/r6f/cases_900000.log
