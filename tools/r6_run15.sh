set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6n
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r6n/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r6n/gpu_tests.log
for off in 1300000 1700000; do
SPY_FUZZ_SCALE=40 SPY_FUZZ_OFFSET=$off timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q --tb=line -n 4 2>&1 | grep -E "Error|FAILED|passed|failed" | cut -c1-500 | tee gpurun_out/r6n/fuzz_offset$off.log
done
python tools/wav_probe.py 200 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6n/wav_probe.txt
bash tools/final_bench.sh > gpurun_out/final_bench.log 2>&1; tail -12 gpurun_out/final_bench.log
