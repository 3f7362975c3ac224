set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6t
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r6t/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r6t/gpu_tests.log
bash tools/final_bench.sh > gpurun_out/final_bench.log 2>&1; tail -12 gpurun_out/final_bench.log
