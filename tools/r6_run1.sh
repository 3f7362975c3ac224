set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
python bench.py --steps 20 --warmup 5 > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r6a/bench.json)"
cp gpurun_out/bench_detail.json gpurun_out/r6a/ 2>/dev/null
python bench.py --gpus 2 --steps 2 > gpurun_out/r6a/bench_gpus2.out 2>&1; echo "gpus2 rc=$?"
timeout 1500 python -m pytest tests/test_gpu_depth.py -m gpu -x -q -s 2>&1 | grep -E "depth|passed|failed|Error|assert" | tee gpurun_out/r6a/depth.log
cp gpurun_out/depth_table.json gpurun_out/r6a/ 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_depth.py 2>&1 | tail -5 | tee gpurun_out/r6a/gpu_tests.log
