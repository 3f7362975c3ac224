cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests/test_gpu_k4h.py -m gpu -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_production.py tests/test_gpu_depth.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r6m/bench.json 2> gpurun_out/r6m/bench.err; tail -2 gpurun_out/r6m/bench.err
python -c "
import json; l=json.load(open('gpurun_out/r6m/bench.json')); print(l['value'], l['secondary']['front_end'], l['value_with_host_copy'])"
