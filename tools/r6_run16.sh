cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6p
timeout 600 python -m pytest tests/test_gpu_k4h.py tests/test_gpu_production.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r6p/bench.json 2>gpurun_out/r6p/bench.err; tail -2 gpurun_out/r6p/bench.err
python -c "
import json; l=json.load(open('gpurun_out/r6p/bench.json')); print(l['value'], l['secondary']['front_end'], l['value_with_host_copy'])"
