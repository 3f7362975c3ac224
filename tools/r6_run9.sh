cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
python tools/k1_layout_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6i/k1_layout_probe.txt
timeout 900 python -m pytest tests/test_gpu_k4h.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
