cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cwt" 2>&1 | tail -3
python tools/wav_probe.py 200 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests -m gpu -x -q -k "wavelet or superlet or cwt or slt or tf_ or timefrequency" 2>&1 | tail -3
