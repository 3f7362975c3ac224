set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cwt" 2>&1 | tail -5 | tee gpurun_out/r6b/cwt_tests.log
python tools/wav_probe.py 200 2>&1 | tee gpurun_out/r6b/wav_probe.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "wavelet or superlet or cwt or slt" 2>&1 | tail -5 | tee gpurun_out/r6b/wav_tests.log
