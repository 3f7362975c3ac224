set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_w -o w --output-format csv -- python $R/tools/wav_probe.py 64 > $R/gpurun_out/r6c/wav_probe.log 2>&1
f=$(find /tmp/prof_w -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|spy" "$f" | cut -c1-200 > $R/gpurun_out/r6c/wav_kernel_stats.csv
cat $R/gpurun_out/r6c/wav_probe.log | tail -4
cat $R/gpurun_out/r6c/wav_kernel_stats.csv
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cwt" 2>&1 | tail -5 | tee gpurun_out/r6c/cwt_tests.log
