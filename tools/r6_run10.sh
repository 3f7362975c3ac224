set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "cwt" 2>&1 | tail -3
python tools/wav_probe.py 200 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6j/wav_probe.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "wavelet or superlet or cwt or slt or tf_ or timefrequency" 2>&1 | tail -3
hipcc -O2 tools/pmc_harness2.cpp -Iinclude -Lsyncopy_amd -lspyhip -Wl,-rpath,$PWD/syncopy_amd -o /tmp/pmc_harness2 || exit 1
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s --output-format csv -- /tmp/pmc_harness2 wav > /tmp/h2.log 2>&1 )
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|spy" "$f" | cut -c1-160 | tee gpurun_out/r6j/wav_kernel_stats.txt
bash tools/pmc_probe.sh wav /tmp/pmc_harness2 wav > /tmp/pmc_wav.log 2>&1
grep -E "^[a-z]|^void|FETCH_SIZE|WRITE_SIZE" gpurun_out/pmc_wav/summary.txt | cut -c1-120 | tee gpurun_out/r6j/wav_pmc.txt
rm -rf gpurun_out/pmc_wav
