#!/bin/bash
# K6 at c5 (256 channels, 2049 frequencies): kernel-trace stats (product path, config_probe) and counters of the final
# kernels (torch-free harness: PMC collection crashes inside torch's kernels on this image); summaries only
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/prof_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_w -o w --output-format csv -- python $R/tools/config_probe.py granger256 > /tmp/w.log 2>&1
grep "^granger" /tmp/w.log
cp $(find /tmp/prof_w -name "*kernel_stats.csv" | head -1) $R/gpurun_out/wilson_kernel_stats.csv
cd $R
mkdir -p gpurun_out/pmc
hipcc -O2 tools/pmc_harness.cpp -Iinclude -Lsyncopy_amd -lspyhip -Wl,-rpath,$PWD/syncopy_amd -o gpurun_out/pmc/harness || exit 1
gpurun_out/pmc/harness 300 1 7 0 || exit 1
bash tools/pmc_probe.sh wilson gpurun_out/pmc/harness 300 1 7 0 > /tmp/pmcw.log 2>&1
grep "^group" /tmp/pmcw.log
cp gpurun_out/pmc_wilson/summary.txt gpurun_out/wilson_pmc.txt
rm -rf gpurun_out/pmc_wilson gpurun_out/pmc
grep -A22 "spywil::zinv64\|spywil::plus4_kernel<12>" gpurun_out/wilson_pmc.txt | head -60
