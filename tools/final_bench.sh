#!/bin/bash
# Round-end measurement on the GPU box: the bench line, the kernel-trace stats of the headline-only run, PMC counters.
# Traces go to /tmp (gpurun_out/ is limited to 64 MiB); only summaries are copied back.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 600 gpurun_out/bench_full.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o p --output-format csv -- python $R/bench.py --no-secondary --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> /tmp/rocprof_bench.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/bench_kernel_stats.csv
head -8 "$f" | cut -c1-160
cd $R
bash tools/pmc_run.sh 1000 0 > /tmp/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt gpurun_out/pmc_summary.txt
rm -rf gpurun_out/pmc
tail -45 gpurun_out/pmc_summary.txt
