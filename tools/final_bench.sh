#!/bin/bash
# ONE script behind every figure of the bench line (VERDICT r3 "next" 3).  On the GPU box:  bash tools/final_bench.sh
# Everything lands in gpurun_out/final/ (summaries only: traces stay in /tmp); the files are then copied to profiles/r6_*
# (bench.py: PROFILE_ROUND).
#   bench.json, bench_detail.json   python bench.py --steps 20 --warmup 5 (the line the driver also produces, and the detail file behind it)
#   bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the headline-only run (K0 / K1 / K4 / K5 rows)
#   bench_under_rocprof.json        that run's own line (HIP-event averages inside the profiled process)
#   secondary_kernel_stats.txt      rocprofv3 --kernel-trace --stats of each `secondary` workload through the torch-free
#                                   harness (tools/pmc_harness2.cpp: same plans, shapes and launch arguments as bench.py)
#   pmc_headline.txt                counters of K1 (fourier) / K4 at the headline launch shape (tools/pmc_run.sh)
#   pmc_secondary.txt               counters per secondary workload: FETCH_SIZE / WRITE_SIZE (bench.py's `traffic`),
#                                   SQ_INSTS_VALU, SQ_WAIT_INST_LDS, SQ_LDS_BANK_CONFLICT ... (tools/pmc_probe.sh groups)
#   wilson_kernel_stats.csv, wilson_pmc.txt     K6 at c5 (tools/wilson_final.sh)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
cd $R
bash tools/pmc_run.sh 1000 0 > /tmp/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt $O/pmc_headline.txt
rm -rf gpurun_out/pmc
cp $O/pmc_headline.txt profiles/r6_pmc_headline.txt      # (bench.py reads its `traffic` from profiles/: counters first)
hipcc -O2 tools/pmc_harness2.cpp -Iinclude -Lsyncopy_amd -lspyhip -Wl,-rpath,$PWD/syncopy_amd -o /tmp/pmc_harness2 || exit 1
MODES=${MODES:-"c2 c2f64 n2000 n2000f64 n3000 n3000f64 n5000 n5000f64 n10000 n10000f64 n12000 n12000f64 n16384 n16384f64 conv wav"}
: > $O/secondary_kernel_stats.txt
: > $O/pmc_secondary.txt
for m in $MODES; do
  rm -rf /tmp/prof_s
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s --output-format csv -- /tmp/pmc_harness2 $m > /tmp/h2.log 2>&1 )
  echo "## $m : $(grep -E 'kernel|wav' /tmp/h2.log | tail -1)" >> $O/secondary_kernel_stats.txt
  f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1)
  grep -E "^\"Name\"|spy" "$f" >> $O/secondary_kernel_stats.txt
  bash tools/pmc_probe.sh $m /tmp/pmc_harness2 $m > /tmp/pmc_$m.log 2>&1
  echo "## $m : $(grep -E 'kernel|wav' /tmp/h2.log | tail -1)" >> $O/pmc_secondary.txt
  grep -v "^# command" gpurun_out/pmc_$m/summary.txt >> $O/pmc_secondary.txt
  rm -rf gpurun_out/pmc_$m
done
grep -E "^##|^\"spy" $O/secondary_kernel_stats.txt | cut -c1-150
cp $O/pmc_secondary.txt profiles/r6_pmc_secondary.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cp gpurun_out/bench_detail.json $O/bench_detail.json
tail -c 400 $O/bench.err
echo "bench line: $(wc -c < $O/bench.json) bytes"
cd /tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o p --output-format csv -- python $R/bench.py --no-secondary --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/rocprof_bench.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|spy" "$f" > $O/bench_kernel_stats.csv
head -6 $O/bench_kernel_stats.csv | cut -c1-170
cd $R
bash tools/wilson_final.sh > /tmp/wilson_final.log 2>&1
mv gpurun_out/wilson_kernel_stats.csv gpurun_out/wilson_pmc.txt $O/ 2>/dev/null
ls -la $O
