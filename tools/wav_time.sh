# kernel durations of the c4 wavelet workload (torch-free harness):  bash tools/wav_time.sh
export TMPDIR=/tmp
hipcc -O2 tools/pmc_harness2.cpp -Iinclude -Lsyncopy_amd -lspyhip -Wl,-rpath,$PWD/syncopy_amd -o /tmp/h2 || exit 1
rm -rf /tmp/prof_s
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s --output-format csv -- /tmp/h2 wav > /tmp/h2.log 2>&1 )
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1)
grep -E "spy" "$f" | awk -F'",' '{print substr($1,1,48), $2}'
