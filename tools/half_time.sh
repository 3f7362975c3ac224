export TMPDIR=/tmp
hipcc -O2 tools/pmc_harness2.cpp -Iinclude -Lsyncopy_amd -lspyhip -Wl,-rpath,$PWD/syncopy_amd -o /tmp/h2 || exit 1
for m in ${MODES:-n12000f64 n16384f64}; do
 for t in 0; do
  rm -rf /tmp/prof_s
  if [ $t = 1 ]; then export SPYHIP_HALF_TRY=1; else unset SPYHIP_HALF_TRY; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s --output-format csv -- /tmp/h2 $m > /tmp/h2.log 2>&1 )
  echo "## $m try=$t: $(grep -E 'kernel' /tmp/h2.log | tail -1)"
  f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1)
  grep -E "spy" "$f" | grep -v seq_mean | sed -e "s/(spyfft::MtmArgs)//" | cut -c20-200
 done
done
