#!/bin/bash
# builds and runs the fft_probe variants; results in gpurun_out/fft_probe.txt
set -u
mkdir -p gpurun_out/fftp
out=gpurun_out/fft_probe.txt; : > $out
i=0
run() {  # label, flags...
  i=$((i+1)); local lab="$1"; shift
  if hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result "$@" tools/fft_probe.hip -o gpurun_out/fftp/p$i 2> gpurun_out/fftp/p$i.err; then
    echo -n "[$lab] " >> $out; timeout 120 gpurun_out/fftp/p$i 125 >> $out 2>&1
  else echo "[$lab] BUILD FAILED" >> $out; tail -3 gpurun_out/fftp/p$i.err >> $out; fi
}
for mode in "-DPOUTK=2 -DPMEAN=0" "-DPOUTK=0 -DPMEAN=1" "-DPOUTK=0 -DPMEAN=0"; do
  run "base $mode" $mode
  run "notaper $mode" $mode -DSPYFFT_ABL=1
  run "notw $mode" $mode -DSPYFFT_ABL=2
  run "nostore $mode" $mode -DSPYFFT_ABL=4
  run "none $mode" $mode -DSPYFFT_ABL=7
done
rm -rf gpurun_out/fftp
cat $out
