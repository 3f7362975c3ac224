#!/bin/bash
set -u
mkdir -p gpurun_out/k1s
out=gpurun_out/k1_session.txt; : > $out
FL="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value"
b() { hipcc $FL "$@" 2>gpurun_out/k1s/err.txt || { echo "BUILD FAILED: $*" >> $out; tail -3 gpurun_out/k1s/err.txt >> $out; return 1; }; }
run() { b "$@" tools/dec_probe.hip -o gpurun_out/k1s/q && timeout 120 gpurun_out/k1s/q 300 >> $out 2>&1; }
W4="-DSPYFFT_KATTR=__attribute__((amdgpu_waves_per_eu(4,4)))"
W3="-DSPYFFT_KATTR=__attribute__((amdgpu_waves_per_eu(3,3)))"
for mode in "-DPOUTK=0 -DPMEAN=1" "-DPOUTK=2 -DPMEAN=0"; do
  run $mode -DDV=10 -DDR1=10 -DDR2=10 -DDR3=2 -DDG=1
  run $mode -DDV=10 -DDR1=10 -DDR2=10 -DDR3=2 -DDG=1 "$W4"
  run $mode -DDV=20 -DDR1=10 -DDR2=10 -DDR3=1 -DDG=2
  run $mode -DDV=20 -DDR1=10 -DDR2=10 -DDR3=1 -DDG=1
  run $mode -DDV=10 -DDR1=10 -DDR2=10 -DDR3=1 -DDG=2
  run $mode -DDV=10 -DDR1=10 -DDR2=10 -DDR3=1 -DDG=2 "$W4"
  run $mode -DDV=10 -DDR1=10 -DDR2=10 -DDR3=5 -DDG=1
  run $mode -DDV=10 -DDR1=10 -DDR2=10 -DDR3=5 -DDG=1 "$W3"
  run $mode -DDV=8 -DDR1=8 -DDR2=8 -DDR3=8 -DDG=1
  run $mode -DDV=8 -DDR1=8 -DDR2=8 -DDR3=8 -DDG=1 "$W4"
  run $mode -DDV=16 -DDR1=16 -DDR2=16 -DDR3=1 -DDG=1
  run $mode -DDV=8 -DDR1=8 -DDR2=8 -DDR3=4 -DDG=1
  run $mode -DDV=8 -DDR1=8 -DDR2=8 -DDR3=4 -DDG=2 "$W4"
done
rm -rf gpurun_out/k1s
cat $out
