#!/bin/bash
set -u
mkdir -p gpurun_out/k1s
out=gpurun_out/k1_session.txt; : > $out
FL="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value"
b() { hipcc $FL "$@" 2>gpurun_out/k1s/err.txt || { echo "BUILD FAILED: $*" >> $out; tail -3 gpurun_out/k1s/err.txt >> $out; return 1; }; }
mode="-DPOUTK=0 -DPMEAN=1"
i=0
for opt in "" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -misched-topdown" "-mllvm -misched-bottomup" "-mllvm -enable-post-misched=0" "-mllvm -amdgpu-use-amdgpu-trackers=1" "-mllvm -amdgpu-igrouplp=0" "-ffast-math" "-O2" "-mllvm -amdgpu-early-inline-all=true" ; do
  echo -n "[quad G=1 '$opt'] " >> $out
  b -DPQUAD=1 $mode $opt tools/fft1_probe.hip -o gpurun_out/k1s/q && timeout 120 gpurun_out/k1s/q 500 >> $out 2>&1
done
rm -rf gpurun_out/k1s
cat $out
