#!/bin/bash
# K1 investigation session on the GPU box.  Output: gpurun_out/k1_session.txt
set -u
mkdir -p gpurun_out/k1s
out=gpurun_out/k1_session.txt; : > $out
FL="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value"
b() { hipcc $FL "$@" 2>gpurun_out/k1s/err.txt || { echo "BUILD FAILED: $*" >> $out; tail -5 gpurun_out/k1s/err.txt >> $out; return 1; }; }
for abl in 0 8 16 24 32 40 56 7 63; do
  echo -n "[quad abl=$abl] " >> $out
  b -DPQUAD=1 -DSPYFFT_ABL=$abl tools/fft1_probe.hip -o gpurun_out/k1s/q && timeout 120 gpurun_out/k1s/q 500 >> $out 2>&1
done
for abl in 0 7; do
  echo -n "[pair abl=$abl] " >> $out
  b -DPQUAD=0 -DSPYFFT_ABL=$abl tools/fft1_probe.hip -o gpurun_out/k1s/p && timeout 120 gpurun_out/k1s/p 500 >> $out 2>&1
done
b -DPQUAD=0 tools/fft1_probe.hip -o gpurun_out/k1s/p
bash tools/pmc_probe.sh pair gpurun_out/k1s/p 500 >> $out 2>&1
rm -rf gpurun_out/k1s
cat $out
