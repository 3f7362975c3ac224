#!/bin/bash
# Counter passes of the two hot kernels through the torch-free harness (one --pmc group per run).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
hipcc -O2 tools/pmc_harness.cpp -Iinclude -Lsyncopy_amd -lspyhip -Wl,-rpath,$PWD/syncopy_amd -o gpurun_out/pmc/harness || exit 1
B=${1:-500}
BLK=${2:-0}
gpurun_out/pmc/harness $B 1 3 $BLK || exit 1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc/g$i -o p --output-format csv -- gpurun_out/pmc/harness $B 2 3 $BLK > gpurun_out/pmc/g$i.log 2>&1
  echo "group $i ($grp): rc=$?"
done
python - $B $BLK <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import sys
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
BLK = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with open("gpurun_out/pmc/summary.txt", "w") as out:
    sys.path.insert(0, ".")
    import bench
    out.write(f"# shape: trials={B} rows={B * 7} F=2049 C=256 blocked={BLK} k4_sources_sha={bench.k4_sources_sha()} "
              f"(per-dispatch means over the repeated launches)\n")
    for k, d in agg.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()):
            out.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open("gpurun_out/pmc/summary.txt").read())
PY
