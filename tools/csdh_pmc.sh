#!/bin/bash
# Counter passes of a torch-free probe binary (one --pmc group per run): tools/csdh_pmc.sh <tag> <binary> [args...]
# writes gpurun_out/pmc_<tag>.txt: per-kernel means of every counter + the kernel-trace durations of the first pass
set -u
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $out/g$i -o p --output-format csv -- "$@" > $out/g$i.log 2>&1
  echo "group $i ($grp): rc=$?"
done
python3 - $out "$@" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/g1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][-70:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
with open(out + ".txt", "w") as o:
    o.write("# " + " ".join(sys.argv[2:]) + "  (per-dispatch means; durations in ms from the first counter pass)\n")
    for k, d in agg.items():
        o.write(k + (f"   n={len(dur[k])} mean_ms={sum(dur[k])/len(dur[k]):.4f} min_ms={min(dur[k]):.4f}" if dur[k] else "") + "\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + ".txt").read())
PY
