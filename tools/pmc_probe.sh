#!/bin/bash
# Counter passes of any probe binary (one --pmc group per run): tools/pmc_probe.sh <out-name> <binary> [args...]
set -u
export TMPDIR=/tmp
NAME=$1; shift
OUT=gpurun_out/pmc_$NAME
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o p --output-format csv -- "$@" > $OUT/g$i.log 2>&1
  echo "group $i ($grp): rc=$?"
done
python - $OUT "$@" <<'PY'
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
with open(out + "/summary.txt", "w") as o:
    o.write("# command: " + " ".join(sys.argv[2:]) + " (per-dispatch means)\n")
    for k, d in agg.items():
        if "fill" in k or "copyBuffer" in k:
            continue
        o.write(k + (f"   [kernel-trace mean duration {sum(dur[k])/len(dur[k]):.3f} ms, n={len(dur[k])}]" if dur.get(k) else "") + "\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/summary.txt").read())
PY
