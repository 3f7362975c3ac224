# ablations of zinv64_mfma_kernel (development aid): per-call time with parts of the kernel switched off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for abl in 0 1 2 4 7; do
  export SPY_ZINV_ABL=$abl
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_abl$abl -o w --output-format csv -- python tools/config_probe.py granger > gpurun_out/abl$abl.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_abl$abl/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "zinv64" in r["Name"]: print("abl=$abl", r["Calls"], r["AverageNs"], r["MaxNs"])
PY
done
