# per-kernel times of the Granger / Wilson stage (c5: 256 channels, 2049 frequencies): bash tools/wilson_profile.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-wil}
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o w --output-format csv -- python tools/config_probe.py granger > gpurun_out/$tag.log 2>&1
grep "^granger" gpurun_out/$tag.log
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "spywil" in r["Name"]: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
rm -f gpurun_out/prof_$tag/*kernel_trace.csv
