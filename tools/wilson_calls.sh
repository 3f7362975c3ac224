# per-call durations of the Wilson kernels at 256 channels (development aid; traces stay in /tmp)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace -d /tmp/prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/config_probe.py granger > /tmp/log.txt 2>&1
grep "granger 256" /tmp/log.txt
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/prof/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "spywil" in n:
        d[n.split("(")[0][-28:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in d.items():
    big=[x for x in v if x>1.0]
    if big: print(k, len(big), "min %.2f med %.2f max %.2f sum %.1f"%(min(big), sorted(big)[len(big)//2], max(big), sum(big)))
PY
