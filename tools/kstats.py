import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    if "spy" in r["Name"]:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} total {float(r["TotalDurationNs"]) / 1e6:9.3f} ms')
        tot += float(r["TotalDurationNs"])
print("total", tot / 1e6, "ms")
