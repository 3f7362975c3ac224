"""Kernel resource table from hipcc's -Rpass-analysis=kernel-resource-usage remarks:  python tools/kres.py <file.hip> [filter]"""
import re, subprocess, sys
src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Iinclude",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"remark: (?:Function )?Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if pat in k:
        print("%-90s vgpr %3d agpr %3d spill %3d sgpr-spill %3d occ %d scratch %d" % (k[:90], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("VGPRs Spill", -1),
              v.get("SGPRs Spill", -1), v.get("Occupancy [waves/SIMD]", -1), v.get("ScratchSize [bytes/lane]", -1)))
