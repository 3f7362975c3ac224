"""Instruction mix of csdh_kernel between barriers (development aid): python tools/csdh_hist.py <file.s> [kernel substring]"""
import collections
import sys

name = sys.argv[2] if len(sys.argv) > 2 else "csdh_kernel"
lines = [l.strip() for l in open(sys.argv[1])]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("s_endpgm"))
ins = [l.split()[0] for l in lines[start:end] if l and not l.startswith((".", ";", "/")) and not l.endswith(":")]
segs, cur = [], []
for op in ins:
    cur.append(op)
    if op == "s_barrier":
        segs.append(cur)
        cur = []
segs.append(cur)
for i, s in enumerate(segs):
    c = collections.Counter()
    for op in s:
        k = ("mfma" if op.startswith("v_mfma") else "ds_read" if op.startswith("ds_read") else "ds_write" if op.startswith("ds_write")
             else "gload" if op.startswith("global_load") else "gstore" if op.startswith("global_store") else "v_mov" if op.startswith("v_mov")
             else "valu" if op.startswith("v_") else "wait" if op.startswith("s_waitcnt") else "nop" if op.startswith("s_nop")
             else "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_") else "other")
        c[k] += 1
    print(i, len(s), dict(sorted(c.items())))
