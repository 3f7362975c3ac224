"""Instruction histogram of one kernel, phase by phase (development aid; VERDICT r3 "next" 4: the floor as a measurement).

    python tools/isa_histogram.py <file.s> <mangled kernel name> [--loop]

<file.s> comes from `hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isyncopy_amd/csrc -Iinclude -S --cuda-device-only
syncopy_amd/csrc/mtmfft.hip -o mtmfft.s`.  Phases are the stretches between s_barrier instructions (the FFT kernels
alternate register butterflies and LDS exchanges); classes: packed fp32 vector (v_pk_*), fp64 vector (v_*_f64),
other vector ALU, LDS, global / scratch memory, scalar, waits.  With --loop only the body of the hottest loop (the
backward branch that spans the most instructions = the taper loop) is counted and its per-iteration totals are given.
Issue cost model of profiles/r2_ubench_valu_lds.txt: a packed op ~5.6 cycles and a plain vector op ~2.8 cycles per wave
at two waves per SIMD, an fp64 op 4 cycles."""
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_") and op.endswith("_f64") or "_f64_" in op:
        return "v_f64"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "global"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "scalar"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    loop_only = "--loop" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = []            # (label or None, opcode)
    for ln in lines[start + 1:end]:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")):
            if re.match(r"^\.LBB\d+_\d+:", t):
                body.append((t.split(":")[0], None))
            continue
        if re.match(r"^\.?LBB\d+_\d+:", t):
            body.append((t.split(":")[0], None))
            continue
        body.append((None, t.split()[0] + (" " + t.split()[1] if t.split()[0].startswith("s_cbranch") or t.split()[0] == "s_branch" else "")))
    labels = {lab: i for i, (lab, _) in enumerate(body) if lab}
    lo, hi = 0, len(body)
    if loop_only:
        best = (0, 0, 0)
        for i, (_, op) in enumerate(body):
            if op and op.startswith(("s_cbranch", "s_branch")):
                tgt = op.split()[-1]
                if tgt in labels and labels[tgt] < i and i - labels[tgt] > best[0]:
                    best = (i - labels[tgt], labels[tgt], i)
        lo, hi = best[1], best[2] + 1
    phases, cur = [], {}
    for _, op in body[lo:hi]:
        if op is None:
            continue
        c = classify(op.split()[0])
        if c == "barrier":
            phases.append(cur)
            cur = {}
            continue
        cur[c] = cur.get(c, 0) + 1
    phases.append(cur)
    cols = ["v_pk", "v_f64", "v_other", "mfma", "lds", "global", "scratch", "scalar", "wait"]
    print(f"# {name}{' (hottest loop body)' if loop_only else ''}: {hi - lo} lines, {len(phases)} phases (split at s_barrier)")
    print("phase " + " ".join(f"{c:>8s}" for c in cols) + "   issue-cycle estimate")
    tot = {c: 0 for c in cols}
    for k, ph in enumerate(phases):
        est = 5.6 * ph.get("v_pk", 0) + 4.0 * ph.get("v_f64", 0) + 2.8 * ph.get("v_other", 0) + 2.8 * ph.get("lds", 0)
        print(f"{k:5d} " + " ".join(f"{ph.get(c, 0):8d}" for c in cols) + f"   {est:9.0f}")
        for c in cols:
            tot[c] += ph.get(c, 0)
    est = 5.6 * tot["v_pk"] + 4.0 * tot["v_f64"] + 2.8 * tot["v_other"] + 2.8 * tot["lds"]
    print("total " + " ".join(f"{tot[c]:8d}" for c in cols) + f"   {est:9.0f}")


if __name__ == "__main__":
    main()
