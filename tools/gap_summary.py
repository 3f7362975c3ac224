"""Summary of the SPY_FUZZ_GAP_LOG files of the two fuzz sweeps (tests/test_gpu_fuzz.py::_log_detrend_gap) ->
profiles/r4_fuzz_detrend_gap.txt.  Usage: python tools/gap_summary.py gap_a.tsv gap_b.tsv > profiles/r4_fuzz_detrend_gap.txt"""
import sys

import numpy as np

rows = []
for path in sys.argv[1:]:
    for line in open(path):
        parts = line.rstrip("\n").split("\t", 2)
        if len(parts) == 3:           # (descriptions that held a newline continue on the next line)
            rows.append((float(parts[0]), float(parts[1]), parts[2]))
rows.sort(key=lambda r: -r[0])
a = np.array([r[0] for r in rows])
b = np.array([r[1] for r in rows])
big = a > 1.0
print("# VERDICT r3 item 9: the polyremoval=1 (linear detrending) comparisons of the two seeded sweeps")
print("# (SPY_FUZZ_SCALE=20 and SPY_FUZZ_SCALE=10 SPY_FUZZ_OFFSET=100000; batched and per-trial route each) WITHOUT the")
print("# detrend widening of tests/test_gpu_fuzz.py::_check.")
print("# column 1: max |kernels - reference| / (1e-5 |ref| + floor max|ref|)   (the plain criterion)")
print("# column 2: max |reference - oracle with a float64 least-squares fit| / the same tolerance")
print("#           (the reference's OWN float32 scipy.linalg.lstsq rounding: LAPACK sgelsd on OpenBLAS kernels)")
print(f"# {len(rows)} comparisons; {int(big.sum())} above 1.0 without the widening; max {a.max():.2f}; median {np.median(a):.3f}")
if big.any():
    print(f"# max |column 1 - column 2| / column 2 over the cases above 1.0: {np.max(np.abs(a[big] - b[big]) / b[big]):.3f}"
          "  (the whole gap is the reference's own fit)")
for r in rows:
    print(f"{r[0]:.4g}\t{r[1]:.4g}\t{r[2]}")
