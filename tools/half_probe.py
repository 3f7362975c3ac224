"""Quad form against HALF form (channel pairs through the schedule of N / 2) of the compile-time-schedule transform kernel,
per output mode.  PYTHONPATH=. python tools/half_probe.py   (SPYHIP_HALF_TRY is read at plan creation)"""
import os
import sys
import numpy as np
import torch
from scipy.signal import windows
from syncopy_amd import backend as be, synthdata

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _event_ms

C, K = 256, 7
for N in [int(a) for a in sys.argv[1:]] or [5000, 10000]:
    T = 200
    d = synthdata.ar2_uncoupled_fast(C, N, T, seed=78)
    tp = windows.dpss(N, 1.0 * N / 1000.0, K) * np.sqrt(N)
    st = torch.arange(T, device="cuda", dtype=torch.int64) * N
    for output, keep in (("pow", False), ("pow", True), ("fourier", True), ("fourier", False), ("abs", True)):
        row = []
        for half in (False, True):
            if half:
                os.environ["SPYHIP_HALF_TRY"] = "1"
            else:
                os.environ.pop("SPYHIP_HALF_TRY", None)
            plan = be.FFTPlan(N, N, C, tp, np.sqrt(2) / N, 0, False, None, output, keep, reference_mean=True)
            buf = torch.empty(plan.out_shape(T), dtype=torch.complex64 if output == "fourier" else torch.float32, device="cuda")
            ms = _event_ms(torch, lambda: plan.execute(d, st, out=buf))
            row.append((plan.kernel_name, 1e3 * ms / T))
            del plan, buf
        print(N, output, "keeptapers" if keep else "taper mean", "| %s: %.1f us/trial | %s: %.1f us/trial" %
              (row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
