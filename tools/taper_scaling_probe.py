"""us/trial of the transform kernel against the taper count: t = a + K b separates the per-segment part (load, detrend,
final stores) from the per-taper part.  PYTHONPATH=. python tools/taper_scaling_probe.py N [N ...]"""
import os
import sys
import numpy as np
import torch
from scipy.signal import windows
from syncopy_amd import backend as be, synthdata

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _event_ms

C, T = 256, 200
for N in [int(a) for a in sys.argv[1:]] or [8192, 16384]:
    d = synthdata.ar2_uncoupled_fast(C, N, T, seed=78)
    st = torch.arange(T, device="cuda", dtype=torch.int64) * N
    res = {}
    for K in (1, 3, 7):
        tp = windows.dpss(N, 4.0, K) * np.sqrt(N)
        tp = tp.reshape(K, N)
        mode = os.environ.get("PROBE_MODE", "pow")
        if mode == "pow":
            plan = be.FFTPlan(N, N, C, tp, np.sqrt(2) / N, None, False, None, "pow", False)      # no detrending: no K0 pre-pass
        else:
            plan = be.FFTPlan(N, N, C, tp, np.sqrt(2) / N, None, False, None, "fourier", True)
        buf = torch.empty(plan.out_shape(T), dtype=torch.complex64 if mode != "pow" else torch.float32, device="cuda")
        res[K] = 1e3 * _event_ms(torch, lambda: plan.execute(d, st, out=buf), reps=5) / T
        name = plan.kernel_name
        del plan, buf
    b = (res[7] - res[1]) / 6
    print(N, name, "K=1/3/7: %.1f %.1f %.1f us/trial -> per taper %.2f, per segment %.2f" % (res[1], res[3], res[7], b, res[1] - b), flush=True)
