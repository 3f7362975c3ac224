"""c4 wavelet (128 ch x 16384 samples, Morlet w0 = 6, 25 scales 4..100 Hz, pow): direct kernels against the staged path,
trial-sum (accumulate=2) and keeptrials modes.  python tools/wav_probe.py [trials]"""
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from syncopy_amd import backend as be, synthdata

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C4, N4 = 128, 16384
d4 = synthdata.ar2_uncoupled_fast(C4, N4, T, seed=77)
tr = torch.arange(T, device="cuda", dtype=torch.int64) * N4
foi = np.arange(4, 104, 4, dtype=float)
scales = (1 / foi) * (6 + np.sqrt(38)) / (4 * np.pi)


def ms(fn, reps=2):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for direct in (True, False):
    plan = be.CWTPlan(N4, C4, scales, 1e-3, 6.0, 0, "pow")
    plan.set_direct(direct)
    out = torch.zeros(plan.out_shape(1), dtype=torch.float32, device="cuda")
    t_sum = ms(lambda: plan.execute(d4, tr, tr, tr + N4, out=out, accumulate=2))
    res[direct] = out.clone()
    nk = min(T, 16)
    outk = torch.empty(plan.out_shape(nk), dtype=torch.float32, device="cuda")
    t_keep = ms(lambda: plan.execute(d4, tr[:nk].contiguous(), tr[:nk].contiguous(), (tr[:nk] + N4).contiguous(), out=outk))
    print("direct=%d  trial sum %.1f us/trial   keeptrials %.1f us/trial (%d trials)" % (direct, 1e3 * t_sum / T, 1e3 * t_keep / nk, nk), flush=True)
    del plan, out, outk
a, b = res[True].double(), res[False].double()
print("trial sums: max |direct - staged| / max = %.3g" % float((a - b).abs().max() / b.abs().max()))
