"""What the (time x channel) layout costs K1: the same number of (segment, channel quad) work items, tapers and bytes, once as
1000 segments of 256 channels (16-byte pieces of 1-KiB rows on the way in, 64-byte pieces of 2-KiB spectral rows on the way
out) and once as 32000 segments of 8 channels (the two quads of a workgroup ARE the row: loads and stores are contiguous
streams).  The difference bounds what any re-layout of the input (a channel-quad-major copy) or of the hand-over could
give.  python tools/k1_layout_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.signal import windows
from syncopy_amd import backend as be

N, K = 4096, 7
tapers = windows.dpss(N, 4.096, K) * np.sqrt(N)


def ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for C, T in ((256, 1000), (8, 32000)):
    g = torch.Generator(device="cuda").manual_seed(3)
    data = torch.randn((T * N, C), generator=g, device="cuda", dtype=torch.float32)
    starts = torch.arange(T, device="cuda", dtype=torch.int64) * N
    for output, keep, refmean in (("fourier", True, True), ("fourier", True, False), ("pow", False, True), ("pow", False, False)):
        plan = be.FFTPlan(N, N, C, tapers, np.sqrt(2) / N, 0, False, None, output, keep, reference_mean=refmean)
        out = torch.empty(plan.out_shape(T), dtype=plan.out_dtype, device="cuda")
        t = ms(lambda: plan.execute(data, starts, out=out))
        print("C = %3d x %5d segments  %-7s keeptapers=%d reference_mean=%d  %-44s %.3f ms = %.2f us per 256-channel trial"
              % (C, T, output, keep, refmean, plan.kernel_name, t, 1e3 * t / 1000), flush=True)
        del plan, out
    del data
