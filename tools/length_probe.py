"""mtmfft pow / taper-mean timing over FFT lengths (development aid): which kernel serves a length, and how fast."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.signal import windows  # noqa: E402

from syncopy_amd import backend as be  # noqa: E402

C, K = 256, int(os.environ.get("LP_K", "7"))
for N in [int(a) for a in sys.argv[1:]] or [1000, 2000, 3000, 4096, 5000, 8192, 10000, 16384]:
    T = max(8, min(200, (1 << 28) // (N * C)))
    data = torch.randn((T * N, C), device="cuda", dtype=torch.float32)
    starts = torch.arange(T, device="cuda", dtype=torch.int64) * N
    try:
        plan = be.FFTPlan(N, N, C, windows.dpss(N, 4.0, K) * np.sqrt(N), np.sqrt(2) / N, 0, False, None, "pow", False)
    except Exception as exc:          # lengths beyond what the LDS kernels support
        print(N, "unsupported:", str(exc)[:80])
        continue
    out = torch.empty(plan.out_shape(T), dtype=torch.float32, device="cuda")
    plan.execute(data, starts, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        plan.execute(data, starts, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"N={N:6d} {plan.kernel_name:44s} {1e6 * dt / T:9.1f} us/trial  {1e9 * dt / (T * N * C):7.3f} ns/channel-sample")
