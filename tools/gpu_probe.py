"""Quick on-GPU timing probe of the hot kernels (development aid; results go to gpurun_out/)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.signal import windows  # noqa: E402

from syncopy_amd import backend as be  # noqa: E402


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


def main():
    res = {}
    C, N, K = 256, 4096, 7
    B = int(os.environ.get("PROBE_TRIALS", "64"))
    data = torch.randn(B * N, C, device="cuda", dtype=torch.float32)
    starts = torch.arange(B, device="cuda", dtype=torch.int64) * N
    tapers = windows.dpss(N, 4.096, K) * np.sqrt(N)
    scale = np.sqrt(2) / N
    for G in (1, 2, 4):
        os.environ["SPYHIP_FFT_G"] = str(G)
        for output, keep in (("pow", False), ("fourier", True)):
            plan = be.FFTPlan(N, N, C, tapers, scale, 0, False, None, output, keep)
            out = torch.empty(plan.out_shape(B), dtype=plan.out_dtype, device="cuda")
            ms = timeit(lambda: plan.execute(data, starts, out=out))
            res[f"mtmfft_G{G}_{output}"] = {"ms": ms, "us_per_trial": 1e3 * ms / B, "kernel": plan.kernel_name}
            print(f"G={G} {output}: {ms:.3f} ms for {B} trials = {1e3*ms/B:.2f} us/trial", flush=True)
    # CSD accumulate
    F = N // 2 + 1
    Bc = min(B, 32)
    spec = torch.randn(Bc * K, F, C, device="cuda", dtype=torch.float32).to(torch.complex64) * (1 + 1j)
    spec = torch.view_as_complex(torch.randn(Bc * K, F, C, 2, device="cuda", dtype=torch.float32))
    acc = torch.zeros(F, C, C, dtype=torch.complex64, device="cuda")
    ms = timeit(lambda: be.csd_accumulate(spec, acc), n=3, warm=1)
    flops = 8.0 * Bc * K * F * C * (C + 1) / 2
    res["csd_accumulate"] = {"ms": ms, "us_per_trial": 1e3 * ms / Bc, "tflops_alg": flops / ms / 1e9}
    print(f"csd_accumulate: {ms:.3f} ms for {Bc} trials = {1e3*ms/Bc:.2f} us/trial, {flops/ms/1e9:.1f} TF (algorithmic)",
          flush=True)
    ms = timeit(lambda: be.coh_normalize(acc, "abs"), n=3, warm=1)
    res["coh_normalize"] = {"ms": ms}
    print(f"coh_normalize: {ms:.3f} ms", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/probe.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
