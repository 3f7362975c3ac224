"""Float32 vs reference-precision (float64) tapered FFT, per transform length (development aid, `profiles/r4_precision_probe.txt`):
us/trial at 256 channels x 7 tapers for the power spectrum with taper mean (BASELINE c2's shape) and for the complex
spectra of every taper (the front half of the coherence path), the kernel that serves each, the ratio float64 / float32,
and the same through the FIRST-generation float64 kernels (SPYHIP_F64_OLD=1) whose results the new ones must reproduce."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.signal import windows  # noqa: E402

from syncopy_amd import backend as be  # noqa: E402

C, K = int(os.environ.get("PP_C", "256")), int(os.environ.get("PP_K", "7"))


def timed(plan, data, starts, out, reps=3):
    plan.execute(data, starts, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(data, starts, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for N in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 8192, 16384, 200, 500, 1000, 2000, 2500, 4000, 5000, 10000, 1009]:
    T = max(8, min(200, (1 << 27) // (N * C)))
    g = torch.Generator(device="cuda").manual_seed(N)
    data = torch.randn((T * N, C), device="cuda", dtype=torch.float32, generator=g)
    starts = torch.arange(T, device="cuda", dtype=torch.int64) * N
    tapers = windows.dpss(N, 4.0, K) * np.sqrt(N) if K > 1 else windows.hann(N)[None]
    for output, keeptapers in (("pow", False), ("fourier", True)):
        res, line = {}, f"N={N:6d} {output:7s}"
        for kind in ("f32", "f64", "f64old"):
            if kind == "f64old":
                os.environ["SPYHIP_F64_OLD"] = "1"
            try:
                plan = be.FFTPlan(N, N, C, tapers, np.sqrt(2) / N, 0, False, None, output, keeptapers, reference_mean=True)
                if kind != "f32" and not plan.set_precision(True):
                    line += f" | {kind}: unsupported"
                    continue
                out = torch.empty(plan.out_shape(T), dtype=plan.out_dtype, device="cuda")
                dt = timed(plan, data, starts, out)
                res[kind] = (dt, out.clone() if T * N * C <= (1 << 26) else None, plan.kernel_name)
                line += f" | {kind}: {1e6 * dt / T:8.1f} us/trial"
            finally:
                os.environ.pop("SPYHIP_F64_OLD", None)
        if "f32" in res and "f64" in res:
            line += f" | f64/f32 = {res['f64'][0] / res['f32'][0]:.2f}"
        if "f64" in res and "f64old" in res and res["f64"][1] is not None and res["f64old"][1] is not None:
            a, b = res["f64"][1], res["f64old"][1]
            line += f" | max |new - old| / max |old| = {float((a - b).abs().max() / b.abs().max()):.1e}"
        print(line, flush=True)
        if output == "pow":
            print("        kernels:", " ; ".join(f"{k}: {v[2]}" for k, v in res.items()), flush=True)
