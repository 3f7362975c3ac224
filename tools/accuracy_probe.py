"""Error of the float32 FFT kernels against a float64 evaluation of the same formula, per transform-length class
(development aid): python tools/accuracy_probe.py [N ...]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from syncopy_amd import backend as be
from oracle import spy_oracle as O

Ns = [int(a) for a in sys.argv[1:]] or [228, 500, 1009, 2000, 3000, 3001, 4096, 4100, 5003, 6000, 8192, 11000, 20000]
rng = np.random.default_rng(0)
for N in Ns:
    C, T = 4, 3
    x = rng.normal(size=(T * N, C)).astype(np.float32)
    tapers = O.taper_table("hann", N, N, {})
    scale = O.spec_scale(N, N)
    plan = be.FFTPlan(N, N, C, tapers, scale, None, False, None, "fourier", True)
    st = torch.arange(T, device="cuda", dtype=torch.int64) * N
    got = plan.execute(torch.from_numpy(x).cuda(), st, st, st + N).cpu().numpy()
    xs = x.astype(np.float64).reshape(T, N, C)
    w = np.asarray(tapers, dtype=np.float64)[0]
    ref = np.fft.rfft(xs * w[None, :, None], axis=1) * scale
    err = np.abs(got[:, 0].astype(np.complex128) - ref)
    rms = np.sqrt(np.mean(np.abs(ref) ** 2))
    print(f"N={N:6d} {plan.kernel_name[:60]:60s} max err/rms {err.max() / rms:.3g}  rms err/rms {np.sqrt(np.mean(err ** 2)) / rms:.3g}  "
          f"err at DC/rms {err[:, 0].max() / rms:.3g}", flush=True)

# ---- constant detrending of channels with an offset: the mean is NumPy's float32 row-order mean (reference_mean=True)
print("with offsets, detrend = 0 (reference-order mean)")
for N in [228, 1009, 4096, 4100, 5003]:
    for C in (1, 4):
        T = 3
        x = (rng.normal(size=(T * N, C)) + 5.0).astype(np.float32)
        tapers = O.taper_table("hann", N, N, {})
        scale = O.spec_scale(N, N)
        for rm in (True, False):
            plan = be.FFTPlan(N, N, C, tapers, scale, 0, False, None, "fourier", True, reference_mean=rm)
            st = torch.arange(T, device="cuda", dtype=torch.int64) * N
            got = plan.execute(torch.from_numpy(x).cuda(), st, st, st + N).cpu().numpy()
            xs = x.reshape(T, N, C)
            d = np.stack([xs[t] - np.mean(xs[t], axis=0) for t in range(T)]).astype(np.float64)   # float32 arithmetic, then exact
            w = np.asarray(tapers, dtype=np.float64)[0]
            ref = np.fft.rfft(d * w[None, :, None], axis=1) * scale
            err = np.abs(got[:, 0].astype(np.complex128) - ref)
            rms = np.sqrt(np.mean(np.abs(ref) ** 2))
            print(f"N={N:6d} C={C} ref_mean={rm!s:5s} {plan.kernel_name[:40]:40s} max err/rms {err.max() / rms:.3g} at bin {np.unravel_index(err.argmax(), err.shape)[1]}"
                  f"  DC {err[:, 0].max() / rms:.3g}  bin1 {err[:, 1].max() / rms:.3g}", flush=True)
