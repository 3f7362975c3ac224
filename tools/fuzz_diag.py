"""Where does a case of tests/test_gpu_fuzz.py disagree?  python tools/fuzz_diag.py mtmfft 5 9 25 ... (development aid)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pytest, types
import test_gpu_fuzz as T
import parity

kind = sys.argv[1]
fn = {"mtmfft": T.test_mtmfft_random_options, "conn": T.test_connectivity_random_options,
      "tf": T.test_timefrequency_random_options, "sel": T.test_mtmfft_selections_and_window_options,
      "welch": T.test_welch_and_superlet_random_options, "toi": T.test_timefrequency_toi_foi_offsets,
      "consel": T.test_connectivity_selections_and_spectral_input, "corr": T.test_corr_and_jackknife_random_options}[kind]
def report(got, ref, exact, what="", atol_rel=parity.ATOL_REL, rtol=parity.RTOL, scale=None):
    if getattr(got, "per_trial_route", None) is not None:
        report(got.per_trial_route, ref, exact, what + " [per-trial route]", atol_rel, rtol, scale)
    a = np.asarray(got.data); b = np.asarray(ref.data)
    tol = rtol * np.abs(b) + atol_rel * max(float(np.abs(b).max()), scale or 0.0)
    if exact is not None:
        base = tol.copy()
        e0 = np.abs(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64) - b)
        x = np.asarray(exact.data)
        ex = np.abs(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64) - x)
        k = np.unravel_index(np.argmax(e0 / base), base.shape)
        print(f"    WITHOUT the detrend widening: max err/tol {float((e0 / base).max()):.3g} at {tuple(int(v) for v in k)}; "
              f"there |ref - exact|/tol {float(np.abs(b - x)[k] / base[k]):.3g}; |got - exact|/tol max {float((ex / base).max()):.3g}")
        tol = tol + 2 * np.abs(b - np.asarray(exact.data))
    err = np.abs(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64) - b)
    r = err / np.where(tol == 0, 1e-38, tol)
    i = np.unravel_index(np.argmax(r), r.shape)
    nbad = int((r > 1).sum())
    bad_f = sorted(set(np.argwhere(r > 1)[:, -2].tolist()))[:12] if r.ndim >= 2 else []
    print(f"  {what}\n    shape {a.shape} max err/tol {r.max():.3g} at {tuple(int(x) for x in i)} got {a[i]} ref {b[i]} max|b| {np.abs(b).max():.4g}; "
          f"{nbad} elements over; axis -2 indices over: {bad_f}")
T._check = report
for s in sys.argv[2:]:
    print("seed", s)
    fn(int(s))
