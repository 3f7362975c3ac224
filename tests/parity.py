"""Parity criterion shared by all tests:

    |a - b| <= rtol*|b| + atol_rel*max|b|      element-wise, rtol = 1e-5, atol_rel = 1e-6

`b` is the reference (oracle / golden vector).  Float32 kernels cannot do better
than ~4e-7 of the *typical* bin amplitude (measured, tests/test_emu_fft.py), so a
purely relative bound on near-zero bins is not meaningful; the atol term is
relative to the largest reference value.
"""
import numpy as np

RTOL = 1e-5
ATOL_REL = 1e-6


def excess(a, b, rtol=RTOL, atol_rel=ATOL_REL):
    """max over elements of |a-b| / (rtol|b| + atol_rel*max|b|); <= 1 means parity."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.iscomplexobj(a) == np.iscomplexobj(b), (a.dtype, b.dtype)
    if b.size == 0:
        return 0.0
    tol = rtol * np.abs(b) + atol_rel * np.abs(b).max()
    tol = np.where(tol == 0, np.finfo(np.float32).tiny, tol)
    err = np.abs(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64) - b)
    assert np.isfinite(err).all(), "non-finite values"
    return float((err / tol).max())


def assert_parity(a, b, rtol=RTOL, atol_rel=ATOL_REL, what=""):
    e = excess(a, b, rtol, atol_rel)
    assert e <= 1.0, f"{what}: parity violated, max err/tol = {e:.3g} (rtol={rtol}, atol_rel={atol_rel})"


def jackknife_tolerances(est, var, T, ulps=8.0):
    """Per-element error bounds of the jackknife statistics when every leave-one-out replicate r_t and the direct
    estimate carry an independent absolute error eps = ulps * 2^-24 * max|estimate| (float32 results of the AV stage;
    a coherency S_ij / sqrt(S_ii S_jj) is rounded at the scale of its unit diagonal, not of |C_ij|):

      var  = (T-1)/T sum_t (r_t - mean r)^2     ->  |d var|  <= 2 sqrt((T-1) var) eps      (Cauchy-Schwarz over t)
      bias = (T-1) (mean_t r_t - direct)        ->  |d bias| <= (T-1) (1 + 1/sqrt(T)) eps

    on top of the shared criterion rtol*|b| + atol_rel*max|b|.  The replicates of a 20-trial coherence differ from the
    direct estimate by 1e-3 ... 1e-2 of it, so var ~ 1e-6 est^2: the float32 rounding of the estimates IS 1e-3 of such a
    variance - which is what a flat rtol of 3e-3 used to stand for, now stated element by element."""
    est = np.abs(np.asarray(est)).astype(np.float64)
    var = np.abs(np.asarray(var)).astype(np.float64)
    eps = ulps * 2.0 ** -24 * (est.max() if est.size else 0.0)
    tol_var = RTOL * var + ATOL_REL * var.max() + 2.0 * np.sqrt((T - 1) * var) * eps
    tol_bias = (T - 1) * (1.0 + 1.0 / np.sqrt(T)) * eps
    return tol_var, tol_bias
