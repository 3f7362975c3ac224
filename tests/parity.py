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
