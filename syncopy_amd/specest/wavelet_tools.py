"""Morlet scale/period conversions (specest/wavelets/wavelets.py:89-101) and the
'optimal' scale set of Torrence & Compo as used by the reference (specest/wavelet.py:52-106)."""
import numpy as np


def morlet_scale_from_period(period, w0=6.0):
    return (period * (np.sqrt(w0 * w0 + 2) + w0)) / (4.0 * np.pi)


def morlet_fourier_period(s, w0=6.0):
    return 4 * np.pi * s / (w0 + (2 + w0 ** 2) ** 0.5)


def optimal_wavelet_scales(nSamples, dt, w0=6.0, dj=0.25, s0=None):
    if s0 is None:
        s0 = morlet_scale_from_period(2 * dt, w0)
    J = int((1 / dj) * np.log2(nSamples * dt / s0))
    return (s0 * 2 ** (dj * np.arange(0, J + 1)))[::-1]
