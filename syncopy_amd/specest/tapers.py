"""Window tables and spectral normalisation (host side, NumPy/SciPy only - shared by the PyTorch-backed front ends
and the NumPy-only host `syncopy_amd.abi`)."""
import numpy as np
from scipy.signal import windows

_taper_cache = {}


def taper_table(taper, nsig, nnorm, taper_opt=None):
    """(K, nsig) float64 window rows, normalised for spectral power
    (semantics of specest/_norm_spec.py:27-46; window length = actual signal
    length, normalisation length = padded length, mtmfft.py:96-101)."""
    taper = "boxcar" if taper is None else taper
    opt = {} if not taper_opt else dict(taper_opt)
    key = (taper, int(nsig), int(nnorm), tuple(sorted(opt.items())))
    if key not in _taper_cache:
        w = np.atleast_2d(getattr(windows, taper)(int(nsig), **opt)).astype(np.float64)
        if taper == "dpss":
            w = w * np.sqrt(nnorm)
        elif taper == "boxcar":
            w = w * np.sqrt(nnorm / w.sum())
        else:
            w = w * (np.sqrt(4 / 3) * np.sqrt(nnorm / w.sum()))
        _taper_cache[key] = w
    return _taper_cache[key]


def spec_scale(nsig, nfft, ft_compat=False):
    """sqrt(2)/N normalisation of every rfft bin (specest/_norm_spec.py:10-24, mtmfft.py:119-127)."""
    if ft_compat:
        return np.sqrt(2) / nfft
    return np.sqrt(2) / (nsig * np.sqrt(nfft / nsig))
