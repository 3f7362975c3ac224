"""Parameter -> kernel-argument mapping of the frontends (rows F1-F3 of SURVEY.md section 8a).

Same names, argument meaning and error behaviour as syncopy/shared/input_processors.py
(process_padding:26, process_foi:93, process_taper:178) and
syncopy/specest/mtmfft.py:132 (_get_dpss_pars)."""
import numbers
from inspect import signature

import numpy as np
from scipy.signal import windows

from .const_def import availablePaddingOpt, availableTapers
from .errors import SPYInfo, SPYValueError, SPYWarning


def _nextpow2(number):
    n = 1
    while n < number:
        n *= 2
    return n


def _get_dpss_pars(tapsmofrq, nSamples, samplerate):
    NW = tapsmofrq * nSamples / samplerate
    Kmax = int(2 * NW - 1)
    return NW, (Kmax if Kmax > 1 else 1)


def process_padding(pad, lenTrials, samplerate):
    """Total number of samples of every trial after padding."""
    lenTrials = np.asarray(lenTrials)
    ok = isinstance(pad, (numbers.Number, str)) and not isinstance(pad, bool)
    if ok and isinstance(pad, str) and pad not in availablePaddingOpt:
        ok = False
    if not ok:
        raise SPYValueError("'maxperlen', 'nextpow2' or a float number", varname="pad", actual=f"{pad}")
    if isinstance(pad, numbers.Number):
        if not (lenTrials.max() / samplerate <= pad < np.inf):
            raise SPYValueError(f"value to be greater or equals {lenTrials.max() / samplerate}", varname="pad",
                                actual=f"{pad}")
        return int(pad * samplerate)
    if pad == "nextpow2":
        return _nextpow2(int(lenTrials.max()))
    abs_pad = int(lenTrials.max())
    if lenTrials.min() != lenTrials.max():
        SPYInfo(f"Unequal trial lengths present, padding all trials to {abs_pad} samples")
    return abs_pad


def _check_array(arr, varname, lims, n=None):
    a = np.asarray(arr, dtype=float)
    if a.ndim != 1 or (n is not None and a.size != n):
        raise SPYValueError("1d array" + (f" of length {n}" if n else ""), varname=varname, actual=f"shape {a.shape}")
    if not np.all(np.isfinite(a)):
        raise SPYValueError("finite values", varname=varname, actual="inf/nan")
    if a.min() < lims[0] or a.max() > lims[1]:
        raise SPYValueError(f"all array elements to be bounded by {lims[0]} and {lims[1]}", varname=varname,
                            actual=f"array with range {a.min()} to {a.max()}")
    return a


def process_foi(foi, foilim, samplerate):
    if foi is not None and foilim is not None:
        raise SPYValueError("either `foi` or `foilim` specification", varname="foi/foilim", actual="both")
    if foi is not None:
        if isinstance(foi, str):
            if foi != "all":
                raise SPYValueError("'all' or `None` or list/array", varname="foi", actual=foi)
            foi = None
        else:
            foi = _check_array(foi, "foi", [0, samplerate / 2])
    if foilim is not None:
        if isinstance(foilim, str):
            if foilim != "all":
                raise SPYValueError("'all' or `None` or `[fmin, fmax]`", varname="foilim", actual=foilim)
            foilim = None
        else:
            foilim = [float(f) for f in _check_array(foilim, "foilim", [0, samplerate / 2], n=2)]
            if foilim[0] > foilim[1]:
                foilim = list(np.sort(foilim))
    return foi, foilim


def process_taper(taper, taper_opt, tapsmofrq, nTaper, keeptapers, foimax, samplerate, nSamples, output):
    """Taper validation and Slepian parameters; returns (taper, taper_opt)."""
    if taper == "dpss":
        raise SPYValueError("set `tapsmofrq` parameter directly for multi-tapering", varname="taper", actual=taper)
    if taper is None and tapsmofrq is None:
        return None, {}
    if taper not in availableTapers:
        raise SPYValueError("one of " + ", ".join(availableTapers), varname="taper", actual=taper)
    if not isinstance(taper_opt, (dict, type(None))):
        raise SPYValueError("dict or None", "taper_opt", type(taper_opt))

    if tapsmofrq is None:
        if nTaper is not None:
            SPYWarning("`nTaper` is only used for multi-tapering!")
        if keeptapers:
            SPYWarning("`keeptapers` is only used for multi-tapering!")
        supported = [k for k in signature(getattr(windows, taper)).parameters if k not in ("M", "sym")]
        if taper_opt is not None:
            if not supported:
                raise SPYValueError(f"`None`, taper '{taper}' has no additional parameters", varname="taper_opt",
                                    actual=taper_opt)
            for key in taper_opt:
                if key not in supported:
                    raise SPYValueError(f"one of {supported} for `taper='{taper}'`", "taper_opt key", key)
            for key in supported:
                if key not in taper_opt:
                    raise SPYValueError(f"additional parameter '{key}' for `taper='{taper}'`", "taper_opt", None)
            return taper, taper_opt
        if supported:
            raise SPYValueError(f"additional parameters for taper '{taper}': {supported}", varname="taper_opt",
                                actual=taper_opt)
        return taper, {}

    # multi-tapering
    if taper != "hann":
        raise SPYValueError("`None` for multi-tapering, just set `tapsmofrq`", varname="taper", actual=taper)
    if taper_opt is not None:
        SPYWarning("For multi-tapering use `tapsmofrq` and `nTaper`, `taper_opt` has no effect")
    if not keeptapers and output != "pow":
        raise SPYValueError(f"'pow'|False or '{output}'|True, set either keeptapers=True or `output='pow'`!",
                            varname="output|keeptapers", actual=f"'{output}'|{keeptapers}")
    minBw = samplerate / nSamples
    maxBw = np.min([samplerate / 2 - 1 / nSamples, samplerate * (nSamples + 1) / (2 * nSamples)])
    if not isinstance(tapsmofrq, numbers.Number) or isinstance(tapsmofrq, bool) or not (0 <= tapsmofrq < np.inf):
        raise SPYValueError("smoothing bandwidth in Hz, typical values are in the range 1-10Hz", varname="tapsmofrq",
                            actual=tapsmofrq)
    if tapsmofrq < minBw:
        SPYInfo(f"Setting tapsmofrq to the minimal attainable bandwidth of {minBw:.2f}Hz")
        tapsmofrq = minBw
    if tapsmofrq > maxBw:
        SPYInfo(f"Setting tapsmofrq to the maximal attainable bandwidth of {maxBw:.2f}Hz")
        tapsmofrq = maxBw
    NW, Kmax = _get_dpss_pars(tapsmofrq, nSamples, samplerate)
    if nTaper is None:
        SPYInfo(f"Using {Kmax} taper(s) for multi-tapering")
        return "dpss", {"NW": NW, "Kmax": Kmax}
    if not (isinstance(nTaper, numbers.Number) and int(nTaper) == nTaper and nTaper >= 1):
        raise SPYValueError("integer >= 1", varname="nTaper", actual=nTaper)
    if nTaper != Kmax:
        SPYWarning(f"Manually setting the number of tapers is not recommended; the optimal number is {Kmax}, "
                   f"you have chosen {nTaper}.")
    return "dpss", {"NW": NW, "Kmax": int(nTaper)}
