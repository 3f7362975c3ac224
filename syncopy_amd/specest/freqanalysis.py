"""`freqanalysis` metafunction: parameter resolution for mtmfft / mtmconvol / welch / wavelet / superlet.

Reproduces the parameter -> kernel-argument mapping of syncopy/specest/freqanalysis.py
(padding :459-461, foi :600-612, tapers :622-632, sliding windows :670-820,
wavelet scales :822-908) and then runs the matching compute class on the GPU.
"""
import numbers

import numpy as np

from ..datatype import AnalogData, SpectralData, selected_trialdefinition
from ..shared.const_def import availableMethods, spectralDTypes
from ..shared.errors import SPYInfo, SPYTypeError, SPYValueError, SPYWarning
from ..shared.kwarg_decorators import attached_selection, unwrap_cfg
from ..shared.input_processors import process_foi, process_padding, process_taper
from ..shared.tools import best_match
from .compRoutines import MultiTaperFFT, MultiTaperFFTConvol

availableWavelets = ("Morlet", "Paul", "DOG", "Ricker", "Marr", "Mexican_hat")      # freqanalysis.py:55


def _scalar(value, varname, lims):
    if not isinstance(value, numbers.Number) or isinstance(value, bool):
        raise SPYTypeError(value, varname=varname, expected="scalar")
    if not (lims[0] <= value <= lims[1]):
        raise SPYValueError(f"value to be greater or equals {lims[0]} and less or equals {lims[1]}", varname=varname,
                            actual=f"{value}")


@unwrap_cfg
def freqanalysis(data, method="mtmfft", output="pow", keeptrials=True, foi=None, foilim=None, pad="maxperlen",
                 polyremoval=0, taper="hann", demean_taper=False, taper_opt=None, tapsmofrq=None, nTaper=None,
                 keeptapers=False, toi="all", t_ftimwin=None, wavelet="Morlet", width=6, order=None, order_max=None,
                 order_min=1, c_1=3, adaptive=False, ft_compat=False, select=None, compute_method=None,
                 routine_classes=None, precision="auto", **kwargs):
    """Spectral estimation of AnalogData on MI355X.  Arguments as spy.freqanalysis
    (freqanalysis.py:62-90).  `compute_method`: None/'hip' (batched from the in-HBM trial
    queue) or 'sequential' (the reference's per-trial loop over the same kernels).
    `routine_classes` lets tests substitute compute classes (e.g. bound to the CPU oracle).
    `precision` (not a reference argument): "float32" transforms in float32 - ~1e-7 of a channel's largest bin;
    "reference" runs the taper product and the FFT in float64 and rounds to complex64 where the reference does
    (mtmfft.py:96-127): every bin to 1e-5 of itself, ~2x the time at the lengths with a compile-time schedule (powers of
    two 256 ... 16384, 200 ... 10000 decimal), 4-10x elsewhere, any transform length up to 2^20; for 'wavelet' / 'superlet'
    float64 FFT convolutions (~50x the float32 kernels; closer to the exact result than the reference itself, whose
    scipy.signal.fftconvolve transforms the float32 trial in single precision - never chosen by "auto").  "auto" (default): float64 on the per-trial route (compute_method="sequential": the copies
    dominate there) and, on the batched route, for the outputs that isolate a PART of the complex spectrum - 'fourier',
    'real', 'imag', 'angle', 'absreal', 'absimag' - where a small part next to a large one inherits the large one's
    float32 error; 'pow' and 'abs' stay float32 (inside the criterion by its floor)."""
    if precision not in ("float32", "reference", "auto"):
        raise SPYValueError("'float32', 'reference' or 'auto'", varname="precision", actual=str(precision))
    if not isinstance(data, AnalogData) or data.data is None:
        raise SPYTypeError(data, varname="data", expected="non-empty AnalogData")
    classes = {"mtmfft": MultiTaperFFT, "mtmconvol": MultiTaperFFTConvol}
    try:
        from .compRoutines import SuperletTransform, WaveletTransform
        classes["wavelet"] = WaveletTransform
        classes["superlet"] = SuperletTransform
    except ImportError:
        pass
    classes.update(routine_classes or {})
    timeAxis = data.dimord.index("time")
    if method not in availableMethods:
        raise SPYValueError("one of " + ", ".join(availableMethods), varname="method", actual=method)
    if output not in spectralDTypes:
        raise SPYValueError("one of " + ", ".join(spectralDTypes), varname="output", actual=output)
    for name, val in (("keeptrials", keeptrials), ("keeptapers", keeptapers), ("demean_taper", demean_taper),
                      ("ft_compat", ft_compat)):
        if not isinstance(val, bool):
            raise SPYTypeError(val, varname=name, expected="Bool")
    if polyremoval is not None:
        if not isinstance(polyremoval, numbers.Number) or polyremoval not in (0, 1):
            raise SPYValueError("0, 1 or None", varname="polyremoval", actual=polyremoval)
        polyremoval = int(polyremoval)

    from . import hip_spectral as hs
    with attached_selection(data, select):
        soft = (precision == "auto" and compute_method in (None, "hip") and method in ("mtmfft", "mtmconvol", "welch")
                and output not in ("pow", "abs"))
        with (hs.soft_reference() if soft else hs.precision(precision)):
            return _freqanalysis(data, classes, timeAxis, method, output, keeptrials, foi, foilim, pad, polyremoval, taper,
                                 demean_taper, taper_opt, tapsmofrq, nTaper, keeptapers, toi, t_ftimwin, wavelet, width,
                                 ft_compat, compute_method, (order_max, order_min, c_1, adaptive), order)


def _freqanalysis(data, classes, timeAxis, method, output, keeptrials, foi, foilim, pad, polyremoval, taper,
                  demean_taper, taper_opt, tapsmofrq, nTaper, keeptapers, toi, t_ftimwin, wavelet, width, ft_compat,
                  compute_method, slt=(None, 1, 3, False), order=None):
    fs = data.samplerate
    trl = selected_trialdefinition(data)
    sinfo = trl[:, :2]
    lenTrials = np.diff(sinfo).squeeze(axis=1)
    numTrials = lenTrials.size
    if data.selection is not None and isinstance(toi, (np.ndarray, list)) and any(
            data.selection.time[t] != (0, int(data.sampleinfo[t, 1] - data.sampleinfo[t, 0]))
            for t in data.selection.trial_ids):
        raise SPYValueError("no `toi` specification due to active in-place time-selection in input dataset",
                            varname="toi", actual=toi)
    tStart = trl[:, 2] / fs
    tEnd = tStart + lenTrials / fs

    if method in ("mtmconvol", "welch") and isinstance(pad, str) and pad != "maxperlen":
        SPYWarning("methods 'mtmconvol' and 'welch' only support in-place padding; `pad` will be ignored.")
    minSampleNum = process_padding(pad, lenTrials, fs) if method == "mtmfft" else lenTrials.min()
    minTrialLength = minSampleNum / fs
    dt = 1 / fs
    foi, foilim = process_foi(foi, foilim, fs)
    log_dct = {"method": method, "output": output, "keeptapers": keeptapers, "keeptrials": keeptrials,
               "polyremoval": polyremoval, "pad": pad}

    if "mtm" in method or method == "welch":
        if method in ("mtmconvol", "welch"):
            _scalar(t_ftimwin, "t_ftimwin", [dt, minTrialLength])
            minSampleNum = int(t_ftimwin * fs)
            if method == "welch":                       # freqanalysis.py:584-596
                if keeptapers:
                    raise SPYValueError("keeptapers='False' with method='welch'", varname="keeptapers",
                                        actual=keeptapers)
                if output != "pow":
                    raise SPYValueError("output='pow' with method='welch'", varname="output", actual=output)
        freqs = np.fft.rfftfreq(int(minSampleNum), dt)
        if foi is not None:
            foi, _ = best_match(freqs, foi, squash_duplicates=True)
        elif foilim is not None:
            foi, _ = best_match(freqs, foilim, span=True, squash_duplicates=True)
        else:
            foi = freqs
        if foi.size == 0:
            raise SPYValueError("non-empty frequency specification", varname="foi/foilim",
                                actual="empty frequency selection")
        taper, taper_opt = process_taper(taper, taper_opt, tapsmofrq, nTaper, keeptapers, foimax=foi.max(),
                                         samplerate=fs, nSamples=lenTrials.mean(), output=output)
        log_dct.update(foi=foi, taper=taper, taper_opt=taper_opt)

    if method == "mtmfft":
        method_kwargs = {"samplerate": fs, "taper": taper, "taper_opt": taper_opt, "nSamples": int(minSampleNum),
                         "demean_taper": demean_taper, "ft_compat": ft_compat}
        cr = classes["mtmfft"](foi=foi, timeAxis=timeAxis, keeptapers=keeptapers, polyremoval=polyremoval,
                               output=output, method_kwargs=method_kwargs)

    elif method in ("mtmconvol", "welch"):
        if method == "welch" and (isinstance(toi, str) or not isinstance(toi, numbers.Number)):
            raise SPYValueError("toi to be a float in range [0, 1] for method='welch'", varname="toi",
                                actual=str(toi))      # freqanalysis.py:684-701
        if isinstance(toi, str):
            if toi != "all":
                raise SPYValueError("`toi = 'all'` to center analysis windows on all time-points", varname="toi",
                                    actual=toi)
            equidistant, overlap = True, np.inf
        elif isinstance(toi, numbers.Number):
            _scalar(toi, "toi", [0, 1])
            equidistant, overlap = True, toi
        else:
            overlap = -1
            toi = np.array(toi, dtype=float)
            if toi.ndim != 1 or toi.size == 0 or not np.all(np.isfinite(toi)):
                raise SPYValueError("1d array of finite time-points", varname="toi", actual=str(toi.shape))
            if toi.min() < tStart.min() or toi.max() > tEnd.max():
                raise SPYValueError(f"all array elements to be bounded by {tStart.min()} and {tEnd.max()}",
                                    varname="toi", actual=f"array with range {toi.min()} to {toi.max()}")
            tSteps = np.diff(toi)
            if (tSteps < 0).any():
                raise SPYValueError("ordered list/array of time-points", varname="toi", actual="unsorted list/array")
            if tSteps.size and np.isclose(tSteps.min(), dt):
                tSteps[np.isclose(tSteps, dt)] = dt
            if tSteps.size and tSteps.min() < dt:
                SPYWarning(f"`toi` selection too fine, max. time resolution is {dt}s")
            equidistant = bool(np.allclose(tSteps, [tSteps[0]] * tSteps.size)) if tSteps.size else True
        nperseg = int(t_ftimwin * fs)
        # hop of the sliding window: a fraction of the window for `toi` in [0, 1], one sample otherwise
        noverlap = min(nperseg - 1, int(overlap * nperseg)) if 0 <= overlap <= 1 else nperseg - 1
        if overlap < 0:
            soi, postSelect, equidistant = _windows_at(toi, equidistant, nperseg, fs, tStart)
        else:
            soi, postSelect = [slice(None)] * numTrials, slice(None)
        method_kwargs = {"samplerate": fs, "nperseg": nperseg, "noverlap": noverlap, "taper": taper,
                         "taper_opt": taper_opt}
        cr = classes["mtmconvol"](soi, [postSelect] * numTrials, equidistant=equidistant, toi=toi, foi=foi,
                                  timeAxis=timeAxis, keeptapers=keeptapers, polyremoval=polyremoval, output=output,
                                  method_kwargs=method_kwargs)

    elif method == "wavelet":
        if "wavelet" not in classes:
            raise NotImplementedError("wavelet transform kernels are not part of this build")
        from .wavelet_tools import (WAVELET_FAMILY, family_fourier_period, family_scale_from_period,
                                    optimal_wavelet_scales)
        if wavelet not in availableWavelets:
            raise SPYValueError("one of " + ", ".join(availableWavelets), varname="wavelet", actual=wavelet)
        if wavelet not in ("Morlet", "Paul"):           # freqanalysis.py:830-836
            SPYWarning(f"the chosen wavelet '{wavelet}' is real-valued and does not provide any information about "
                       "amplitude or phase of the data. This wavelet function may be used to isolate peaks or "
                       "discontinuities in the signal. ")
        family, takes_order = WAVELET_FAMILY[wavelet]
        if wavelet == "Morlet":
            _scalar(width, "width", [1, np.inf])
        # (the reference's warning about `width` for the other wavelets compares the argument with itself,
        # freqanalysis.py:846-848, and never fires)
        # Reference quirk kept for parity: freqanalysis.py:844 builds Morlet(w0=width) but the
        # `order` branch at :861-864 then replaces it by a default-constructed Morlet(), so
        # `width` never reaches the transform and w0 is always 6.
        width = 6.0
        if wavelet == "Paul":
            _int_like(order, "order", 4, np.inf)        # :850-855
        elif wavelet == "DOG":
            _int_like(order, "order", 1, np.inf)        # :856-861
        elif order is not None:
            SPYWarning(f"option `order` has no effect for wavelet '{wavelet}'")
        m = int(order) if takes_order else (2 if family == "DOG" else None)      # Ricker = DOG(m=2), wavelets.py:352-361
        to_scale, to_period = family_scale_from_period(family, m, width), family_fourier_period(family, m, width)
        toi, preSelect, postSelect = _wavelet_toi(toi, numTrials, tStart, tEnd, lenTrials, fs)
        if foi is None and foilim is None:
            scales = optimal_wavelet_scales(int(minTrialLength * fs), dt, w0=width, scale_from_period=to_scale)
            foi = 1 / to_period(scales)
        else:
            if foilim is not None:
                foi = np.arange(foilim[0], foilim[1] + 1, dtype=float)
            foi = np.asarray(foi, dtype=float).copy()
            foi[foi < 0.01] = 0.01
            scales = to_scale(1 / foi)
        method_kwargs = {"samplerate": fs, "scales": scales, "w0": float(width)}
        if family != "Morlet":
            method_kwargs.update(family=family, order=m)
        cr = classes["wavelet"](preSelect, postSelect, toi=toi, timeAxis=timeAxis, polyremoval=polyremoval,
                                output=output, method_kwargs=method_kwargs)
        cr._foi = foi

    elif method == "superlet":
        if "superlet" not in classes:
            raise NotImplementedError("superlet transform kernels are not part of this build")
        from .wavelet_tools import optimal_wavelet_scales
        order_max, order_min, c_1, adaptive = slt
        if order_max is None:
            raise SPYValueError("Positive integer needed for order_max", varname="order_max", actual=None)
        _int_like(order_max, "order_max", 1, np.inf)
        _int_like(order_min, "order_min", 1, order_max)
        _int_like(c_1, "c_1", 1, np.inf)
        toi, preSelect, postSelect = _wavelet_toi(toi, numTrials, tStart, tEnd, lenTrials, fs)
        if foi is None and foilim is None:
            # scale_from_period(p) = p / (2 pi), fourier_period(s) = 2 pi s (superlet.py:295-308)
            scales = optimal_wavelet_scales(int(minTrialLength * fs), dt, scale_from_period=lambda p: p / (2 * np.pi))
            foi = 1 / (2 * np.pi * scales)
        else:
            if foilim is not None:
                foi = np.arange(foilim[0], foilim[1] + 1, dtype=float)
            foi = np.asarray(foi, dtype=float).copy()
            foi[foi < 0.01] = 0.01
            scales = (1.0 / foi) / (2 * np.pi)
        if adaptive:
            if len(scales) < 2:
                raise SPYValueError("A range of frequencies", varname="foi", actual="Single frequency")
            if np.any(np.diff(scales) > 0):
                SPYWarning("Sorting frequencies low to high for adaptive SLT..")
                scales = np.sort(scales)[::-1]
        method_kwargs = {"samplerate": fs, "scales": scales, "order_max": int(order_max), "order_min": int(order_min),
                         "c_1": int(c_1), "adaptive": bool(adaptive)}
        cr = classes["superlet"](preSelect, postSelect, toi=toi, timeAxis=timeAxis, polyremoval=polyremoval,
                                 output=output, method_kwargs=method_kwargs)
        cr._foi = foi

    out = SpectralData(dimord=SpectralData._defaultDimord)
    cr.initialize(data, out._stackingDim, chan_per_worker=None, keeptrials=keeptrials)
    from contextlib import ExitStack
    from . import hip_spectral as hs
    with ExitStack() as stack:
        if (hs.requested_precision() is None and compute_method in (None, "hip") and hasattr(cr, "compute_hip")
                and method in ("mtmfft", "mtmconvol", "welch") and _selection_hides_peak(data, cr, method, foi, freqs)):
            stack.enter_context(hs.soft_reference())      # precision="auto": kept bins far below the spectrum's peak
        if method == "welch":
            cr.reduce_time = True
        cr.compute(data, out, parallel=False, log_dict=log_dct, method=compute_method)
    if method == "welch":
        out = _time_mean(out, getattr(cr, "time_means", None))
    return out


def _selection_hides_peak(data, cr, method, foi, freqs):
    """See hip_spectral.selection_hides_peak: a handful of this rank's trials (mtmfft) or of the first trial's windows
    (mtmconvol / welch) through the float32 kernels, whole axis, against the bins the call keeps."""
    from .. import parallel
    from ..datatype import device_rows, selected_channels
    from . import hip_spectral as hs
    if foi is None or freqs is None or len(foi) >= len(freqs):
        return False
    _, fidx = best_match(freqs, foi, squash_duplicates=True)
    dev = data.device_data()
    rows, chans = device_rows(data), selected_channels(data)
    lo, hi = parallel.my_shard(len(rows))
    mine = rows[lo:hi]
    mk = cr.cfg["method_kwargs"]
    if method == "mtmfft":
        seg, nfft, pr, opt = mine, mk["nSamples"], cr.cfg["polyremoval"], mk["taper_opt"]
    else:
        n = mk["nperseg"]
        seg = [(a, a + n) for r0, r1 in mine[:1] for a in range(r0, r1 - n + 1, max(n, (r1 - r0) // 16))]
        # windows picked by a `toi` array are never detrended (compRoutines.py:392-408), sliding ones per frame
        pr = None if not cr.cfg.get("equidistant", True) else cr.cfg["polyremoval"]
        nfft, opt = n, dict(mk["taper_opt"] or {})
        if mk["taper"] == "dpss":
            opt["sym"] = False
    return hs.selection_hides_peak(dev, seg, chans, nfft, mk["taper"], opt, pr, fidx, len(freqs))


def _windows_at(toi, even, nperseg, fs, tStart):
    """Which samples of every trial the sliding-window analysis reads when `toi` lists explicit time points, and
    which of the resulting frames are kept (integers as freqanalysis.py:752-793 produces them).

    `centre[k, i]` is the (fractional) sample of trial k on which window i is centred.
      * closely spaced, evenly spaced points (largest step <= half a window): ONE stretch per trial from half a window
        before the first to half a window after the last point, transformed with hop 1; every `stride`-th frame is a
        requested point.  Returns (stretches, slice(None, None, stride), True).
      * anything else: one window per point, [centre - half, centre + half], shifted right by the part of the first
        window that would start before the trial.  Returns (lists of windows, slice(None), False)."""
    half = nperseg // 2
    centre = fs * (toi[None, :] - np.asarray(tStart)[:, None])
    steps = np.diff(toi)
    if even and not (steps.size and steps.max() * fs > half):
        first = np.maximum(0, np.rint(centre[:, 0] - half).astype(np.intp))
        last = np.rint(centre[:, -1] + half + 1).astype(np.intp)
        stride = max(1, int(np.rint((last[0] - first[0]) / toi.size)))
        return [slice(int(a), int(b)) for a, b in zip(first, last)], slice(None, None, stride), True
    lead = np.maximum(0, half - np.trunc(centre[:, 0]).astype(np.intp))[:, None]
    lo = np.trunc(centre - half).astype(np.intp) + lead
    hi = np.trunc(centre + half + 1).astype(np.intp) + lead
    hi = np.maximum(hi, hi - lo)                      # a window that starts before sample 0 keeps its length
    return [[slice(int(a), int(b)) for a, b in zip(r0, r1)] for r0, r1 in zip(lo, hi)], slice(None), False


def _wavelet_toi(toi, numTrials, tStart, tEnd, lenTrials, fs):
    """`toi` of the wavelet based methods (freqanalysis.py:511-560): 'all' or equidistant time-points -> per-trial
    pre-selection (interval to transform) and post-selection (samples to keep)."""
    preSelect, postSelect = [slice(None)] * numTrials, [slice(None)] * numTrials
    if isinstance(toi, str):
        if toi != "all":
            raise SPYValueError("`toi = 'all'` to center wavelets on all time-points", varname="toi", actual=toi)
    else:
        toi = np.array(toi, dtype=float)
        if toi.ndim != 1 or toi.size == 0:
            raise SPYValueError("1d array of time-points", varname="toi", actual=str(toi.shape))
        if toi.min() < tStart.min() or toi.max() > tEnd.max():
            raise SPYValueError(f"all array elements to be bounded by {tStart.min()} and {tEnd.max()}",
                                varname="toi", actual=f"array with range {toi.min()} to {toi.max()}")
        if toi.size > 2 and not np.allclose(np.diff(toi, 2), np.zeros(len(toi) - 2)):
            raise SPYValueError("array of equidistant time-points or 'all' for wavelet based methods",
                                varname="toi", actual=toi)
        preSelect, postSelect = [], []
        for tk in range(numTrials):
            start = int(fs * (toi[0] - tStart[tk]))
            stop = int(fs * (toi[-1] - tStart[tk]) + 1)
            preSelect.append(slice(max(0, start), max(stop, stop - start)))
            smpIdx = np.minimum(lenTrials[tk] - 1, fs * (toi - tStart[tk]) - start)
            postSelect.append(smpIdx.astype(np.intp))
    return toi, preSelect, postSelect


def _int_like(val, name, lo, hi):
    """scalar_parser(..., ntype="int_like") (shared/parsers.py:133-222)."""
    if isinstance(val, bool) or not isinstance(val, numbers.Number) or not np.isfinite(val) or int(val) != val:
        raise SPYTypeError(val, varname=name, expected="int_like scalar")
    if not lo <= val <= hi:
        raise SPYValueError(f"value to be greater or equals {lo} and less or equals {hi}", varname=name, actual=val)


def _time_mean(spec, rows=None):
    """`spy.mean(spec, dim="time")` as freqanalysis.py:1054-1056 applies it for method='welch':
    np.nanmean over the time axis of every trial (statistics/compRoutines.py:36-57, float32 in, float32 out);
    each trial keeps ONE stacked sample (trialdefinition [[k, k+1, 0]], statistics/compRoutines.py:98-117)."""
    td = np.asarray(spec.trialdefinition)
    if rows is None:                  # (`rows`: the same means taken on the device, MultiTaperFFTConvol.compute_hip)
        rows = np.concatenate([np.nanmean(spec.data[int(a):int(b)], axis=0, keepdims=True) for a, b in td[:, :2]], axis=0)
    n = len(rows)
    k = np.arange(n, dtype=float)[:, None]
    res = SpectralData(np.asarray(rows).astype(spec.data_dtype, copy=False), samplerate=spec.samplerate,
                       trialdefinition=np.hstack((k, k + 1, np.zeros((n, 1)))), dimord=spec.dimord)
    res.freq, res.taper, res.channel = spec.freq, spec.taper, spec.channel
    return res
