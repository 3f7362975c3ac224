"""`spy.mean`: averages along a dimension of a data object or over its trials
(syncopy/statistics/summary_stats.py:24-52, 205-318 and statistics/compRoutines.py:22-143), on the device.

    spy.mean(data, dim="trials")                  # one trial: the sequential sum over the trials in the data's own
                                                  # dtype, then ONE division (summary_stats.py:321-400, :408-428)
    spy.mean(data, dim="time" | "freq" | "channel" | ..., keeptrials=True)
                                                  # np.nanmean(trial, axis, keepdims=True) per trial
                                                  # (compRoutines.py:22-57), then the usual trial average
                                                  # (computational_routine.py:1022-1032) if keeptrials=False

Both run as kernels of libspyhip (`spyhip_trial_mean_f32`: one thread per element walks the trials in order - the
reference's rounding sequence, bit for bit; `spyhip_axis_nanmean`: NumPy's summation order along the axis); there is no
CPU path.  `compute_method="sequential"` with `routine_classes` swaps in the oracle's NumPy functions for the tests.
"""
import numpy as np

from ..datatype import AnalogData, CrossSpectralData, SpectralData
from ..shared.errors import SPYTypeError, SPYValueError

__all__ = ["mean"]

_DIMPROPS = ("channel", "freq", "taper", "channel_i", "channel_j")


def _selected_trials(data):
    """[(trial array, absolute trial id)] honouring an in-place selection (trials in the given order, channels of
    AnalogData / SpectralData along their channel axis, latency windows along time)."""
    sel = data.selection
    trials = data.trials
    ids = list(range(len(trials))) if sel is None else list(sel.trial_ids)
    out = []
    for t in ids:
        x = trials[t]
        if sel is not None:
            a, b = sel.time[t]
            tax = data.dimord.index("time")
            if (a, b) != (0, x.shape[tax]):
                idx = [slice(None)] * x.ndim
                idx[tax] = slice(a, b)
                x = x[tuple(idx)]
            if "channel" in data.dimord and list(sel.channel) != list(range(x.shape[data.dimord.index("channel")])):
                x = np.take(x, sel.channel, axis=data.dimord.index("channel"))
        out.append((x, t))
    return out


def _new_like(data, arr, trialdefinition, dim=None, trials_sel=None):
    cls = data.__class__
    if cls is AnalogData:
        out = AnalogData(arr, samplerate=data.samplerate, trialdefinition=trialdefinition, dimord=data.dimord)
    else:
        out = cls(arr, samplerate=data.samplerate, trialdefinition=trialdefinition, dimord=data.dimord)
    for prop in _DIMPROPS:
        if not hasattr(data, prop) or getattr(data, prop) is None:
            continue
        val = np.asarray(getattr(data, prop))
        if prop == "channel" and data.selection is not None:
            val = val[list(data.selection.channel)]
        if dim is not None and dim in prop:
            # the averaged dimension: one entry labelled with the operation; a numerical freq axis is gone
            # (compRoutines.py:131-141 - `dim in prop`, so dim="channel" also relabels channel_i / channel_j)
            setattr(out, prop, None if dim == "freq" else np.array(["mean"]))
            continue
        setattr(out, prop, val)
    out.cfg = dict(getattr(data, "cfg", {}) or {})
    return out


def mean(spy_data, dim, keeptrials=True, select=None, compute_method=None, routine_classes=None, **kwargs):
    """Average of `spy_data` along the dimension `dim` (a label of its dimord) or over its trials (dim="trials").

    spy_data   : AnalogData, SpectralData or CrossSpectralData
    dim        : "trials" or one of spy_data.dimord
    keeptrials : False additionally averages the per-trial results over the trials (no effect for dim="trials")
    select     : in-place selection {"trials", "channel", "latency"}

    Returns a new object of the same class.  Trial averages need trials of identical shape
    (summary_stats.py:259-266)."""
    if not isinstance(spy_data, (AnalogData, SpectralData, CrossSpectralData)):
        raise SPYTypeError(spy_data, varname="spy_data", expected="Syncopy data object")
    if spy_data.data is None or spy_data.trialdefinition is None:
        raise SPYValueError("non-empty Syncopy data object", varname="spy_data", actual="empty object")
    if dim != "trials" and dim not in spy_data.dimord:
        raise SPYValueError(f"one of {spy_data.dimord} or 'trials'", varname="dim", actual=str(dim))
    had_selection = spy_data.selection
    if select is not None:
        spy_data.selectdata(select)
    try:
        trials = _selected_trials(spy_data)
        if len(trials) < 1:
            raise SPYValueError("at least 1 trial", varname="in_data", actual=f"got {len(trials)} trials")
        seldef = (spy_data.trialdefinition if spy_data.selection is None else spy_data.selection.trialdefinition)
        ops = _device_ops() if compute_method in (None, "hip") else routine_classes
        if dim == "trials":
            shape0 = trials[0][0].shape
            for x, _ in trials:
                if x.shape != shape0:
                    raise SPYValueError("all trials to have the same shape", varname="in_data",
                                        actual=f"found trials of different shape: {shape0} and {x.shape}")
            res = ops["trial_mean"]([x for x, _ in trials])
            trldef = np.array(seldef[0, :], dtype=float)[None, :]
            trldef[0, :2] = [0, res.shape[spy_data.dimord.index("time")]]
            return _new_like(spy_data, res, trldef, dim=None)
        axis = spy_data.dimord.index(dim)
        per_trial = [ops["axis_mean"](x, axis) for x, _ in trials]
        tax = spy_data.dimord.index("time")
        if not keeptrials:
            shape0 = per_trial[0].shape
            if any(r.shape != shape0 for r in per_trial):
                raise NotImplementedError("trial averaging needs trials of equal length")      # computational_routine.py:319-321
            res = ops["trial_mean"](per_trial)
            n = res.shape[tax]
            trldef = np.array([[0, 1, 0]], dtype=float) if dim == "time" else np.array([[0, n, seldef[0, 2]]], dtype=float)
            return _new_like(spy_data, res, trldef, dim=dim)
        res = np.concatenate(per_trial, axis=tax)
        if dim == "time":
            k = np.arange(len(per_trial), dtype=float)[:, None]
            trldef = np.hstack((k, k + 1, np.zeros((len(per_trial), 1))))
        else:
            trldef = np.array(seldef, dtype=float)
        return _new_like(spy_data, res, trldef, dim=dim)
    finally:
        spy_data.selection = had_selection


def _device_ops():
    import torch
    from .. import backend
    backend.require_gpu()

    def to_dev(x):
        return torch.from_numpy(np.ascontiguousarray(x)).cuda()

    def trial_mean(trials):
        stacked = torch.stack([to_dev(x) for x in trials])
        if stacked.dtype not in (torch.float32, torch.complex64):
            raise SPYTypeError(stacked.dtype, varname="data", expected="float32 or complex64 data")
        return backend.to_host(backend.trial_mean(stacked.contiguous()))

    def axis_mean(x, axis):
        d = to_dev(x)
        if d.dtype not in (torch.float32, torch.complex64):
            raise SPYTypeError(d.dtype, varname="data", expected="float32 or complex64 data")
        return backend.to_host(backend.axis_nanmean(d, axis))

    return {"trial_mean": trial_mean, "axis_mean": axis_mean}
