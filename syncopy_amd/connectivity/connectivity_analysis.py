"""`connectivityanalysis` metafunction: method = 'csd' | 'coh' | 'granger' | 'ppc' | 'corr'.

Two-stage pipeline of syncopy/connectivity/connectivity_analysis.py: single-trial
cross-spectra accumulated over trials (ST stage, :587-599), then one evaluation on the
trial average (AV stage, :677-679).  Parameter handling follows :286-473 and the
`cross_spectra` helper (:775-872).
"""
import numbers
import weakref

import numpy as np

from ..datatype import AnalogData, CrossSpectralData, SpectralData, selected_channels, selected_trialdefinition
from ..shared.const_def import connectivity_outputs, connectivityMethods
from ..shared.errors import SPYTypeError, SPYValueError, SPYWarning
from ..shared.kwarg_decorators import attached_selection, unwrap_cfg
from ..shared.input_processors import process_foi, process_padding, process_taper
from ..shared.tools import best_match
from .AV_compRoutines import NormalizeCrossCov, NormalizeCrossSpectra, pairwise_phase_consistency
from .ST_compRoutines import CrossCovariance, CrossSpectra, SpectralDyadicProduct


@unwrap_cfg
def connectivityanalysis(data, method="coh", keeptrials=False, output="abs", foi=None, foilim=None, pad="maxperlen",
                         polyremoval=0, tapsmofrq=None, nTaper=None, taper="hann", taper_opt=None, jackknife=False,
                         channelcmb=None, select=None, compute_method=None, routine_classes=None, precision="auto",
                         **kwargs):
    """Cross-spectral connectivity of AnalogData on MI355X (arguments as spy.connectivityanalysis,
    connectivity_analysis.py:51-67).
    `precision` (not a reference argument; as in `freqanalysis`): "reference" runs the taper product and the FFT of the
    single-trial spectra in float64 and rounds to complex64 where the reference does (mtmfft.py:96-127) - coherence, ppc
    and Granger are RATIOS of spectra, and where a channel's power is 40 dB or more below its peak the float32
    transform's absolute error (5e-7 of the rms bin) is no longer small against the bin itself; any transform length up
    to 2^20; ~2x the time of the transform stage at the lengths with a compile-time schedule (powers of two 256 ...
    16384, 200 ... 10000 decimal), 4-10x elsewhere.  "float32": the fast kernels whatever the data.  "auto" (default):
    the per-trial route (compute_method="sequential") transforms in float64 (its cost is invisible next to the copies);
    the batched route transforms the first 16 trials in float32, looks at the dynamic range of their spectra
    (CrossSpectra.needs_float64: coh / granger / ppc) and runs the analysis with float64 transforms when the float32
    error in the result would pass its floor (the benchmark's AR(2) data never does; cost of the look: 16 trials'
    transforms and one host synchronisation per call)."""
    if precision not in ("float32", "reference", "auto"):
        raise SPYValueError("'float32', 'reference' or 'auto'", varname="precision", actual=str(precision))
    if not isinstance(data, (AnalogData, SpectralData)) or data.data is None:
        raise SPYValueError("either AnalogData or SpectralData as input", "data", data.__class__.__name__)
    if method not in connectivityMethods:
        raise SPYValueError("one of " + ", ".join(connectivityMethods), varname="method", actual=method)
    if not isinstance(jackknife, bool):
        raise SPYTypeError(jackknife, "jackknife", "boolean")
    if jackknife and method not in ("coh", "granger"):
        SPYWarning(f"Jackknife is not available for method {method}")       # connectivity_analysis.py:310-317
        jackknife = False
    if polyremoval is not None:
        if not isinstance(polyremoval, numbers.Number) or polyremoval not in (0, 1):
            raise SPYValueError("0, 1 or None", varname="polyremoval", actual=polyremoval)
    classes = {"csd": CrossSpectra, "coh": NormalizeCrossSpectra, "dyadic": SpectralDyadicProduct,
               "ppc": pairwise_phase_consistency, "ccov": CrossCovariance, "ccov_norm": NormalizeCrossCov}
    try:
        from .AV_compRoutines import GrangerCausality
        classes["granger"] = GrangerCausality
    except ImportError:
        pass
    classes.update(routine_classes or {})
    with attached_selection(data, select):
        cmb = _parse_channelcmb(data, channelcmb)
        from ..specest import hip_spectral as hs

        def run():
            return _connectivity(data, classes, method, keeptrials, output, foi, foilim, pad, polyremoval, tapsmofrq,
                                 nTaper, taper, taper_opt, compute_method, jackknife, cmb)
        with hs.precision(precision):
            return run()


def _parse_channelcmb(data, channelcmb):
    """[senders, receivers] -> two lists of channel indices, validated as in connectivity_analysis.py:335-381."""
    if channelcmb is None:
        return None
    if not isinstance(data, SpectralData):
        raise SPYTypeError(data, "data", expected="SpectralData, `channelcmb` not supported for other data types, ")
    if not isinstance(channelcmb, list):
        raise SPYTypeError(channelcmb, "channelcmb", expected="list")
    if len(channelcmb) != 2:
        raise SPYValueError(legal="list with exactly two elements: [senders, receivers]", varname="channelcmb",
                            actual=f"length of {len(channelcmb)}")
    if selected_channels(data) is not None and list(selected_channels(data)) != list(range(len(data.channel))):
        raise SPYValueError("either channel selection or use channelcmb", "select/channelcmb", "both")
    senders, receivers = channelcmb
    for seq, name in ((senders, "channelcmb[senders,"), (receivers, "channelcmb[,receivers]")):
        if isinstance(seq, (str, bytes)) or not hasattr(seq, "__len__") or len(seq) == 0:
            raise SPYTypeError(seq, name, expected="sequence of channel names or indices")
    if isinstance(senders[0], (bool, np.bool_)) or not isinstance(senders[0], (str, int, np.integer)):
        raise SPYTypeError(senders[0], "channelcmb[senders,", "either `int` or `str`")
    by_name = isinstance(senders[0], str)
    names = [str(c) for c in data.channel]
    out = []
    for seq, name in ((senders, "channelcmb[senders,"), (receivers, "channelcmb[,receivers]")):
        idx = []
        for chan in seq:
            if isinstance(chan, str) != by_name or (not by_name and not isinstance(chan, (int, np.integer))):
                raise SPYTypeError(chan, name, expected="str" if by_name else "int")
            if (by_name and chan not in names) or (not by_name and not 0 <= chan < len(names)):
                raise SPYValueError("names or indices of existing channels", "channelcmb", chan)
            idx.append(names.index(chan) if by_name else int(chan))
        out.append(idx)
    return out


def _trial_average(x):
    """`spy.mean(dim="trials")`: sequential accumulation in the data dtype, then one division
    (statistics/summary_stats.py:321-400)."""
    acc = np.zeros(x.shape[1:], dtype=x.dtype)
    for t in range(x.shape[0]):
        acc += x[t]
    acc /= x.shape[0]
    return acc


def _as_single_trials(template, arr):
    """CrossSpectralData whose trials are the slices arr[t] (stacked along the time axis)."""
    T = arr.shape[0]
    obj = CrossSpectralData(dimord=template.dimord)
    obj.data = np.ascontiguousarray(arr)
    obj.samplerate = template.samplerate
    k = np.arange(T, dtype=float)[:, None]
    obj.trialdefinition = np.hstack((k, k + 1, np.zeros((T, 1))))
    obj.freq, obj.channel_i, obj.channel_j = template.freq, template.channel_i, template.channel_j
    return obj


def _post_select(out, send, rec):
    """out[..., senders, receivers] with the labels - coherence with `channelcmb` (connectivity_analysis.py:760-763)."""
    sel = CrossSpectralData(dimord=out.dimord)
    sel.data = np.ascontiguousarray(np.asarray(out.data)[..., send, :][..., rec])
    sel.samplerate, sel.trialdefinition, sel.freq = out.samplerate, out.trialdefinition, out.freq
    sel.channel_i, sel.channel_j = np.array(out.channel_i)[send], np.array(out.channel_j)[rec]
    sel.cfg = out.cfg
    return sel


def _pairwise_granger(data, classes, st, compute_method, log_dict, send, rec):
    """Granger with `channelcmb` (connectivity_analysis.py:681-733): one bivariate factorisation per
    (sender, receiver) pair on the 2 x 2 sub-block of the trial-averaged cross spectra; only the direction
    sender -> receiver is kept."""
    st_out = CrossSpectralData(dimord=CrossSpectra.dimord)
    st.initialize(data, st_out._stackingDim, chan_per_worker=None, keeptrials=False)
    st.compute(data, st_out, parallel=False, log_dict=log_dict, method=compute_method)
    S = np.asarray(st_out.data)
    res = np.empty((1, S.shape[1], len(send), len(rec)), dtype=np.float32)
    pair_out = None
    for i1, ch1 in enumerate(send):
        for i2, ch2 in enumerate(rec):
            pair = _as_single_trials(st_out, S[..., [ch1, ch2], :][..., [ch1, ch2]])
            pair.channel_i = pair.channel_j = np.array(st_out.channel_i)[[ch1, ch2]]
            pair.trialdefinition = np.array([[0, 1.0, 0]])
            av = classes["granger"](rtol=5e-6, nIter=100, cond_max=1e4)
            pair_out = CrossSpectralData(dimord=st_out.dimord)
            av.initialize(pair, pair_out._stackingDim, chan_per_worker=None, keeptrials=False)
            av.pre_check()
            av.compute(pair, pair_out, parallel=False, log_dict=log_dict, method=compute_method)
            res[0, :, i1, i2] = np.asarray(pair_out.data)[0, :, 0, 1]
    out = CrossSpectralData(dimord=st_out.dimord)
    out.data = res
    out.samplerate, out.freq = st_out.samplerate, st_out.freq
    out.channel_i, out.channel_j = np.array(data.channel)[send], np.array(data.channel)[rec]
    out.trialdefinition = np.array([[0, 1.0, 0]])
    out.cfg = dict(log_dict or {})
    return out


def _connectivity_from_spectra(data, classes, method, keeptrials, output, compute_method, jackknife, cmb=None):
    """SpectralData input (connectivity_analysis.py:475-538): the spectra exist already, the ST stage is the dyadic
    product; everything about tapers / padding / frequencies was decided in freqanalysis."""
    if not np.issubdtype(np.asarray(data.data).dtype, np.complexfloating):
        raise SPYValueError("complex valued spectra, set `output='fourier'` in spy.freqanalysis!", "data",
                            "real valued spectral data")
    if method == "granger" and data.data.shape[data.dimord.index("time")] != len(data.sampleinfo):
        raise NotImplementedError("Time resolved Granger causality from tf-spectra not available atm")
    log_dict = {"method": method, "output": output, "keeptrials": keeptrials}
    if cmb is not None and method in ("csd", "ppc"):
        # truly rectangular products (connectivity_analysis.py:503-529)
        st = classes["dyadic"](send_idx=cmb[0], send_N=len(cmb[0]), rec_idx=cmb[1], rec_N=len(cmb[1]))
    else:
        st = classes["dyadic"]()
    if cmb is not None and method == "granger":
        if "granger" not in classes:
            raise NotImplementedError("Wilson/Granger kernels are not part of this build")
        if jackknife:
            SPYWarning("jackknife estimates are not computed for pairwise (channelcmb) Granger causality")
        return _pairwise_granger(data, classes, st, compute_method, log_dict, *cmb)
    out = _run_stages(data, classes, st, method, keeptrials, output, compute_method, jackknife, log_dict)
    if cmb is not None and method == "coh":
        jack = {k: getattr(out, k, None) for k in ("jack_var", "jack_bias")}
        out = _post_select(out, *cmb)
        for k, v in jack.items():
            if v is not None:
                setattr(out, k, np.ascontiguousarray(np.asarray(v)[..., cmb[0], :][..., cmb[1]]))
    return out


def _cross_correlation(data, classes, keeptrials, foi, foilim, pad, polyremoval, compute_method):
    """method='corr' (connectivity_analysis.py:383-386,408-437): single-trial cross-covariances over the lags
    0 .. N/2, trial-averaged and normalised to cross-correlations - or, with keeptrials, normalised per trial."""
    if not isinstance(data, AnalogData):
        raise SPYValueError("AnalogData instance as input for method corr", "data", data.__class__.__name__)
    if pad != "maxperlen":
        raise SPYValueError("'maxperlen', no padding needed/allowed for cross-correlations", varname="pad",
                            actual=f"{pad}")
    if foi is not None or foilim is not None:
        SPYWarning("Parameter `foi` has no effect for method `corr`")
    log_dict = {"method": "corr", "keeptrials": keeptrials, "polyremoval": polyremoval, "pad": pad}
    st = classes["ccov"](samplerate=data.samplerate, polyremoval=polyremoval, timeAxis=data.dimord.index("time"),
                         norm=bool(keeptrials))
    st_out = CrossSpectralData(dimord=CrossCovariance.dimord)
    st.initialize(data, st_out._stackingDim, chan_per_worker=None, keeptrials=bool(keeptrials))
    st.compute(data, st_out, parallel=False, log_dict=log_dict, method=compute_method)
    if keeptrials:
        return st_out
    av = classes["ccov_norm"]()
    out = CrossSpectralData(dimord=st_out.dimord)
    av.initialize(st_out, out._stackingDim, chan_per_worker=None, keeptrials=False)
    av.pre_check()
    av.compute(st_out, out, parallel=False, log_dict=log_dict, method=compute_method)
    return out


def _connectivity(data, classes, method, keeptrials, output, foi, foilim, pad, polyremoval, tapsmofrq, nTaper, taper,
                  taper_opt, compute_method, jackknife=False, cmb=None):
    fs = data.samplerate
    timeAxis = data.dimord.index("time")
    trl = selected_trialdefinition(data)
    lenTrials = np.diff(trl[:, :2]).squeeze(axis=1)
    nTrials = lenTrials.size
    if nTrials == 1 and method != "corr":
        raise SPYValueError("multi-trial input data, spectral connectivity measures critically depend on trial "
                            "averaging!", "data", "only one trial")
    if keeptrials is not False and method in ("coh", "ppc", "granger"):
        raise SPYValueError(f"False, trial averaging needed for method {method}!", varname="keeptrials",
                            actual=keeptrials)
    if method == "corr":
        return _cross_correlation(data, classes, keeptrials, foi, foilim, pad, polyremoval, compute_method)
    if isinstance(data, SpectralData):
        return _connectivity_from_spectra(data, classes, method, keeptrials, output, compute_method, jackknife, cmb)
    nSamples = process_padding(pad, lenTrials, fs)
    foi, foilim = process_foi(foi, foilim, fs)
    if method == "granger":
        if foi is not None or foilim is not None:
            raise SPYValueError("no foi specification for Granger analysis", "foi/foilim",
                                "foi or foilim specification")
        chans = selected_channels(data)
        nChannels = len(data.channel) if chans is None else len(chans)
        if nChannels / nTrials > 0.1:
            SPYWarning("Multi-channel Granger analysis can be numerically unstable, it is recommended to have at "
                       "least 10 times the number of trials compared to the number of channels.")
    freqs = np.fft.rfftfreq(nSamples, 1 / fs)
    if foi is not None:
        foi, _ = best_match(freqs, foi, squash_duplicates=True)
    elif foilim is not None:
        foi, _ = best_match(freqs, foilim, span=True, squash_duplicates=True)
    else:
        foi = freqs
    taper, taper_opt = process_taper(taper, taper_opt, tapsmofrq, nTaper, keeptapers=False, foimax=foi.max(),
                                     samplerate=fs, nSamples=lenTrials.mean(), output="pow")
    log_dict = {"method": method, "output": output, "keeptrials": keeptrials, "polyremoval": polyremoval,
                "pad": pad, "foi": foi, "taper": taper, "taper_opt": taper_opt}

    st = classes["csd"](samplerate=fs, nSamples=nSamples, taper=taper, taper_opt=taper_opt,
                        demean_taper=(method == "granger"), polyremoval=polyremoval, timeAxis=timeAxis, foi=foi)
    return _run_stages(data, classes, st, method, keeptrials, output, compute_method, jackknife, log_dict)


def _jackknife_on_device(data, st, av, st_out, log_dict):
    """jackknife=True with the kernels: CrossSpectra.jackknife_hip streams the leave-one-out replicates through the
    AV stage on the device; only the direct estimate, bias and variance come back."""
    from .. import backend
    st.initialize(data, st_out._stackingDim, chan_per_worker=None, keeptrials=False)
    av.metadata = []
    S, direct, bias, var = st.jackknife_hip(data, av.evaluate_device, fused=getattr(av, "jackknife_accumulate", None))
    st_out._dev = S.reshape(st.outputShape)
    st_ref = weakref.ref(st_out)                            # (no cycle through the object's own thunk)
    st_out.set_pending(lambda: backend.to_host(st_ref()._dev), st.outputShape, np.complex64)
    st.process_metadata(data, st_out)
    out = CrossSpectralData(dimord=st_out.dimord)
    av.initialize(st_out, out._stackingDim, chan_per_worker=None, keeptrials=False)
    out._dev = direct.unsqueeze(0)
    out.data = backend.to_host(out._dev)
    av.process_metadata(st_out, out)
    out.cfg = dict(log_dict or {})
    out.jack_bias = backend.to_host(bias)[None]
    out.jack_var = backend.to_host(var)[None]
    return out


def _ppc(data, classes, st, compute_method, log_dict):
    """Pairwise phase consistency (connectivity_analysis.py:551-562,590,624-663): all trial pairs of the single-trial
    cross spectra.  With the kernels nothing is kept (K7 streams the trials); the per-trial engine path keeps the
    single-trial cross spectra as the reference does and hands them to classes["ppc"]."""
    from .. import backend
    out = CrossSpectralData(dimord=CrossSpectra.dimord)
    if compute_method in (None, "hip") and hasattr(st, "ppc_hip"):
        st.initialize(data, out._stackingDim, chan_per_worker=None, keeptrials=False)
        res = st.ppc_hip(data)                       # (F, Ci, Cj) from raw trials, (nTime, F, Ci, Cj) from spectra
        out._dev = res if res.dim() == 4 else res.unsqueeze(0)
        out.data = backend.to_host(out._dev)
        st.process_metadata(data, out)
    else:
        st_out = CrossSpectralData(dimord=CrossSpectra.dimord)
        st.initialize(data, st_out._stackingDim, chan_per_worker=None, keeptrials=True)
        st.compute(data, st_out, parallel=False, log_dict=log_dict, method=compute_method)
        single = np.asarray(st_out.data)
        lens = {int(b - a) for a, b in st_out.sampleinfo}
        if len(lens) != 1:
            raise SPYValueError("trials of equal length", varname="data",
                                actual=f"time-resolved spectra with {sorted(lens)} samples per trial")
        L = lens.pop()
        per_time = single.reshape((-1, L) + single.shape[1:])            # (trials, time, F, C_i, C_j)
        out.data = np.ascontiguousarray(np.concatenate([classes["ppc"](per_time[:, ti]) for ti in range(L)], axis=0),
                                        dtype=np.float32)
        out.samplerate, out.freq = st_out.samplerate, st_out.freq
        out.channel_i, out.channel_j = st_out.channel_i, st_out.channel_j
        out.trialdefinition = st_out.trialdefinition[:1].copy() if L > 1 else np.array([[0, 1.0, 0]])
    out.cfg = dict(log_dict or {})
    return out


def _run_stages(data, classes, st, method, keeptrials, output, compute_method, jackknife, log_dict):
    """ST stage (single-trial cross spectra, trial-averaged unless kept) -> AV stage, plus the jackknife.
    Coherence outputs that are the imaginary part or the phase run K4 with directly summed imaginary parts
    (backend.csd_phase_exact): the default 3-multiplication kernels subtract three rounded row sums there."""
    from contextlib import ExitStack
    from ..specest import hip_spectral as hs
    batched = compute_method in (None, "hip")
    with ExitStack() as stack:
        # precision="auto": float64 transforms when the OUTPUT isolates a part of the complex coherency (its imaginary or real
        # part, its phase: 1.5e-7 of the modulus is not small against a part that nearly vanishes - as freqanalysis decides for
        # its own outputs), or when the DATA ask for them (CrossSpectra.needs_float64)
        if (batched and hs.requested_precision() is None and method in ("coh", "granger", "ppc") and isinstance(data, AnalogData)
                and ((method == "coh" and output in ("imag", "angle", "real"))
                     or (hasattr(st, "needs_float64") and st.needs_float64(data, method)))):
            stack.enter_context(hs.soft_reference())
        if method == "coh" and output in ("imag", "angle") and batched:
            from .. import backend
            stack.enter_context(backend.csd_phase_exact(True))
        return _run_stages_impl(data, classes, st, method, keeptrials, output, compute_method, jackknife, log_dict)


def _run_stages_impl(data, classes, st, method, keeptrials, output, compute_method, jackknife, log_dict):
    if method == "ppc":
        return _ppc(data, classes, st, compute_method, log_dict)
    if method == "coh":
        if output not in connectivity_outputs:
            raise SPYValueError(f"one of {sorted(connectivity_outputs)}", varname="output", actual=output)
        av = classes["coh"](output=output)
    elif method == "granger":
        if "granger" not in classes:
            raise NotImplementedError("Wilson/Granger kernels are not part of this build")
        av = classes["granger"](rtol=5e-6, nIter=100, cond_max=1e4)
    else:
        av = None

    st_out = CrossSpectralData(dimord=CrossSpectra.dimord)
    if (jackknife and av is not None and compute_method in (None, "hip") and hasattr(st, "jackknife_hip")
            and hasattr(av, "evaluate_device")):
        return _jackknife_on_device(data, st, av, st_out, log_dict)
    # single trials are needed for the jackknife (connectivity_analysis.py:589-590)
    st.initialize(data, st_out._stackingDim, chan_per_worker=None, keeptrials=bool(keeptrials) or jackknife)
    st.compute(data, st_out, parallel=False, log_dict=log_dict, method=compute_method)
    if av is None:
        return st_out
    replicates = None
    if jackknife:
        # leave-one-out trial averages (statistics/jackknifing.py:14-108): (T * mean - trial) / (T - 1)
        S = np.asarray(st_out.data)
        T = S.shape[0]
        mean = _trial_average(S)
        rep = np.empty_like(S)
        for t in range(T):
            loo = T * mean - S[t]
            loo /= T - 1
            rep[t] = loo
        replicates = _as_single_trials(st_out, rep)
        st_out = _as_single_trials(st_out, mean[None])
        st_out.trialdefinition = np.array([[0, 1.0, 0]])
    out = CrossSpectralData(dimord=st_out.dimord)
    av.initialize(st_out, out._stackingDim, chan_per_worker=None, keeptrials=False)
    av.pre_check()
    av.compute(st_out, out, parallel=False, log_dict=log_dict, method=compute_method)
    if jackknife:
        # the same AV routine on every replicate, then bias and variance (jackknifing.py:111-184)
        jack_rep = CrossSpectralData(dimord=st_out.dimord)
        av_rep = av.__class__(**av.cfg)
        av_rep.initialize(replicates, jack_rep._stackingDim, chan_per_worker=None, keeptrials=True)
        av_rep.compute(replicates, jack_rep, parallel=False, log_dict=log_dict, method=compute_method)
        R = np.asarray(jack_rep.data)
        T = R.shape[0]
        direct = np.asarray(out.data)
        jack_avg = _trial_average(R)[None]
        prefac = (T - 1) + 0j if np.issubdtype(direct.dtype, np.complexfloating) else (T - 1)
        out.jack_bias = (prefac * (jack_avg - direct)).astype(direct.dtype)
        var = np.zeros(direct.shape, dtype=np.float32)
        for t in range(T):
            var += (np.abs(jack_avg - R[t])) ** 2
        var *= T - 1
        out.jack_var = var
    return out
