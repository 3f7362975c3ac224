"""Compute functions / compute classes of spectral estimation on MI355X.

Signatures, dry-run behaviour and return conventions follow
syncopy/specest/compRoutines.py (mtmfft_cF:60, mtmconvol_cF:245, wavelet_cF:483;
MultiTaperFFT:194, MultiTaperFFTConvol:417, WaveletTransform:598) so the classes can
be bound under spy.freqanalysis unchanged; all arithmetic runs in libspyhip.
"""
import numbers
from hashlib import blake2b

import numpy as np
import torch

from .. import parallel
from ..datatype import device_rows, selected_channels, trial_rows
from ..shared.computational_routine import ComputationalRoutine, propagate_properties
from ..shared.const_def import spectralDTypes
from ..shared.tools import best_match
from . import hip_spectral as hs


def _freqs_hash(freqs):
    return np.array(blake2b(freqs).hexdigest().encode("utf-8"))


def _as_device_trial(trl_dat, timeAxis):
    """One host trial -> (time x channel) float32 matrix in HBM."""
    backend = hs.backend
    backend.require_gpu()
    dat = trl_dat.T if timeAxis != 0 else trl_dat
    return torch.from_numpy(np.ascontiguousarray(dat, dtype=np.float32)).cuda()


# --------------------------------------------------------------------------- mtmfft
def mtmfft_cF(trl_dat, foi=None, timeAxis=0, keeptapers=True, polyremoval=None, output="pow", noCompute=False,
              chunkShape=None, method_kwargs=None):
    """(Multi-)tapered Fourier transform of one trial; returns (1, nTaper|1, nFreq, nChannel)."""
    dat = trl_dat.T if timeAxis != 0 else trl_dat
    nSamples = dat.shape[0] if method_kwargs["nSamples"] is None else method_kwargs["nSamples"]
    nChannels = dat.shape[1]
    freqs = np.fft.rfftfreq(nSamples, 1 / method_kwargs["samplerate"])
    _, freq_idx = best_match(freqs, foi, squash_duplicates=True)
    nTaper = method_kwargs["taper_opt"].get("Kmax", 1)
    outShape = (1, max(1, nTaper * keeptapers), freq_idx.size, nChannels)
    if noCompute:
        return outShape, spectralDTypes[output]

    dev = _as_device_trial(trl_dat, timeAxis)
    with hs.per_trial_route():
        res = hs.run_mtmfft(dev, [(0, dev.shape[0])], None, nSamples, method_kwargs["taper"], method_kwargs["taper_opt"],
                            method_kwargs.get("demean_taper", False), method_kwargs.get("ft_compat", False), polyremoval,
                            freq_idx, output, keeptapers)[0]
    spec = hs.backend.to_host(res)[np.newaxis]
    return spec, {"freqs_hash": _freqs_hash(freqs)}


class MultiTaperFFT(ComputationalRoutine):
    computeFunction = staticmethod(mtmfft_cF)
    valid_kws = ["samplerate", "nSamples", "taper", "taper_opt", "demean_taper", "ft_compat", "foi", "timeAxis",
                 "keeptapers", "polyremoval", "output", "method_kwargs", "tapsmofrq", "nTaper", "pad"]

    def compute_hip(self, data, out):
        """All trials from the in-HBM queue; trial mean (keeptrials=False) in the reference's order."""
        mk, cfg = self.cfg["method_kwargs"], self.cfg
        dev = data.device_data()
        rows, chans = device_rows(data), selected_channels(data)
        lengths = [b - a for a, b in rows]
        freqs = np.fft.rfftfreq(mk["nSamples"] if mk["nSamples"] is not None else lengths[0], 1 / mk["samplerate"])
        _, freq_idx = best_match(freqs, cfg["foi"], squash_duplicates=True)
        rows = [rows[k] for k in self.my_trials()]          # this rank's trial shard
        res = hs.run_mtmfft(dev, rows, chans, mk["nSamples"], mk["taper"], mk["taper_opt"],
                            mk.get("demean_taper", False), mk.get("ft_compat", False), cfg["polyremoval"], freq_idx,
                            cfg["output"], cfg["keeptapers"])
        self.metadata = [{"freqs_hash": _freqs_hash(freqs)}] * self.numTrials
        _store_trials(self, out, res, stack=True)

    def process_metadata(self, data, out):
        hashes = {bytes(m["freqs_hash"]) for m in self.metadata if m}
        if len(hashes) > 1:
            raise ValueError("frequency axes of the trials differ")
        propagate_properties(data, out, self.keeptrials)
        taper_kw = self.cfg["method_kwargs"]["taper"]
        if taper_kw is None:
            out.taper = np.array(["None"])
        elif taper_kw == "dpss":
            nTaper = self.outputShape[out.dimord.index("taper")]
            out.taper = np.array([taper_kw + str(i) for i in range(nTaper)])
        else:
            out.taper = np.array([taper_kw])
        out.freq = self.cfg["foi"]


# --------------------------------------------------------------------------- mtmconvol
def _frame_ids(n_frames, postselect):
    return np.arange(n_frames)[postselect]


def _stft_geometry(nsamp, nperseg, noverlap, toi_is_array):
    """(nTime, boundary) of mtmconvol for a trial of nsamp samples (mtmconvol.py:120-126)."""
    step = nperseg - noverlap
    nTime = int(np.ceil(nsamp / step))
    if toi_is_array:
        return nTime - nperseg, False
    return nTime, True


def _first_window_bins(n0, fs, foi, n_chan):
    """Bin indices matched on the first window of a trial (compRoutines.py:402-404).  The reference writes the selected bins
    into a block of foi.size frequencies: another count does not fit (NumPy's broadcast error), a single bin fills the block."""
    _, fi = best_match(np.fft.rfftfreq(n0, 1 / fs), foi, squash_duplicates=True)
    if fi.size != foi.size:
        if fi.size != 1:
            raise ValueError(f"could not broadcast input array from shape (1,{fi.size},{n_chan}) into shape "
                             f"(1,{foi.size},{n_chan})")
        fi = np.repeat(fi, foi.size)
    return fi


def _mtmconvol_device(dev, row0, nsamp, soi, postselect, equidistant, toi, foi, keeptapers, polyremoval, output,
                      method_kwargs, chans):
    """Time-frequency spectrum of one trial (rows [row0, row0+nsamp) of `dev`) -> device tensor."""
    nperseg, noverlap = method_kwargs["nperseg"], method_kwargs["noverlap"]
    taper, taper_opt = method_kwargs["taper"], method_kwargs["taper_opt"]
    fs = method_kwargs["samplerate"]
    if equidistant:
        s0, s1, _ = soi.indices(nsamp)
        nTime, boundary = _stft_geometry(s1 - s0, nperseg, noverlap, isinstance(toi, np.ndarray))
        freqs = np.fft.rfftfreq(nperseg, 1 / fs)
        _, fidx = best_match(freqs, foi, squash_duplicates=True)
        frames = _frame_ids(max(nTime, 0), postselect)
        return hs.run_stft(dev, row0, s0, s1, frames, nperseg, nperseg - noverlap, boundary, chans, taper,
                           taper_opt, polyremoval, fidx, output, keeptapers)
    # one window per soi entry: plain (un-detrended, un-padded) mtmfft per window (compRoutines.py:392-408)
    rows = []
    for sl in soi:
        a, b, _ = sl.indices(nsamp)
        rows.append((row0 + a, row0 + max(a, b)))
    # compRoutines.py:403-408: the bin indices matched on the FIRST window serve every window of the trial
    if not rows:
        raise IndexError("list index out of range")
    lens, seen_len, fi = {b - a for a, b in rows}, set(), None
    for a, b in rows:                 # (in window order: the first offending window speaks)
        n = b - a
        if n not in seen_len:
            seen_len.add(n)
            hs.taper_table(taper, n, n, taper_opt)                                      # the window function's own checks
        if fi is None:
            fi = _first_window_bins(n, fs, foi, dev.shape[1] if chans is None else len(chans))
        if fi.size and int(fi.max()) >= n // 2 + 1:
            raise IndexError(f"index {int(fi.max())} is out of bounds for axis 1 with size {n // 2 + 1}")
    res = [None] * len(rows)
    for n in lens:
        which = [i for i, (a, b) in enumerate(rows) if b - a == n]
        part = hs.run_mtmfft(dev, [rows[i] for i in which], chans, None, taper, taper_opt, False, False, None,
                             fi, output, keeptapers)
        for i, r in zip(which, part):
            res[i] = r
    return torch.stack(res, dim=0)


def mtmconvol_cF(trl_dat, soi, postselect, equidistant=True, toi=None, foi=None, nTaper=1, tapsmofrq=None,
                 timeAxis=0, keeptapers=True, polyremoval=0, output="pow", noCompute=False, chunkShape=None,
                 method_kwargs=None):
    """Sliding-window (multi-)tapered FFT of one trial; returns (nTime, nTaper|1, nFreq, nChannel)."""
    dat = trl_dat.T if timeAxis != 0 else trl_dat
    nChannels = dat.shape[1]
    if isinstance(toi, np.ndarray):
        nTime = toi.size
    else:
        nTime = int(np.ceil(dat.shape[0] / (method_kwargs["nperseg"] - method_kwargs["noverlap"])))
    taper_opt = method_kwargs["taper_opt"]
    if taper_opt:
        nTaper = taper_opt.get("Kmax", 1)
    outShape = (nTime, max(1, nTaper * keeptapers), foi.size, nChannels)
    if noCompute:
        return outShape, spectralDTypes[output]
    dev = _as_device_trial(trl_dat, timeAxis)
    with hs.per_trial_route():
        res = _mtmconvol_device(dev, 0, dev.shape[0], soi, postselect, equidistant, toi, foi, keeptapers, polyremoval,
                                output, method_kwargs, None)
    return hs.backend.to_host(res)


class MultiTaperFFTConvol(ComputationalRoutine):
    computeFunction = staticmethod(mtmconvol_cF)
    valid_kws = ["soi", "postselect", "equidistant", "toi", "foi", "nTaper", "tapsmofrq", "timeAxis", "keeptapers",
                 "polyremoval", "output", "method_kwargs", "samplerate", "nperseg", "noverlap", "taper", "taper_opt",
                 "t_ftimwin", "pad"]

    def compute_hip(self, data, out):
        cfg = self.cfg
        dev = data.device_data()
        rows, chans = device_rows(data), selected_channels(data)
        mine = list(self.my_trials())
        if cfg["equidistant"] and mine:
            # every trial's frames in one launch (the frames of 200 trials x 64 windows are 12800 segments of one plan;
            # trial by trial the call is 12 x slower than its kernels): stft.py:16-159 / mtmconvol.py:120-150
            mk = cfg["method_kwargs"]
            nperseg, noverlap = mk["nperseg"], mk["noverlap"]
            _, fidx = best_match(np.fft.rfftfreq(nperseg, 1 / mk["samplerate"]), cfg["foi"], squash_duplicates=True)
            boundary, trials = True, []
            for k in mine:
                a, b = rows[k]
                soi, postselect = self._argv(k)
                s0, s1, _ = soi.indices(b - a)
                nTime, boundary = _stft_geometry(s1 - s0, nperseg, noverlap, isinstance(cfg["toi"], np.ndarray))
                trials.append((a, s0, s1, _frame_ids(max(nTime, 0), postselect)))
            parts = hs.run_stft_trials(dev, trials, nperseg, nperseg - noverlap, boundary, chans, mk["taper"], mk["taper_opt"],
                                       cfg["polyremoval"], fidx, cfg["output"], cfg["keeptapers"])
        elif mine:
            # one window per soi entry (compRoutines.py:392-408: plain, un-detrended, un-padded mtmfft per window): the windows
            # of ALL trials by length, one launch per length
            mk = cfg["method_kwargs"]
            wins, owner = [], []
            for i, k in enumerate(mine):
                a, b = rows[k]
                soi, _ = self._argv(k)
                for sl in soi:
                    s0, s1, _ = sl.indices(b - a)
                    wins.append((a + s0, a + max(s0, s1)))
                    owner.append(i)
            # compRoutines.py:403-408: the bin indices matched on the FIRST window of a trial serve all of its windows (windows
            # cut short at a trial edge then read the same indices of a coarser frequency axis - or fall off its end)
            fs, first, groups, seen_len = mk["samplerate"], {}, {}, set()
            for i, ((w0, w1), o) in enumerate(zip(wins, owner)):      # (in window order: the first offending window speaks)
                if w1 - w0 not in seen_len:
                    seen_len.add(w1 - w0)
                    hs.taper_table(mk["taper"], w1 - w0, w1 - w0, mk["taper_opt"])      # the window function's own checks
                if o not in first:
                    first[o] = _first_window_bins(w1 - w0, fs, cfg["foi"], dev.shape[1] if chans is None else len(chans))
                fi = first[o]
                nf = (w1 - w0) // 2 + 1
                if fi.size and int(fi.max()) >= nf:
                    raise IndexError(f"index {int(fi.max())} is out of bounds for axis 1 with size {nf}")
                groups.setdefault((w1 - w0, fi.tobytes()), (fi, []))[1].append(i)
            res = [None] * len(wins)
            for (n, _), (fi, which) in groups.items():
                part = hs.run_mtmfft(dev, [wins[i] for i in which], chans, None, mk["taper"], mk["taper_opt"], False, False, None,
                                     fi, cfg["output"], cfg["keeptapers"])
                for i, r in zip(which, part):
                    res[i] = r
            per_trial = [[] for _ in mine]
            for r, o in zip(res, owner):
                per_trial[o].append(r)
            parts = [torch.stack(p, dim=0) for p in per_trial]
        else:
            parts = []
        if getattr(self, "reduce_time", False) and self.keeptrials and parts and not parallel.collective_active():
            # method="welch": the time mean of every trial (freqanalysis.py:1054-1056) taken on the device in NumPy's
            # order (spyhip_axis_nanmean) - the windowed spectra themselves reach the host only if somebody reads them
            self.time_means = hs.backend.to_host(torch.cat([hs.backend.axis_nanmean(p.contiguous(), 0) for p in parts], dim=0))
            out.set_pending(lambda: hs.backend.to_host(torch.cat(parts, dim=0)).reshape(self.outputShape), self.outputShape,
                            self.dtype)
            return
        _store_trials(self, out, parts)

    def process_metadata(self, data, out):
        propagate_properties(data, out, self.keeptrials, time_axis=True)
        trl, fs = _make_trialdef(self.cfg, out.trialdefinition.copy(), data.samplerate)
        out.trialdefinition, out.samplerate = trl, fs
        nTaper = self.outputShape[out.dimord.index("taper")]
        out.taper = np.array([str(self.cfg["method_kwargs"]["taper"])] * nTaper)
        out.freq = self.cfg["foi"]


# --------------------------------------------------------------------------- wavelet
def _postselect_map(nsig, postselect):
    """Sample -> output-slot map of a post-selection (slice or index list).  Returns (tpos or None,
    n_unique, gather) where `gather` re-creates duplicates / order of an index list (or None)."""
    if isinstance(postselect, slice):
        idx = np.arange(nsig)[postselect]
    else:
        idx = np.asarray(postselect, dtype=np.int64)
        idx = np.where(idx < 0, idx + nsig, idx)
    if idx.size == nsig and np.array_equal(idx, np.arange(nsig)):
        return None, nsig, None
    uniq, inverse = np.unique(idx, return_inverse=True)
    tpos = np.full(nsig, -1, dtype=np.int32)
    tpos[uniq] = np.arange(uniq.size, dtype=np.int32)
    gather = None if np.array_equal(inverse, np.arange(idx.size)) else inverse
    return tpos, int(uniq.size), gather


_cwt_plans = {}


def _cwt_reference():
    """Wavelet transforms run in float64 only when the call says precision="reference".  The reference's own transform is
    scipy.signal.fftconvolve of the float32 trial - whose forward FFT scipy computes in SINGLE precision (error 1.5e-7 of
    the largest coefficient): float64 convolutions are closer to the exact result, not to the reference, and cost ~50x -
    no route takes them by default."""
    return hs.requested_precision() == "reference"


def _cwt_precision(plan):
    if _cwt_reference() and not plan.set_precision(True):
        raise hs.PrecisionUnavailable("a convolution length up to 2^22 for float64 wavelet transforms", varname="precision",
                                      actual=f"nsig = {plan.nsig}")


def _wavelet_params(method_kwargs):
    """(w0, family, order) of the transform.  Two spellings of `method_kwargs`: this package's front end passes plain
    numbers ("w0", "family", "order"); the reference's freqanalysis passes the wavelet function OBJECT it built
    (freqanalysis.py:893-897: {"samplerate", "scales", "wavelet"}; classes Morlet(w0) / Paul(m) / DOG(m) and Ricker, Marr,
    Mexican_hat = DOG(2): specest/wavelets/wavelets.py:13-363) - read by class name and attribute, so that wavelet_cF is a
    drop-in under the reference's own WaveletTransform."""
    wav = method_kwargs.get("wavelet")
    if wav is None or isinstance(wav, str):
        return float(method_kwargs.get("w0", 6.0)), method_kwargs.get("family"), method_kwargs.get("order")
    names = [c.__name__ for c in type(wav).__mro__]
    if "Paul" in names:
        return 6.0, "Paul", int(wav.m)
    if "DOG" in names:
        return 6.0, "DOG", int(wav.m)
    if "Morlet" in names:
        return float(wav.w0), None, None
    raise ValueError("unknown wavelet function %r" % type(wav).__name__)


def _wavelet_device(dev, rows, pre, post, chans, polyremoval, output, method_kwargs, sum_trials=False):
    """Wavelet spectra of trials `rows` (absolute [start, stop)) with per-trial pre/post-selections.
    Returns a list of (nTime, 1, nScales, C) device tensors - or, with `sum_trials`, their sum as ONE such
    tensor accumulated on the fly (None when the trials do not share one plan: the caller then sums the list)."""
    device = dev.device
    nchan = dev.shape[1] if chans is None else len(chans)
    ci = None if chans is None else torch.tensor(np.asarray(chans), dtype=torch.int32, device=device)
    scales = np.asarray(method_kwargs["scales"], dtype=np.float64)
    dt = 1.0 / method_kwargs["samplerate"]
    w0, family, order = _wavelet_params(method_kwargs)
    results = [None] * len(rows)
    groups = {}
    for k, ((a, b), ps, qs) in enumerate(zip(rows, pre, post)):
        s0, s1, _ = ps.indices(b - a)
        nsig = max(s1 - s0, 0)
        tpos, nuniq, gather = _postselect_map(nsig, qs)
        key = (nsig, None if tpos is None else tpos.tobytes(), None if gather is None else gather.tobytes())
        groups.setdefault(key, (nsig, tpos, nuniq, gather, []))[4].append((k, a + s0, a, b))
    if sum_trials and (len(groups) != 1 or next(iter(groups.values()))[3] is not None):
        return None
    for nsig, tpos, nuniq, gather, members in groups.values():
        pkey = (nsig, nchan, scales.tobytes(), dt, w0, polyremoval, output, None if tpos is None else tpos.tobytes(),
                str(device), _cwt_reference(), family, order)
        plan = hs._cache_hit(_cwt_plans, pkey)
        if plan is None:
            plan = hs.backend.CWTPlan(nsig, nchan, scales, dt, w0, polyremoval, output, tpos, nuniq, device=device,
                                      family=family, order=order)
            _cwt_precision(plan)
            hs._bounded_put(_cwt_plans, pkey, plan)
        starts = torch.tensor([m[1] for m in members], dtype=torch.int64, device=device)
        lo = torch.tensor([m[2] for m in members], dtype=torch.int64, device=device)
        hi = torch.tensor([m[3] for m in members], dtype=torch.int64, device=device)
        if sum_trials:
            total = torch.zeros(plan.out_shape(1), dtype=plan.out_dtype, device=device)
            plan.execute(dev, starts, lo, hi, chan_idx=ci, out=total, accumulate=2)
            return total[0].unsqueeze(1)
        out = plan.execute(dev, starts, lo, hi, chan_idx=ci)
        for i, m in enumerate(members):
            r = out[i]
            if gather is not None:
                r = r.index_select(0, torch.from_numpy(gather).to(device))
            results[m[0]] = r.unsqueeze(1)
    return results


def wavelet_cF(trl_dat, preselect, postselect, toi=None, timeAxis=0, polyremoval=0, output="pow", noCompute=False,
               chunkShape=None, method_kwargs=None):
    """Morlet wavelet transform of one trial; returns (nTime, 1, nScales, nChannel)."""
    dat = trl_dat.T if timeAxis != 0 else trl_dat
    nChannels = dat.shape[1]
    nTime = toi.size if isinstance(toi, np.ndarray) else dat.shape[0]
    nScales = method_kwargs["scales"].size
    outShape = (nTime, 1, nScales, nChannels)
    if noCompute:
        return outShape, spectralDTypes[output]
    dev = _as_device_trial(trl_dat, timeAxis)
    res = _wavelet_device(dev, [(0, dev.shape[0])], [preselect], [postselect], None, polyremoval, output,
                          method_kwargs)[0]
    return hs.backend.to_host(res)


class WaveletTransform(ComputationalRoutine):
    computeFunction = staticmethod(wavelet_cF)
    valid_kws = ["preselect", "postselect", "toi", "timeAxis", "polyremoval", "output", "method_kwargs", "samplerate",
                 "scales", "wavelet", "width", "foi"]

    def compute_hip(self, data, out):
        cfg = self.cfg
        dev = data.device_data()
        rows, chans = device_rows(data), selected_channels(data)
        mine = list(self.my_trials())
        pre = [self._argv(k)[0] for k in mine]
        post = [self._argv(k)[1] for k in mine]
        if not self.keeptrials and mine:
            # trial average accumulated on the device while the trials are transformed (no per-trial outputs)
            total = _wavelet_device(dev, [rows[k] for k in mine], pre, post, chans, cfg["polyremoval"], cfg["output"],
                                    cfg["method_kwargs"], sum_trials=True)
            if total is not None:
                if tuple(total.shape) != tuple(self.targetShapes[mine[0]]):
                    raise ValueError(f"result shape {tuple(total.shape)} != dry-run shape {self.targetShapes[mine[0]]}")
                total = total.contiguous()
                parallel.allreduce_sum_(total)
                out.data = hs.backend.to_host((total / self.numTrials)).reshape(self.outputShape)
                return
        parts = _wavelet_device(dev, [rows[k] for k in mine], pre, post, chans, cfg["polyremoval"], cfg["output"],
                                cfg["method_kwargs"])
        _store_trials(self, out, parts)

    def process_metadata(self, data, out):
        propagate_properties(data, out, self.keeptrials, time_axis=True)
        out.trialdefinition, out.samplerate = _make_trialdef(self.cfg, out.trialdefinition.copy(), data.samplerate)
        out.taper = np.array(["None"])
        out.freq = getattr(self, "_foi", None)


def _superlet_device(dev, rows, pre, post, chans, polyremoval, output, method_kwargs):
    """Superlet spectra of trials `rows` with per-trial pre/post-selections: one CWT plan per order of the set
    (MorletSL kernels, complex output), each folded into the geometric mean by spyhip_slt_combine, converted at the
    end.  Returns a list of (nTime, 1, nScales, C) device tensors."""
    from .wavelet_tools import superlet_steps
    device = dev.device
    nchan = dev.shape[1] if chans is None else len(chans)
    ci = None if chans is None else torch.tensor(np.asarray(chans), dtype=torch.int32, device=device)
    scales = np.asarray(method_kwargs["scales"], dtype=np.float64)
    dt = 1.0 / method_kwargs["samplerate"]
    steps = superlet_steps(scales, method_kwargs["order_max"], method_kwargs.get("order_min", 1),
                           method_kwargs.get("c_1", 3), method_kwargs.get("adaptive", False))
    results = [None] * len(rows)
    groups = {}
    for k, ((a, b), ps, qs) in enumerate(zip(rows, pre, post)):
        s0, s1, _ = ps.indices(b - a)
        nsig = max(s1 - s0, 0)
        tpos, nuniq, gather = _postselect_map(nsig, qs)
        key = (nsig, None if tpos is None else tpos.tobytes(), None if gather is None else gather.tobytes())
        groups.setdefault(key, (nsig, tpos, nuniq, gather, []))[4].append((k, a + s0, a, b))
    for nsig, tpos, nuniq, gather, members in groups.values():
        ntime = nsig if tpos is None else nuniq
        # bound the two complex work arrays (product and one order's transform) to ~8 GiB
        per_trial = ntime * scales.size * nchan * 8 * 2
        bmax = max(1, int((8 << 30) // max(per_trial, 1)))
        for i0 in range(0, len(members), bmax):
            part = members[i0:i0 + bmax]
            starts = torch.tensor([m[1] for m in part], dtype=torch.int64, device=device)
            lo = torch.tensor([m[2] for m in part], dtype=torch.int64, device=device)
            hi = torch.tensor([m[3] for m in part], dtype=torch.int64, device=device)
            # pow / abs need moduli only: the plans deliver |W| as float32 and the product stays real (half the
            # traffic, no phase arithmetic); every other output folds the complex transforms
            real = output in ("pow", "abs")
            wdt = torch.float32 if real else torch.complex64
            acc = torch.empty((len(part), ntime, scales.size, nchan), dtype=wdt, device=device)
            for n, (cycles, sc0, expo) in enumerate(steps):
                pkey = ("sl", nsig, nchan, scales[sc0:].tobytes(), dt, cycles, polyremoval, real,
                        None if tpos is None else tpos.tobytes(), str(device), _cwt_reference())
                plan = hs._cache_hit(_cwt_plans, pkey)
                if plan is None:
                    plan = hs.backend.CWTPlan(nsig, nchan, scales[sc0:], dt, detrend=polyremoval,
                                              output="abs" if real else "fourier", tpos=tpos,
                                              ntime_out=nuniq, device=device, sl_cycles=cycles)
                    _cwt_precision(plan)
                    hs._bounded_put(_cwt_plans, pkey, plan)
                buf = hs.backend.handover_buffer(plan.out_shape(len(part)), device, dtype=wdt)
                spec = plan.execute(dev, starts, lo, hi, chan_idx=ci, out=buf)
                # POW: the product is squared on its way out - with the last factor if that one covers every scale
                # (multiplicative set), else in one more pass (adaptive set: later orders cover fewer scales)
                fused_sq = real and output == "pow" and n == len(steps) - 1 and sc0 == 0
                hs.backend.slt_combine(acc, spec, sc0, expo, init=(n == 0), square=fused_sq)
            if real and output == "pow" and not fused_sq:
                hs.backend.slt_combine(acc, acc, 0, np.zeros(scales.size), init=False, square=True)
            res = acc if real else hs.backend.spec_convert(acc, output)
            for i, m in enumerate(part):
                r = res[i]
                if gather is not None:
                    r = r.index_select(0, torch.from_numpy(gather).to(device))
                results[m[0]] = r.unsqueeze(1)
    return results


def superlet_cF(trl_dat, preselect, postselect, toi=None, timeAxis=0, polyremoval=0, output="pow", noCompute=False,
                chunkShape=None, method_kwargs=None):
    """Superlet transform of one trial (signature of specest/compRoutines.py:655-764); returns
    (nTime, 1, nScales, nChannel)."""
    dat = trl_dat.T if timeAxis != 0 else trl_dat
    nChannels = dat.shape[1]
    nTime = toi.size if isinstance(toi, np.ndarray) else dat.shape[0]
    nScales = method_kwargs["scales"].size
    outShape = (nTime, 1, nScales, nChannels)
    if noCompute:
        return outShape, spectralDTypes[output]
    dev = _as_device_trial(trl_dat, timeAxis)
    res = _superlet_device(dev, [(0, dev.shape[0])], [preselect], [postselect], None, polyremoval, output,
                           method_kwargs)[0]
    return hs.backend.to_host(res)


class SuperletTransform(ComputationalRoutine):
    computeFunction = staticmethod(superlet_cF)
    valid_kws = ["preselect", "postselect", "toi", "timeAxis", "polyremoval", "output", "method_kwargs", "samplerate",
                 "scales", "order_max", "order_min", "c_1", "adaptive"]

    def compute_hip(self, data, out):
        cfg = self.cfg
        dev = data.device_data()
        rows, chans = device_rows(data), selected_channels(data)
        mine = list(self.my_trials())
        pre = [self._argv(k)[0] for k in mine]
        post = [self._argv(k)[1] for k in mine]
        parts = _superlet_device(dev, [rows[k] for k in mine], pre, post, chans, cfg["polyremoval"], cfg["output"],
                                 cfg["method_kwargs"])
        _store_trials(self, out, parts)

    def process_metadata(self, data, out):
        propagate_properties(data, out, self.keeptrials, time_axis=True)
        out.trialdefinition, out.samplerate = _make_trialdef(self.cfg, out.trialdefinition.copy(), data.samplerate)
        out.taper = np.array(["None"])
        out.freq = getattr(self, "_foi", None)


def _store_trials(cr, out, parts, stack=False):
    """Results of this rank's trials -> `out.data`: concatenated along the stacking axis in rank order
    (keeptrials) or summed sequentially in the output dtype, all-reduced once and divided by the global
    trial count (computational_routine.py:1022-1032 / kwarg_decorators.py:723-735)."""
    mine = list(cr.my_trials())
    if stack:
        parts = [p.unsqueeze(0) for p in parts]
    for k, p in zip(mine, parts):
        if tuple(p.shape) != tuple(cr.targetShapes[k]):
            raise ValueError(f"trial {k}: result shape {tuple(p.shape)} != dry-run shape {cr.targetShapes[k]}")
    dev = torch.device("cuda")
    dtype = torch.complex64 if np.issubdtype(cr.dtype, np.complexfloating) else torch.float32
    if cr.keeptrials:
        tail = tuple(cr.outputShape[1:])
        local = hs.backend.to_host(torch.cat(parts, dim=0)) if parts else np.zeros((0,) + tail, dtype=cr.dtype)
        out.data = parallel.gather_trials(local).reshape(cr.outputShape)
        return
    if parts:
        # (results that are consecutive pieces of one device array - trials batched into one launch - are not copied)
        p0, nb = parts[0], parts[0].numel() * parts[0].element_size()
        if (len(parts) > 1 and p0.is_contiguous() and p0._base is not None and p0._base.is_contiguous() and
                all(q._base is p0._base and q.shape == p0.shape and q.data_ptr() == p0.data_ptr() + i * nb
                    for i, q in enumerate(parts))):
            first = (p0.data_ptr() - p0._base.data_ptr()) // nb
            stacked = p0._base.reshape((-1,) + tuple(p0.shape))[first:first + len(parts)]
        else:
            stacked = torch.stack(parts, dim=0).contiguous()
        n = stacked.shape[0]
        if stacked.dtype == torch.float32:
            total = hs.backend.trial_mean(stacked) * n
        else:
            total = torch.view_as_complex(hs.backend.trial_mean(torch.view_as_real(stacked).contiguous())) * n
    else:
        total = torch.zeros(cr.outputShape, dtype=dtype, device=dev)
    parallel.allreduce_sum_(total)
    out.data = hs.backend.to_host((total / cr.numTrials)).reshape(cr.outputShape)


def _make_trialdef(cfg, trialdefinition, samplerate):
    """Timing of time-frequency outputs (what specest/compRoutines.py:813-900 arrives at): the number of time points
    every trial contributes, their stacking positions, trigger offsets and the output sampling rate.
      `toi` array   : toi.size points per trial; rate 1/step for evenly spaced points (else 1.0), offset toi[0] * rate
      `toi` fraction: one point per hop of the sliding window; offsets and rate divided by the hop
      `toi` 'all'   : one point per input sample; the first trial keeps its original start sample"""
    trl = np.array(trialdefinition, dtype=float)
    n_in = trl[:, 1] - trl[:, 0]
    toi = cfg["toi"]
    origin = 0.0
    if isinstance(toi, np.ndarray):
        n_out = np.full(trl.shape[0], float(toi.size))
        steps = np.diff(toi)
        samplerate = 1 / (toi[1] - toi[0]) if steps.size and np.allclose(steps, steps[0]) else 1.0
        trl[:, 2] = toi[0] * samplerate
    elif isinstance(toi, numbers.Number):
        hop = cfg["method_kwargs"]["nperseg"] - cfg["method_kwargs"]["noverlap"]
        n_out = np.ceil(n_in / hop)
        trl[:, 2] /= hop
        samplerate = np.round(samplerate / hop, 2)
    else:
        n_out, origin = n_in, trl[0, 0]
    stop = np.cumsum(n_out)
    trl[:, 0] = stop - n_out
    trl[:, 1] = stop
    trl[0, 0] += origin
    return trl, samplerate
