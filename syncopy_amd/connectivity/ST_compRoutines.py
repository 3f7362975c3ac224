"""Single-trial cross-spectra and cross-covariances on MI355X (signatures of
syncopy/connectivity/ST_compRoutines.py: cross_spectra_cF:269 / CrossSpectra:427, spectral_dyadic_product_cF:30,
cross_covariance_cF:466 / CrossCovariance:587)."""
import weakref
from hashlib import blake2b

import numpy as np
import torch

from .. import backend, parallel
from ..datatype import device_rows, selected_channels, trial_rows
from ..shared.computational_routine import ComputationalRoutine, propagate_properties
from ..shared.const_def import spectralDTypes
from ..shared.errors import SPYValueError
from ..shared.tools import best_match
from ..specest import hip_spectral as hs


def _freqs_hash(freqs):
    return np.array(blake2b(freqs).hexdigest().encode("utf-8"))


def _freq_selection(nSamples, samplerate, foi):
    freqs = np.fft.rfftfreq(nSamples, 1 / samplerate)
    if foi is not None:
        _, freq_idx = best_match(freqs, foi, squash_duplicates=True)
    else:
        freq_idx = np.arange(freqs.size)
    return freqs, freq_idx


def _csd_of_rows(dev, rows, chans, nSamples, taper, taper_opt, demean_taper, polyremoval, freq_idx, acc_of_trial,
                 single_acc=False, upload=None, ranges=None):
    """Accumulate sum_k X_k X_k^H of every trial in `rows` into `acc_of_trial(i)` (device (F,C,C) c64).
    `single_acc`: every trial lands in the same accumulator, so the spectra may take the channel-blocked
    hand-over layout between the FFT and the CSD kernel (coalesced stores, identical results)."""
    ntaper = 1
    for sel, spec in hs.run_mtmfft_batches(dev, rows, chans, nSamples, taper, taper_opt, demean_taper, False,
                                           polyremoval, freq_idx, "fourier", True,
                                           blocked=single_acc and backend.USE_BLOCKED_HANDOVER, reuse=True, upload=upload):
        ntaper = spec.spyhip_ntaper
        if spec.spyhip_blocked:
            backend.csd_accumulate(spec, acc_of_trial(sel[0]), blocked=True)
            continue
        am = getattr(spec, "spyhip_absmax", None)       # range of the whole batch: a bound for each of its trials too
        if single_acc:                                  # (one accumulator for every trial: no per-trial bookkeeping)
            backend.csd_accumulate(spec, acc_of_trial(sel[0]), absmax=am, ranges=ranges)
            continue
        groups = {}
        for k, i in enumerate(sel):
            groups.setdefault(id(acc_of_trial(i)), (acc_of_trial(i), []))[1].append(k)
        for acc, ks in groups.values():
            if len(ks) == spec.shape[0]:
                backend.csd_accumulate(spec, acc, absmax=am)
            else:
                for k in ks:
                    backend.csd_accumulate(spec[k], acc, absmax=am)
    return ntaper


def cross_spectra_cF(trl_dat, samplerate=1, nSamples=None, foi=None, taper="hann", taper_opt=None,
                     demean_taper=False, polyremoval=False, timeAxis=0, chunkShape=None, noCompute=False):
    """Single-trial cross spectra between all channels; returns (1, nFreq, N, N) complex64."""
    dat = trl_dat.T if timeAxis != 0 else trl_dat
    if nSamples is None:
        nSamples = dat.shape[0]
    nChannels = dat.shape[1]
    freqs, freq_idx = _freq_selection(nSamples, samplerate, foi)
    outShape = (1, freq_idx.size, nChannels, nChannels)
    if noCompute:
        return outShape, spectralDTypes["fourier"]
    backend.require_gpu()
    dev = torch.from_numpy(np.ascontiguousarray(dat, dtype=np.float32)).cuda()
    acc = torch.zeros(outShape[1:], dtype=torch.complex64, device=dev.device)
    pr = polyremoval if polyremoval in (0, 1) and polyremoval is not False else None
    with hs.per_trial_route():
        K = _csd_of_rows(dev, [(0, dev.shape[0])], None, nSamples, taper, taper_opt, demean_taper, pr, freq_idx,
                         lambda i: acc, single_acc=True)
    backend.csd_finalize(acc, 1.0 / K)
    return backend.to_host(acc)[np.newaxis], {"freqs_hash": _freqs_hash(freqs)}


def spectral_dyadic_product_cF(specs, send_idx=None, send_N=None, rec_idx=None, rec_N=None, chunkShape=None,
                               noCompute=False):
    """Single-trial cross spectra straight from complex spectra (nTime, nTaper, nFreq, N): the outer product
    over channels, averaged over tapers (syncopy/connectivity/ST_compRoutines.py:30-117) - the MFMA kernel with
    the tapers of one time sample as rows.  Returns (nTime, nFreq, N, N) complex64."""
    nTime, nTaper, nFreq, nChannels = specs.shape
    sub = _cmb_union(send_idx, rec_idx)
    outShape = (nTime, nFreq, nChannels, nChannels) if sub is None else (nTime, nFreq, len(sub[1]), len(sub[2]))
    if noCompute:
        return outShape, spectralDTypes["fourier"]
    backend.require_gpu()
    if sub is not None:
        specs = np.asarray(specs)[..., sub[0]]
    dev = torch.from_numpy(np.ascontiguousarray(specs, dtype=np.complex64)).cuda()
    n = dev.shape[-1]
    acc = torch.zeros((nTime, nFreq, n, n), dtype=torch.complex64, device=dev.device)
    for t in range(nTime):
        backend.csd_accumulate(dev[t].contiguous(), acc[t])
        backend.csd_finalize(acc[t], 1.0 / nTaper)
    if sub is not None:
        acc = _cmb_block(acc, sub)
    return backend.to_host(acc)


def _cmb_union(send_idx, rec_idx):
    """`channelcmb` (connectivity_analysis.py:501-529): the rectangular sender x receiver block is cut out of the
    Hermitian product of the channels that occur at all - (channels of the union, senders' and receivers' positions
    in it), or None without `channelcmb`."""
    if send_idx is None:
        return None
    send, rec = np.asarray(send_idx, dtype=int).ravel(), np.asarray(rec_idx, dtype=int).ravel()
    union = np.unique(np.concatenate((send, rec)))
    return union, np.searchsorted(union, send), np.searchsorted(union, rec)


def _cmb_block(acc, sub):
    """acc[..., senders, receivers] on the device."""
    si = torch.as_tensor(sub[1], device=acc.device)
    ri = torch.as_tensor(sub[2], device=acc.device)
    return acc.index_select(-2, si).index_select(-1, ri).contiguous()


class SpectralDyadicProduct(ComputationalRoutine):
    """CrossSpectra's sibling for SpectralData input (`freqanalysis(output="fourier", keeptapers=True)` chained into
    `connectivityanalysis`): nothing but K4 on spectra that already exist."""
    dimord = ["time", "freq", "channel_i", "channel_j"]
    computeFunction = staticmethod(spectral_dyadic_product_cF)
    valid_kws = ["send_idx", "send_N", "rec_idx", "rec_N", "output"]

    def compute_hip(self, data, out):
        """All trials in one go: the whole (rows, nFreq, N) block of a rank's trials goes through the MFMA kernel
        (keeptrials=False), or one launch per time sample (keeptrials=True)."""
        rows, chans = trial_rows(data), selected_channels(data)
        T = self.numTrials
        mine = self.my_trials()
        host = np.asarray(data.data)
        sub = _cmb_union(self.cfg.get("send_idx"), self.cfg.get("rec_idx"))
        if sub is not None:
            chans = list(sub[0])                       # a channel selection next to channelcmb is ruled out upstream
        if chans is not None:
            host = host[..., chans]
        F, C = self.targetShapes[0][1], host.shape[-1]
        K = host.shape[1]
        lens = {rows[k][1] - rows[k][0] for k in range(T)}
        if self.keeptrials or lens != {1}:
            # time-resolved spectra or kept trials: per-trial compute function (one launch per time sample)
            kw = {} if sub is None else dict(send_idx=sub[1], rec_idx=sub[2])
            parts = [torch.from_numpy(self.computeFunction(host[rows[k][0]:rows[k][1]], **kw)).cuda() for k in mine]
            from ..specest.compRoutines import _store_trials
            _store_trials(self, out, parts)
            return
        acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
        if len(mine):
            sel = np.concatenate([np.arange(rows[k][0], rows[k][1]) for k in mine])
            dev = torch.from_numpy(np.ascontiguousarray(host[sel], dtype=np.complex64)).cuda()
            backend.csd_accumulate(dev, acc)            # rows = trials x tapers
        backend.csd_allreduce_(acc)
        backend.csd_finalize(acc, 1.0 / (K * T))
        if sub is not None:
            acc = _cmb_block(acc, sub)
        out._dev = acc.reshape(self.outputShape)
        out.data = backend.to_host(out._dev)

    def ppc_hip(self, data):
        """Pairwise phase consistency of spectra that exist already: K7 on the uploaded (trials x tapers, F, C) block of
        every time sample (time-resolved spectra: all trials must have the same number of samples, as the reference's
        accumulator `st_out.trials[0].shape` requires, connectivity_analysis.py:626); `channelcmb` rectangles are
        cut out of the channels that occur at all.  Returns (nTime, F, C_i, C_j) on the device."""
        rows, chans = trial_rows(data), selected_channels(data)
        T = self.numTrials
        lens = {rows[k][1] - rows[k][0] for k in range(T)}
        if len(lens) != 1:
            raise SPYValueError("trials of equal length", varname="data",
                                actual=f"time-resolved spectra with {sorted(lens)} samples per trial")
        L = lens.pop()
        host = np.asarray(data.data)
        sub = _cmb_union(self.cfg.get("send_idx"), self.cfg.get("rec_idx"))
        if sub is not None:
            chans = list(sub[0])
        if chans is not None:
            host = host[..., chans]
        K, F, C = host.shape[1:]
        mine = self.my_trials()
        first = np.array([rows[k][0] for k in mine], dtype=np.int64)
        res = []
        for ti in range(L):
            U = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
            if len(mine):
                dev = torch.from_numpy(np.ascontiguousarray(host[first + ti], dtype=np.complex64)).cuda()
                backend.ppc_accumulate(dev.reshape(-1, F, C), K, U)
            parallel.allreduce_sum_(U)
            r = backend.ppc_finalize(U, T, lower_only=True)
            res.append(r if sub is None else _cmb_block(r, sub))
        return torch.stack(res, dim=0)

    def process_metadata(self, data, out):
        time_axis = bool(np.any(np.diff(data.trialdefinition)[:, 0] != 1))
        propagate_properties(data, out, self.keeptrials, time_axis)
        if self.cfg.get("send_idx") is not None:          # ST_compRoutines.py:151-155
            names = np.array(data.channel)
            out.channel_i = names[np.asarray(self.cfg["send_idx"], dtype=int)]
            out.channel_j = names[np.asarray(self.cfg["rec_idx"], dtype=int)]
        out.freq = data.freq


class CrossSpectra(ComputationalRoutine):
    dimord = ["time", "freq", "channel_i", "channel_j"]
    computeFunction = staticmethod(cross_spectra_cF)
    valid_kws = ["samplerate", "nSamples", "foi", "taper", "taper_opt", "demean_taper", "polyremoval", "timeAxis",
                 "tapsmofrq", "nTaper", "pad", "output"]

    def compute_hip(self, data, out):
        """Trial-accumulated CSD straight from the in-HBM trial queue (MFMA rank-K updates).  A recording that is still
        being uploaded (first call on host data) is consumed chunk by chunk behind the copy (backend.Upload)."""
        cfg = self.cfg
        dev = data.device_data(partial=True)
        upload = data.upload_in_flight()
        rows, chans = device_rows(data), selected_channels(data)
        nS = cfg["nSamples"] if cfg["nSamples"] is not None else rows[0][1] - rows[0][0]
        freqs, freq_idx = _freq_selection(nS, cfg["samplerate"], cfg["foi"])
        pr = cfg["polyremoval"] if cfg["polyremoval"] in (0, 1) and cfg["polyremoval"] is not False else None
        F, C = self.targetShapes[0][1], self.targetShapes[0][2]
        T = self.numTrials
        mine = self.my_trials()                            # this rank's contiguous trial shard
        rows = [rows[k] for k in mine]
        if self.keeptrials:
            acc = torch.zeros((len(rows), F, C, C), dtype=torch.complex64, device=dev.device)
            getter = lambda i: acc[i]                      # noqa: E731
        else:
            acc = torch.zeros((F, C, C), dtype=torch.complex64, device=dev.device)
            getter = lambda i: acc                         # noqa: E731
        # one rank, a result of 64 MB and more, 256 channels: the cross-spectral update goes frequency range by frequency
        # range so that the AV stage can normalise and ship range r under the products of range r + 1 (backend.coh_pipeline)
        ranges = None
        if not self.keeptrials and C == 256 and not parallel.collective_active() and F * C * C * 4 >= (64 << 20):
            ranges = backend.frequency_ranges(F, dev.device)
            if ranges:
                backend.prewarm_landing(F * C * C * 4)       # (the float32 coherence outputs; complex ones find no block and allocate)
        K = _csd_of_rows(dev, rows, chans, cfg["nSamples"], cfg["taper"], cfg["taper_opt"], cfg["demean_taper"], pr,
                         freq_idx, getter, single_acc=not self.keeptrials, upload=upload, ranges=ranges) if rows else 1
        data.device_data()                                  # (an upload in flight ends here at the latest)
        K = int(cfg["taper_opt"].get("Kmax", K)) if cfg["taper_opt"] else K
        self.metadata = [{"freqs_hash": _freqs_hash(freqs)}] * T
        if self.keeptrials:
            for t in range(len(rows)):
                backend.csd_finalize(acc[t], 1.0 / K)
            out._dev = None
            out.data = parallel.gather_trials(backend.to_host(acc)).reshape(self.outputShape)
        else:
            backend.csd_allreduce_(acc)                    # the ONE collective of the path (RCCL over xGMI,
                                                           # lower triangle only)
            # The trial-averaged CSD is finalised (scaled + mirrored) and copied to the host only if somebody asks
            # for it: the coherence stage reads the raw lower-triangle accumulator through the fused kernel.
            scale, shape = 1.0 / (K * T), self.outputShape
            state = {"acc": acc, "final": None}
            out_ref = weakref.ref(out)                     # (the closure lives IN `out`: a strong reference would be a cycle,
                                                           # and a dropped result would keep its 1 GB of HBM until the cyclic
                                                           # collector happens to run)

            def device_csd():
                if state["final"] is None:
                    backend.csd_finalize(state["acc"], scale)
                    state["final"] = state["acc"].reshape(shape)
                    o = out_ref()
                    if o is not None:
                        o._acc_raw = None
                return state["final"]

            out._acc_raw, out._acc_scale = acc, scale
            out._dev_thunk = device_csd
            out.set_pending(lambda: backend.to_host(device_csd()), shape, np.complex64)

    def needs_float64(self, data, method, sample=16):
        """precision="auto" on the batched route: do the data ask for float64 transforms?  Coherence, Granger causality
        and ppc are RATIOS (or phases) of spectra.  The float32 transform leaves an ABSOLUTE error of ~5e-7 of a channel's
        rms bin in every single-trial spectrum - the reference transforms in float64 and has none - which averages down
        with the number N of (trial, taper) products: a bin whose power sits a factor R below the channel's mean carries
        ~5e-7 sqrt(R / N) of error in its coherence (floor of the parity criterion: 1e-6), and 5e-7 sqrt(R) of phase error
        per trial in ppc, where a trial weighs 2 / T in the pair average (floor 5e-6 / sqrt(T)).  R = hs.dynamic_range of
        the float32 spectra of (up to) `sample` of this rank's trials, the largest over the ranks so that every rank
        takes the same branch.  Costs `sample` trials' transforms and one host synchronisation - once per recording and
        set of options: the measured ratio is kept on the data object until its array changes (`invalidate`); AR(2)-type
        data never asks, line noise 50 dB above the floor or 1/f spectra over four decades do."""
        cfg = self.cfg
        dev = data.device_data(partial=True)
        upload = data.upload_in_flight()
        rows, chans = device_rows(data), selected_channels(data)
        nS = cfg["nSamples"] if cfg["nSamples"] is not None else rows[0][1] - rows[0][0]
        _, freq_idx = _freq_selection(nS, cfg["samplerate"], cfg["foi"])
        T = len(rows)
        if method == "ppc" and T <= 16:
            # every trial's unit phasor weighs 2 / T >= 1 / 8 of the estimate: where a single-trial cross spectrum passes
            # near zero its phase is the transform's rounding noise (5e-7 of the spectrum in float32, 6e-8 in the reference's
            # complex64 products of float64 transforms) and nothing averages it away - 1.49 x the criterion for one of
            # 262 144 elements of five 6000-sample trials (tests/test_gpu_fuzz.py family 1300000, seed 790).  A handful of
            # float64 transforms costs nothing.
            return True
        if freq_idx.size < 4:
            return False
        pr = cfg["polyremoval"] if cfg["polyremoval"] in (0, 1) and cfg["polyremoval"] is not False else None
        lo, hi = parallel.my_shard(T)
        mine = rows[lo:hi][:sample]
        looks = getattr(data, "_looks", None)
        key = (tuple(mine), None if chans is None else tuple(int(c) for c in chans), cfg["nSamples"], str(cfg["taper"]),
               repr(sorted((cfg["taper_opt"] or {}).items())), bool(cfg["demean_taper"]), pr, freq_idx.tobytes(), T)
        if looks is not None and key in looks:
            ratio, K = looks[key]
        else:
            ratio, K = 0.0, 1
            with hs.precision("float32"):
                for _, spec in hs.run_mtmfft_batches(dev, mine, chans, cfg["nSamples"], cfg["taper"], cfg["taper_opt"],
                                                     cfg["demean_taper"], False, pr, None, "pow", True, reuse=True,
                                                     upload=upload):
                    K = spec.shape[1]
                    ratio = max(ratio, hs.dynamic_range(spec, kept=freq_idx, few_products=T * K <= 16))   # whole axis: see hs.dynamic_range
            if looks is not None:
                looks[key] = (ratio, K)
        ratio = parallel.allreduce_max(ratio)
        # (7.5e-7, not the 5e-7 of the estimate above: two of 48000 seeded cases sat at 1.2 x and 1.5 x the criterion with float32
        # transforms the plain estimate had let through - profiles/r4_fuzz_offset1300000.log)
        if method == "ppc":
            return bool(7.5e-7 * np.sqrt(ratio) * 2 / T > 5e-6 / np.sqrt(max(T, 1)))
        return bool(7.5e-7 * np.sqrt(ratio / (T * K)) > 1e-6)

    def ppc_hip(self, data):
        """Pairwise phase consistency with the kernels (connectivity_analysis.py:624-663, ST_compRoutines.py:159-233):
        the tapered spectra of each batch of trials go straight into K7, which forms every trial's taper-averaged
        cross spectrum, normalises it to a unit phasor and sums over trials in registers; one sum over ranks, then
        ppc = (|U|^2 - T)/(T(T-1)).  No single-trial cross spectrum is ever stored (the reference keeps all T of them
        and visits T(T-1)/2 pairs).  Returns the (F, C, C) float32 device tensor."""
        cfg = self.cfg
        dev = data.device_data()
        rows, chans = device_rows(data), selected_channels(data)
        nS = cfg["nSamples"] if cfg["nSamples"] is not None else rows[0][1] - rows[0][0]
        freqs, freq_idx = _freq_selection(nS, cfg["samplerate"], cfg["foi"])
        pr = cfg["polyremoval"] if cfg["polyremoval"] in (0, 1) and cfg["polyremoval"] is not False else None
        F, C = self.targetShapes[0][1], self.targetShapes[0][2]
        T = self.numTrials
        mine = [rows[k] for k in self.my_trials()]
        U = torch.zeros((F, C, C), dtype=torch.complex64, device=dev.device)
        for _, spec in hs.run_mtmfft_batches(dev, mine, chans, cfg["nSamples"], cfg["taper"], cfg["taper_opt"],
                                             cfg["demean_taper"], False, pr, freq_idx, "fourier", True, reuse=True):
            backend.ppc_accumulate(spec.reshape(-1, F, C), spec.shape[1], U)
        parallel.allreduce_sum_(U)
        self.metadata = [{"freqs_hash": _freqs_hash(freqs)}] * T
        return backend.ppc_finalize(U, T, lower_only=True)

    def jackknife_hip(self, data, evaluate, fused=None):
        """Streaming jackknife on the device (connectivity_analysis.py:601-606,736-757; statistics/jackknifing.py):
        never more than a handful of (F, C, C) arrays live at once instead of the reference's T single-trial CSDs.
        Pass 1 accumulates the trial-averaged CSD S (one all-reduce); pass 2 re-computes every trial's own CSD S_t,
        forms the leave-one-out average (T*S - S_t)/(T-1) in complex64 exactly as the reference does, evaluates the
        AV stage on it (`evaluate`: (F, C, C) complex64 Hermitian -> (F, C, C) tensor) and keeps only the float64 sums
        of d_t = replicate_t - direct and |d_t|^2.  Returns (S, direct, jack_bias, jack_var) device tensors:
            bias = (T-1) * mean_t d_t,   var = (T-1) * (sum_t |d_t|^2 - |sum_t d_t|^2 / T)
        (= the reference's (T-1) (jack_avg - direct) and (T-1) sum |jack_avg - replicate_t|^2).
        `fused(spec, ntaper, S, direct, T, sum_d, sum_d2)`: optional single-kernel form of pass 2 for a batch of
        trials (NormalizeCrossSpectra.jackknife_accumulate)."""
        cfg = self.cfg
        dev = data.device_data()
        rows, chans = device_rows(data), selected_channels(data)
        nS = cfg["nSamples"] if cfg["nSamples"] is not None else rows[0][1] - rows[0][0]
        freqs, freq_idx = _freq_selection(nS, cfg["samplerate"], cfg["foi"])
        pr = cfg["polyremoval"] if cfg["polyremoval"] in (0, 1) and cfg["polyremoval"] is not False else None
        F, C = self.targetShapes[0][1], self.targetShapes[0][2]
        T = self.numTrials
        mine = [rows[k] for k in self.my_trials()]
        args = (dev, chans, cfg["nSamples"], cfg["taper"], cfg["taper_opt"], cfg["demean_taper"], pr, freq_idx)

        def batches(rr):
            return hs.run_mtmfft_batches(args[0], rr, args[1], args[2], args[3], args[4], args[5], False, args[6],
                                         args[7], "fourier", True, reuse=True)

        # ---- pass 1: trial average
        S = torch.zeros((F, C, C), dtype=torch.complex64, device=dev.device)
        K = 1
        for _, spec in batches(mine):
            K = spec.spyhip_ntaper
            backend.csd_accumulate(spec, S, absmax=getattr(spec, "spyhip_absmax", None))
        K = int(cfg["taper_opt"].get("Kmax", K)) if cfg["taper_opt"] else K
        backend.csd_allreduce_(S)
        backend.csd_finalize(S, 1.0 / (K * T))
        direct = evaluate(S)
        # ---- pass 2: leave-one-out replicates, one trial at a time
        cplx = direct.is_complex()
        sum_d = torch.zeros(direct.shape, dtype=torch.complex128 if cplx else torch.float64, device=dev.device)
        sum_d2 = torch.zeros(direct.shape, dtype=torch.float64, device=dev.device)
        St = None if fused is not None else torch.empty_like(S)
        for _, spec in batches(mine):
            if fused is not None:
                # the AV stage offers the whole pass 2 as one kernel (coherence: K9): replicates never exist in HBM
                fused(spec.reshape(-1, F, C), spec.shape[1], S, direct, T, sum_d, sum_d2)
                continue
            for t in range(spec.shape[0]):
                St.zero_()
                backend.csd_accumulate(spec[t], St, absmax=getattr(spec, "spyhip_absmax", None))
                backend.csd_finalize(St, 1.0 / K)
                loo = T * S - St                       # complex64, the reference's operation order
                loo /= T - 1
                d = (evaluate(loo) - direct).to(sum_d.dtype)
                sum_d += d
                sum_d2 += (d.real ** 2 + d.imag ** 2) if cplx else d * d
        parallel.allreduce_sum_(sum_d)
        parallel.allreduce_sum_(sum_d2)
        bias = ((T - 1) * (sum_d / T)).to(direct.dtype)
        mag2 = (sum_d.real ** 2 + sum_d.imag ** 2) if cplx else sum_d * sum_d
        var = ((T - 1) * (sum_d2 - mag2 / T)).clamp_(min=0).to(torch.float32)
        self.metadata = [{"freqs_hash": _freqs_hash(freqs)}] * T
        return S, direct, bias, var

    def process_metadata(self, data, out):
        propagate_properties(data, out, self.keeptrials)
        out.freq = self.cfg["foi"]


def _padded_spectra(dev, rows, chans, n, polyremoval, max_bytes=8 << 30):
    """Batches (B, F, C) of plain DFT spectra (boxcar, no normalisation) of the equally long trials `rows`,
    zero-padded to the length K8 needs for n samples."""
    L = backend.ccov_nfft(n)
    nchan = dev.shape[1] if chans is None else len(chans)
    ci = None if chans is None else torch.tensor(np.asarray(chans), dtype=torch.int32, device=dev.device)
    # float32 on every route: the reference's own cross-covariances are float32 FFT convolutions
    # (scipy.signal.fftconvolve of float32 trials, ST_compRoutines.py:540-573)
    with hs.precision("float32"):
        plan = hs.get_plan(n, L, nchan, "boxcar", None, n, 1.0, polyremoval, False, None, "fourier", True, dev.device)
    per_trial = (L // 2 + 1) * nchan * 8
    bmax = max(1, int(max_bytes // per_trial))
    for i in range(0, len(rows), bmax):
        starts = torch.tensor([r[0] for r in rows[i:i + bmax]], dtype=torch.int64, device=dev.device)
        buf = backend.handover_buffer(plan.out_shape(len(starts)), dev.device)
        spec = plan.execute(dev, starts, chan_idx=ci, out=buf)
        yield spec.reshape(len(starts), L // 2 + 1, nchan)


def _ccov_trials(dev, rows, chans, polyremoval, scale, norm):
    """K8 on the trials `rows` (equal length): cross spectra of the zero-padded trials summed by the MFMA kernel,
    one inverse transform per channel pair.  Returns ((nlag, C, C) float32 device tensor, raw accumulator)."""
    n = rows[0][1] - rows[0][0]
    L = backend.ccov_nfft(n)
    nchan = dev.shape[1] if chans is None else len(chans)
    acc = torch.zeros((L // 2 + 1, nchan, nchan), dtype=torch.complex64, device=dev.device)
    for spec in _padded_spectra(dev, rows, chans, n, polyremoval):
        backend.csd_accumulate(spec, acc)
    return acc, n


def cross_covariance_cF(trl_dat, samplerate=1, polyremoval=0, timeAxis=0, norm=False, fullOutput=False,
                        chunkShape=None, noCompute=False):
    """Single-trial cross-covariance (`norm=True`: cross-correlation) between all channels for the lags
    0 .. N/2; returns (nLags, 1, N, N) float32 [and the lags in s]."""
    dat = trl_dat.T if timeAxis != 0 else trl_dat
    nSamples, nChannels = dat.shape
    nlag = nSamples // 2 + (nSamples & 1)
    outShape = (nlag, 1, nChannels, nChannels)
    if noCompute:
        return outShape, spectralDTypes["abs"]
    backend.require_gpu()
    dev = torch.from_numpy(np.ascontiguousarray(dat, dtype=np.float32)).cuda()
    pr = polyremoval if polyremoval in (0, 1) and polyremoval is not False else None
    acc, n = _ccov_trials(dev, [(0, nSamples)], None, pr, 1.0, norm)
    CC = backend.to_host(backend.ccov_from_accumulator(acc, n, 1.0, 2 if norm else 0))[:, np.newaxis]
    if fullOutput:
        return CC, np.arange(nlag) / samplerate
    return CC


class CrossCovariance(ComputationalRoutine):
    dimord = ["time", "freq", "channel_i", "channel_j"]
    computeFunction = staticmethod(cross_covariance_cF)
    valid_kws = ["samplerate", "polyremoval", "timeAxis", "norm", "fullOutput"]

    def compute_hip(self, data, out):
        """All trials of this rank at once.  Trial average (keeptrials=False): the cross spectra of the zero-padded
        trials are summed by the MFMA kernel and over ranks, then ONE inverse transform per channel pair - the
        reference's per-trial, per-pair convolutions never happen.  Kept trials: the same per trial."""
        cfg = self.cfg
        dev = data.device_data()
        rows, chans = device_rows(data), selected_channels(data)
        pr = cfg["polyremoval"] if cfg["polyremoval"] in (0, 1) and cfg["polyremoval"] is not False else None
        T = self.numTrials
        mine = [rows[k] for k in self.my_trials()]
        if self.keeptrials:
            parts = []
            for r in mine:
                acc, n = _ccov_trials(dev, [r], chans, pr, 1.0, cfg["norm"])
                parts.append(backend.ccov_from_accumulator(acc, n, 1.0, 2 if cfg["norm"] else 0).unsqueeze(1))
            from ..specest.compRoutines import _store_trials
            _store_trials(self, out, parts)
            return
        lens = {r[1] - r[0] for r in rows}
        if len(lens) != 1:
            raise ValueError("trial averaging of cross-covariances needs trials of equal length")
        n = lens.pop()
        L = backend.ccov_nfft(n)
        nchan = self.targetShapes[0][2]
        acc = torch.zeros((L // 2 + 1, nchan, nchan), dtype=torch.complex64, device=dev.device)
        for spec in _padded_spectra(dev, mine, chans, n, pr):
            backend.csd_accumulate(spec, acc)
        backend.csd_allreduce_(acc)
        res = backend.ccov_from_accumulator(acc, n, 1.0 / T, 2 if cfg["norm"] else 0)
        out._dev = res.unsqueeze(1)
        out.data = backend.to_host(out._dev)

    def process_metadata(self, data, out):
        # lags live on the time axis, offset 0 (ST_compRoutines.py:610-640)
        from ..datatype import selected_trialdefinition
        chans = selected_channels(data)
        names = np.array(data.channel) if chans is None else np.array(data.channel)[chans]
        old = selected_trialdefinition(data)
        sizes = np.ceil(np.diff(old[:, :2], axis=1)[:, 0] / 2)
        si = np.r_[0, np.cumsum(sizes)]
        trl = np.column_stack([si[:-1], si[1:], np.zeros(len(sizes))])
        out.trialdefinition = trl if self.keeptrials else trl[[0], :]
        out.samplerate = data.samplerate
        out.channel_i, out.channel_j = names, names.copy()
