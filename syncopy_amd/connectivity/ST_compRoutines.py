"""Single-trial cross-spectra on MI355X (signatures of syncopy/connectivity/ST_compRoutines.py:
cross_spectra_cF:269 / CrossSpectra:427, spectral_dyadic_product_cF:30)."""
from hashlib import blake2b

import numpy as np
import torch

from .. import backend, parallel
from ..datatype import selected_channels, trial_rows
from ..shared.computational_routine import ComputationalRoutine, propagate_properties
from ..shared.const_def import spectralDTypes
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
                 single_acc=False):
    """Accumulate sum_k X_k X_k^H of every trial in `rows` into `acc_of_trial(i)` (device (F,C,C) c64).
    `single_acc`: every trial lands in the same accumulator, so the spectra may take the channel-blocked
    hand-over layout between the FFT and the CSD kernel (coalesced stores, identical results)."""
    ntaper = 1
    for sel, spec in hs.run_mtmfft_batches(dev, rows, chans, nSamples, taper, taper_opt, demean_taper, False,
                                           polyremoval, freq_idx, "fourier", True,
                                           blocked=single_acc and backend.USE_BLOCKED_HANDOVER, reuse=True):
        ntaper = spec.spyhip_ntaper
        if spec.spyhip_blocked:
            backend.csd_accumulate(spec, acc_of_trial(sel[0]), blocked=True)
            continue
        groups = {}
        for k, i in enumerate(sel):
            groups.setdefault(id(acc_of_trial(i)), (acc_of_trial(i), []))[1].append(k)
        for acc, ks in groups.values():
            if len(ks) == spec.shape[0]:
                backend.csd_accumulate(spec, acc)
            else:
                for k in ks:
                    backend.csd_accumulate(spec[k], acc)
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
    K = _csd_of_rows(dev, [(0, dev.shape[0])], None, nSamples, taper, taper_opt, demean_taper, pr, freq_idx,
                     lambda i: acc, single_acc=True)
    backend.csd_finalize(acc, 1.0 / K)
    return backend.to_host(acc)[np.newaxis], {"freqs_hash": _freqs_hash(freqs)}


def spectral_dyadic_product_cF(specs, send_idx=None, send_N=None, rec_idx=None, rec_N=None, chunkShape=None,
                               noCompute=False):
    """Single-trial cross spectra straight from complex spectra (nTime, nTaper, nFreq, N): the outer product
    over channels, averaged over tapers (syncopy/connectivity/ST_compRoutines.py:30-117) - the MFMA kernel with
    the tapers of one time sample as rows.  Returns (nTime, nFreq, N, N) complex64."""
    if send_idx is not None:
        raise NotImplementedError("channelcmb (rectangular sender/receiver blocks) is listed as 'next' in SURVEY.md 8f")
    nTime, nTaper, nFreq, nChannels = specs.shape
    outShape = (nTime, nFreq, nChannels, nChannels)
    if noCompute:
        return outShape, spectralDTypes["fourier"]
    backend.require_gpu()
    dev = torch.from_numpy(np.ascontiguousarray(specs, dtype=np.complex64)).cuda()
    acc = torch.zeros(outShape, dtype=torch.complex64, device=dev.device)
    for t in range(nTime):
        backend.csd_accumulate(dev[t].contiguous(), acc[t])
        backend.csd_finalize(acc[t], 1.0 / nTaper)
    return backend.to_host(acc)


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
        if chans is not None:
            host = host[..., chans]
        F, C = self.targetShapes[0][1], self.targetShapes[0][2]
        K = host.shape[1]
        lens = {rows[k][1] - rows[k][0] for k in range(T)}
        if self.keeptrials or lens != {1}:
            # time-resolved spectra or kept trials: per-trial compute function (one launch per time sample)
            parts = [torch.from_numpy(self.computeFunction(host[rows[k][0]:rows[k][1]])).cuda() for k in mine]
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
        out._dev = acc.reshape(self.outputShape)
        out.data = backend.to_host(out._dev)

    def process_metadata(self, data, out):
        time_axis = bool(np.any(np.diff(data.trialdefinition)[:, 0] != 1))
        propagate_properties(data, out, self.keeptrials, time_axis)
        out.freq = data.freq


class CrossSpectra(ComputationalRoutine):
    dimord = ["time", "freq", "channel_i", "channel_j"]
    computeFunction = staticmethod(cross_spectra_cF)
    valid_kws = ["samplerate", "nSamples", "foi", "taper", "taper_opt", "demean_taper", "polyremoval", "timeAxis",
                 "tapsmofrq", "nTaper", "pad", "output"]

    def compute_hip(self, data, out):
        """Trial-accumulated CSD straight from the in-HBM trial queue (MFMA rank-K updates)."""
        cfg = self.cfg
        dev = data.device_data()
        rows, chans = trial_rows(data), selected_channels(data)
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
        K = _csd_of_rows(dev, rows, chans, cfg["nSamples"], cfg["taper"], cfg["taper_opt"], cfg["demean_taper"], pr,
                         freq_idx, getter, single_acc=not self.keeptrials) if rows else 1
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

            def device_csd():
                if state["final"] is None:
                    backend.csd_finalize(state["acc"], scale)
                    state["final"] = state["acc"].reshape(shape)
                    out._acc_raw = None
                return state["final"]

            out._acc_raw, out._acc_scale = acc, scale
            out._dev_thunk = device_csd
            out.set_pending(lambda: backend.to_host(device_csd()), shape, np.complex64)

    def process_metadata(self, data, out):
        propagate_properties(data, out, self.keeptrials)
        out.freq = self.cfg["foi"]
