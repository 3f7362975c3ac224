"""Compute functions on trial averages (signatures of syncopy/connectivity/AV_compRoutines.py:
normalize_csd_cF:36 / NormalizeCrossSpectra:115, granger_cF:293 / GrangerCausality:415)."""
import numpy as np
import torch

from .. import backend, parallel
from ..shared.computational_routine import ComputationalRoutine
from ..shared.const_def import spectralDTypes
from ..shared.errors import SPYValueError


def normalize_csd_cF(csd_av_dat, output="abs", chunkShape=None, noCompute=False):
    """Coherency C_ij = S_ij / sqrt(S_ii S_jj) of the trial-averaged CSD (nTime, nFreq, N, N)."""
    outShape = csd_av_dat.shape
    fmt = spectralDTypes["fourier"] if output in ("complex", "fourier") else spectralDTypes["abs"]
    if noCompute:
        return outShape, fmt
    backend.require_gpu()
    dev = torch.from_numpy(np.ascontiguousarray(csd_av_dat, dtype=np.complex64)).cuda()
    res = [backend.coh_normalize(dev[t], output) for t in range(dev.shape[0])]
    return backend.to_host(torch.stack(res, dim=0))


def pairwise_phase_consistency(st_csd):
    """PPC of kept single-trial cross spectra (T, F, Ni, Nj) complex64 -> (1, F, Ni, Nj) float32: what the loop over
    PPC_column routines and its weighted average amount to (connectivity_analysis.py:624-663;
    ST_compRoutines.py:159-233), through K7's closed form - unit phasors summed over trials, then
    (|U|^2 - T)/(T(T-1))."""
    backend.require_gpu()
    st_csd = np.asarray(st_csd)
    T = st_csd.shape[0]
    acc = torch.zeros(st_csd.shape[1:], dtype=torch.complex64, device="cuda")
    per = max(1, int((1 << 30) // max(1, acc.numel() * 8)))
    for t0 in range(0, T, per):
        dev = torch.from_numpy(np.ascontiguousarray(st_csd[t0:t0 + per], dtype=np.complex64)).cuda()
        backend.ppc_accumulate_csd(dev, acc)
    return backend.to_host(backend.ppc_finalize(acc, T, lower_only=False))[np.newaxis]


class _AverageRoutine(ComputationalRoutine):
    dimord = ["time", "freq", "channel_i", "channel_j"]

    def pre_check(self):
        if self.numTrials is None:
            raise SPYValueError("Initialize the computational Routine first!", varname=self.__class__.__name__,
                                actual="ComputationalRoutine not initialized!")
        if self.numTrials != 1:
            raise SPYValueError("1 trial: normalizations can only be done on averaged quantities!", varname="data",
                                actual=f"DataSet contains {self.numTrials} trials")

    def _device_input(self, data):
        dev = getattr(data, "_dev", None)
        if dev is None:
            dev = torch.from_numpy(np.ascontiguousarray(data.data, dtype=np.complex64)).cuda()
        return dev

    def process_metadata(self, data, out):
        out.channel_i = np.array(data.channel_i)
        out.channel_j = np.array(data.channel_j)
        out.freq = data.freq
        out.trialdefinition = data.trialdefinition.copy()
        out.samplerate = data.samplerate


class NormalizeCrossSpectra(_AverageRoutine):
    computeFunction = staticmethod(normalize_csd_cF)
    method = ""
    valid_kws = ["output"]

    def evaluate_device(self, csd):
        """AV stage on one device CSD (F, C, C) -> (F, C, C); used by the streaming jackknife."""
        return backend.coh_normalize(csd.contiguous(), self.cfg["output"])

    def jackknife_accumulate(self, spec, ntaper, csd, direct, ntrials_total, sum_d, sum_d2):
        """Pass 2 of the streaming jackknife for a batch of trials in ONE kernel (K9): single-trial cross spectra,
        leave-one-out averages, their coherence and the sums of d_t / |d_t|^2 never leave the registers."""
        backend.jack_coh_accumulate(spec, ntaper, csd.contiguous(), direct.contiguous(), self.cfg["output"],
                                    ntrials_total, sum_d, sum_d2)

    def compute_hip(self, data, out):
        raw = getattr(data, "_acc_raw", None)
        events = getattr(raw, "spyhip_range_events", None) if raw is not None else None
        if events:
            # the ST stage's last update came frequency range by frequency range: normalise and ship each range to a
            # page-locked landing block while the next is still being accumulated (backend.coh_pipeline) - reading
            # `.data` then waits for the copy of the last range only (compute_sequential hands over a result the caller
            # can read, computational_routine.py:1022-1036)
            res, landing = backend.coh_pipeline(raw, data._acc_scale, self.cfg["output"], events)
            res = res.unsqueeze(0)
            out._dev = res
            shape = tuple(res.shape)
            out.set_pending((lambda: landing.array().reshape(shape)) if landing is not None else (lambda: backend.to_host(res)),
                            shape, np.complex64 if res.is_complex() else np.float32)
            return
        if raw is not None:
            # straight from the ST stage's raw accumulator: scale + normalise + convert + mirror in one pass
            res = backend.coh_from_accumulator(raw, data._acc_scale, self.cfg["output"]).unsqueeze(0)
            out._dev = res
            # the 0.5 GB result stays in HBM until somebody reads `.data` (then one pinned, chunked copy): a chained
            # analysis or a plot of a few channel pairs never pays for the whole array
            out.set_pending(lambda: backend.to_host(res), tuple(res.shape),
                            np.complex64 if res.is_complex() else np.float32)
            return
        dev = self._device_input(data)
        res = torch.stack([backend.coh_normalize(dev[t].contiguous(), self.cfg["output"])
                           for t in range(dev.shape[0])], dim=0)
        out._dev = res
        out.data = backend.to_host(res)


def granger_cF(csd_av_dat, rtol=5e-6, nIter=100, cond_max=1e4, chunkShape=None, noCompute=False):
    """Pairwise Granger-Geweke causality from the trial-averaged CSD (1, nFreq, N, N): regularisation,
    Wilson factorisation and the Granger formula all run on the device.  Returns (1, nFreq, N, N) float32
    and the reference's metadata keys (AV_compRoutines.py:404-409)."""
    outShape = csd_av_dat.shape
    if noCompute:
        return outShape, spectralDTypes["abs"]
    backend.require_gpu()
    dev = torch.from_numpy(np.ascontiguousarray(csd_av_dat[0], dtype=np.complex64)).cuda()
    G, meta = backend.granger(dev, rtol=rtol, niter=nIter, cond_max=cond_max, eps_max=1e-1)
    return backend.to_host(G)[None, ...], _granger_metadata(meta)


def _granger_metadata(meta):
    return {
        "converged--bool": np.array(meta["converged"]),
        "max rel. err--float": np.array(meta["max rel. err"]),
        "reg. factor--float": np.array(meta["reg. factor"]),
        "initial cond. num--float": np.array(meta["initial cond. num"]),
    }


class GrangerCausality(_AverageRoutine):
    computeFunction = staticmethod(granger_cF)
    method = ""
    valid_kws = ["rtol", "nIter", "cond_max"]
    metadata_keys = ("converged", "max rel. err", "reg. factor", "initial cond. num")

    def evaluate_device(self, csd):
        """AV stage on one device CSD (F, C, C) -> Granger (F, C, C) float32; keeps the metadata of the first call
        (the direct estimate) for `out.info`."""
        if not self.metadata:                    # the direct estimate: every rank calls with the same CSD
            G, meta = self._granger(csd.contiguous())
            self.metadata = [_granger_metadata(meta)]
            return G
        # jackknife replicates belong to the rank that owns the left-out trial: no collective here
        return backend.granger(csd.contiguous(), rtol=self.cfg["rtol"], niter=self.cfg["nIter"],
                               cond_max=self.cfg["cond_max"], eps_max=1e-1)[0]

    def _granger(self, csd):
        """One trial-averaged CSD every rank holds -> Granger values on every rank.  With a process group the
        frequencies are sharded over the ranks (wilson_sharded.py); otherwise one spyhip_granger call."""
        kw = dict(rtol=self.cfg["rtol"], niter=self.cfg["nIter"], cond_max=self.cfg["cond_max"], eps_max=1e-1)
        if parallel.collective_active():
            from .wilson_sharded import granger_hip_sharded
            return granger_hip_sharded(csd, **kw)
        return backend.granger(csd, **kw)

    def compute_hip(self, data, out):
        dev = self._device_input(data)
        res, self.metadata = [], []
        for t in range(dev.shape[0]):            # one trial average, or the jackknife's leave-one-out replicates
            G, meta = self._granger(dev[t].contiguous())
            res.append(G)
            self.metadata.append(_granger_metadata(meta))
        out._dev = torch.stack(res, dim=0)
        out.data = backend.to_host(out._dev)

    def process_metadata(self, data, out):
        super().process_metadata(data, out)
        for key, value in (self.metadata[0] or {}).items():
            label, cast = key.split("--")
            out.info[label] = bool(value) if cast == "bool" else float(value)


def normalize_ccov_cF(trl_av_dat, chunkShape=None, noCompute=False):
    """Cross-correlation from the trial-averaged cross-covariance (nLags, 1, N, N): divided by the square roots of
    the zero-lag auto-covariances (AV_compRoutines.py:166-228)."""
    outShape = trl_av_dat.shape
    if noCompute:
        return outShape, spectralDTypes["abs"]
    backend.require_gpu()
    dev = torch.from_numpy(np.ascontiguousarray(trl_av_dat, dtype=np.float32)).cuda()
    return backend.to_host(backend.ccov_normalize_(dev[:, 0].contiguous()))[:, np.newaxis]


class NormalizeCrossCov(_AverageRoutine):
    computeFunction = staticmethod(normalize_ccov_cF)
    method = ""
    valid_kws = []

    def compute_hip(self, data, out):
        dev = getattr(data, "_dev", None)
        if dev is None:
            dev = torch.from_numpy(np.ascontiguousarray(data.data, dtype=np.float32)).cuda()
        res = backend.ccov_normalize_(dev[:, 0].contiguous().clone()).unsqueeze(1)
        out._dev = res
        out.data = backend.to_host(res)
