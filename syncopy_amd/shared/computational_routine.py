"""The compute-class boundary: same protocol as syncopy/shared/computational_routine.py
(`initialize` dry run -> `compute` -> `process_metadata`) without HDF5 and Dask.

 * `computeFunction(trl_dat, *argv, chunkShape=None, noCompute=False, **cfg)` is the
   reference's cF contract (doc/source/developer/compute_kernels.rst:63-88): one trial
   in, `(shape, dtype)` for the dry run, an array (optionally `(array, metadata)`) out.
 * `compute(..., method=<name>)` dispatches to `compute_<name>` exactly like the reference
   (computational_routine.py:724-731).  `compute_sequential` is the reference's trial loop
   (:944-1036) on host arrays; `compute_hip` (implemented by the subclasses) runs all
   trials from the in-HBM trial queue in a few launches.
"""
from abc import ABC, abstractmethod
from inspect import signature

import numpy as np

from .. import parallel
from ..datatype import FauxTrial, selected_channels, trial_rows
from .errors import SPYValueError


def parse_cF_returns(res):
    """cF results are `array` or `(array, dict)` (shared/metadata.py:135-176)."""
    if isinstance(res, tuple):
        if len(res) != 2 or not isinstance(res[1], (dict, type(None))):
            raise SPYValueError("ndarray or (ndarray, dict)", varname="computeFunction return value", actual=str(type(res)))
        return res
    return res, None


def _arg_key(v):
    """Hashable stand-in of a compute-function argument (slices, index arrays, nested lists) for the dry-run cache."""
    if isinstance(v, slice):
        return ("slice", v.start, v.stop, v.step)
    if isinstance(v, np.ndarray):
        return ("array", v.shape, v.dtype.str, v.tobytes())
    if isinstance(v, (list, tuple)):
        return ("seq",) + tuple(_arg_key(x) for x in v)
    try:
        hash(v)
        return v
    except TypeError:
        return ("id", id(v))


class ComputationalRoutine(ABC):
    computeFunction = None
    valid_kws = []

    def __init__(self, *argv, **kwargs):
        self.argv = list(argv)
        self.defaultCfg = {k: v.default for k, v in signature(self.computeFunction).parameters.items()
                           if v.default is not v.empty and k not in ("noCompute", "chunkShape")}
        self.cfg = dict(self.defaultCfg)
        for k, v in kwargs.items():
            if k in self.cfg:
                self.cfg[k] = v
        self.keeptrials = None
        self.numTrials = None
        self.targetShapes = None
        self.outputShape = None
        self.dtype = None
        self.chunkShape = None
        self.metadata = []

    # ------------------------------------------------------------------ dry run
    def initialize(self, data, out_stackingdim=0, chan_per_worker=None, keeptrials=True):
        rows = trial_rows(data)
        chans = selected_channels(data)
        self.numTrials = len(rows)
        self.keeptrials = keeptrials
        tax = data.dimord.index("time") if "time" in data.dimord else 0
        shapes, dtp = [], None
        per_trial_args = any(isinstance(a, (list, tuple, np.ndarray)) and len(a) == self.numTrials for a in self.argv)
        seen = {}                        # dry runs of equally long trials are identical (unless argv is per trial)
        base = list(data.data_shape)
        if chans is not None and "channel" in data.dimord:
            base[data.dimord.index("channel")] = len(chans)
        for k, (a, b) in enumerate(rows):
            # (per-trial arguments - the sample and frame selections of the time-frequency methods - are usually the same
            # few values over and over: they join the key instead of forcing one dry run per trial)
            key = (b - a, tuple(_arg_key(v) for v in self._argv(k))) if per_trial_args else b - a
            hit = seen.get(key) if key is not None else None
            if hit is None:
                shp = list(base)
                shp[tax] = b - a
                trial = FauxTrial(shp, data.data_dtype)
                chk, dt = self.computeFunction(trial, *self._argv(k), noCompute=True, chunkShape=None, **self.cfg)
                hit = (tuple(int(s) for s in chk), np.dtype(dt))
                if key is not None:
                    seen[key] = hit
            shapes.append(hit[0])
            dtp = hit[1]
        self.targetShapes = shapes
        self.dtype = dtp
        stack = out_stackingdim
        if not keeptrials:
            if len(set(shapes)) != 1:
                raise NotImplementedError("Averaging trials of unequal lengths in output currently not supported!")
            self.outputShape = shapes[0]
        else:
            tot = sum(s[stack] for s in shapes)
            first = list(shapes[0])
            if any(s[:stack] + s[stack + 1:] != shapes[0][:stack] + shapes[0][stack + 1:] for s in set(shapes)):
                raise SPYValueError("identical non-stacking dimensions of all trial results", varname="data")
            first[stack] = tot
            self.outputShape = tuple(first)
        self.chunkShape = max(set(shapes), key=lambda s: int(np.prod(s)))
        self.stackingDim = stack

    def _argv(self, k):
        return tuple(a[k] if isinstance(a, (list, tuple, np.ndarray)) and len(a) == self.numTrials else a
                     for a in self.argv)

    # ------------------------------------------------------------------ compute
    def compute(self, data, out, parallel=False, log_dict=None, method=None):
        if self.numTrials is None:
            raise SPYValueError("Initialize the computational Routine first!", varname=self.__class__.__name__,
                                actual="ComputationalRoutine not initialized!")
        if method is None:
            method = "hip" if hasattr(self, "compute_hip") else "sequential"
        fn = getattr(self, "compute_" + method, None)
        if fn is None:
            raise AttributeError(f"Unknown computation method `{method}`")
        self.metadata = []
        fn(data, out)
        self.process_metadata(data, out)
        out.cfg = dict(log_dict or {})
        out.log = f"computed {self.computeFunction.__name__} on {self.numTrials} trials ({method})"

    def _host_trial(self, data, k, rows, chans):
        a, b = rows[k]
        tax = data.dimord.index("time")
        arr = data.data[a:b] if tax == 0 else data.data[:, a:b]
        if chans is not None:
            arr = arr[:, chans] if tax == 0 else arr[chans, :]
        # fresh C-ordered copy, as the reference hands the cF (computational_routine.py:1001: h5py reads a selection into
        # a new C-contiguous array).  NumPy's own `arr[:, list]` is laid out column by column, and np.mean(axis=0) - the
        # detrending of every cF - then sums each column PAIRWISE instead of row by row: a different float32 rounding
        return np.array(arr, order="C")

    def my_trials(self):
        """Trial indices of this rank (all trials without a process group; parallel.py)."""
        lo, hi = parallel.my_shard(self.numTrials)
        return range(lo, hi)

    def compute_sequential(self, data, out):
        """Trial loop of the reference: call the cF per trial, stack or accumulate in the output dtype.
        With a process group each rank loops over its contiguous trial shard; partial sums are
        all-reduced once, stacked results are concatenated in rank order."""
        rows = trial_rows(data)
        chans = selected_channels(data)
        mine = self.my_trials()
        if self.keeptrials:
            shp = list(self.outputShape)
            shp[self.stackingDim] = sum(self.targetShapes[k][self.stackingDim] for k in mine)
            target = np.zeros(shp, dtype=self.dtype)
        else:
            target = np.zeros(self.outputShape, dtype=self.dtype)
        self.metadata = [None] * self.numTrials
        pos = 0
        for k in mine:
            arr = self._host_trial(data, k, rows, chans)
            res, details = parse_cF_returns(self.computeFunction(arr, *self._argv(k), chunkShape=self.chunkShape,
                                                                 noCompute=False, **self.cfg))
            res = np.asarray(res).reshape(self.targetShapes[k])
            self.metadata[k] = details
            if self.keeptrials:
                n = res.shape[self.stackingDim]
                idx = [slice(None)] * res.ndim
                idx[self.stackingDim] = slice(pos, pos + n)
                target[tuple(idx)] = res
                pos += n
            else:
                target += res
        if not self.keeptrials:
            target = parallel.allreduce_sum_numpy(target)
            target /= self.numTrials
        elif self.stackingDim == 0:
            target = parallel.gather_trials(target)
        out.data = target

    @abstractmethod
    def process_metadata(self, data, out):
        pass


def propagate_properties(in_data, out_data, keeptrials=True, time_axis=False):
    """Channels / trialdefinition / samplerate of the output object
    (rules of computational_routine.py:1114-1231 for the AnalogData -> Spectral /
    CrossSpectral cases the hot path needs)."""
    from ..datatype import CrossSpectralData, SpectralData, selected_trialdefinition
    chans = selected_channels(in_data)
    names = np.array(in_data.channel) if chans is None else np.array(in_data.channel)[chans]
    if isinstance(out_data, SpectralData):
        out_data.channel = names
    elif isinstance(out_data, CrossSpectralData):
        out_data.channel_i = names
        out_data.channel_j = names.copy()
    trl = selected_trialdefinition(in_data)
    if not time_axis:
        if keeptrials:
            for row in range(trl.shape[0]):
                trl[row, :2] = [row, row + 1]
            out_data.trialdefinition = trl
        else:
            out_data.trialdefinition = np.array([[0, 1, 0]])
    else:
        out_data.trialdefinition = trl if keeptrials else trl[0, :][None, :]
    out_data.samplerate = in_data.samplerate
