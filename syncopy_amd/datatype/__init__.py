"""In-memory / in-HBM stand-ins for the Syncopy data classes the hot path touches.

Only the *layout contract* of the reference is kept:
 - AnalogData: one (time x channel) float32 C-order matrix, channel fastest
   (syncopy/datatype/continuous_data.py:405), trial `t` = rows
   [trialdefinition[t,0], trialdefinition[t,1]) with trigger offset trialdefinition[t,2];
 - SpectralData: dimord ["time","taper","freq","channel"] (continuous_data.py:550);
 - CrossSpectralData: dimord ["time","freq","channel_i","channel_j"] (continuous_data.py:721);
 - in-place selections {"trials","channel","latency"} (datatype/selector.py:126-454).
HDF5 storage, Dask and the .spy container format are bypassed: the trial matrix is
uploaded once to HBM (`AnalogData.device_data`) and stays there - the in-HBM trial queue.
"""
import numpy as np

from ..shared.errors import SPYTypeError, SPYValueError
from ..shared.tools import best_match


class _Base:
    _defaultDimord = None

    def __init__(self, data=None, samplerate=None, trialdefinition=None, dimord=None):
        self.dimord = list(dimord) if dimord is not None else list(self._defaultDimord)
        self.samplerate = None if samplerate is None else float(samplerate)
        self._device = None
        self._device_key = None
        self._data = None
        self._pending = None          # (thunk, shape, dtype): host array produced on first access (results that
                                      # may never leave the device, e.g. the CSD between the ST and the AV stage)
        self.data = data
        self._trialdefinition = None
        if trialdefinition is not None:
            self.trialdefinition = trialdefinition
        self.selection = None
        self.cfg = {}
        self.info = {}
        self.log = ""

    @property
    def _stackingDim(self):
        return 0

    @property
    def data(self):
        if self._data is None and self._pending is not None:
            thunk, self._pending = self._pending[0], None
            self._data = thunk()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value
        self._pending = None
        self.invalidate()

    def invalidate(self):
        """Forget device copies of `.data`.  Assigning `.data` does this by itself; after editing the host array IN
        PLACE (`obj.data[...] = ...`) call it explicitly - in-place edits cannot be seen from here."""
        up = getattr(self, "_upload", None)
        if up is not None:
            up.finish()                   # (never drop a tensor a copy thread still writes into)
        self._upload = None
        self._device = None
        self._device_key = None
        self._looks = {}                  # what precision="auto" measured on this array (CrossSpectra.needs_float64)

    def set_pending(self, thunk, shape, dtype):
        """The host array is `thunk()` - evaluated only if somebody reads `.data`."""
        self._data = None
        self._pending = (thunk, tuple(shape), np.dtype(dtype))

    @property
    def data_shape(self):
        return self._pending[1] if (self._data is None and self._pending is not None) else self.data.shape

    @property
    def data_dtype(self):
        return self._pending[2] if (self._data is None and self._pending is not None) else self.data.dtype

    @property
    def trialdefinition(self):
        return self._trialdefinition

    @trialdefinition.setter
    def trialdefinition(self, trl):
        trl = np.array(trl, dtype=float)
        if trl.ndim != 2 or trl.shape[1] < 3:
            raise SPYValueError("M x 3 array [start, stop, offset]", varname="trialdefinition", actual=f"{trl.shape}")
        self._trialdefinition = trl

    @property
    def sampleinfo(self):
        return None if self._trialdefinition is None else self._trialdefinition[:, :2].astype(np.int64)

    @property
    def trialintervals(self):
        si, off = self.sampleinfo, self._trialdefinition[:, 2]
        n = si[:, 1] - si[:, 0]
        return np.stack([off, off + n - 1], axis=1) / self.samplerate

    @property
    def trials(self):
        ax = self.dimord.index("time")
        out = []
        for a, b in self.sampleinfo:
            idx = [slice(None)] * self.data.ndim
            idx[ax] = slice(int(a), int(b))
            out.append(self.data[tuple(idx)])
        return out

    @property
    def time(self):
        """Per-trial time axes (n + offset) / samplerate (datatype/util.py:80-83)."""
        return [(np.arange(0, int(b - a)) + off) / self.samplerate
                for (a, b), off in zip(self.sampleinfo, self._trialdefinition[:, 2])]


class AnalogData(_Base):
    _defaultDimord = ["time", "channel"]

    def __init__(self, data=None, samplerate=None, trialdefinition=None, channel=None, dimord=None):
        if isinstance(data, (list, tuple)):
            shapes = {np.shape(t) for t in data}
            if len(shapes) != 1:
                raise SPYValueError("NumPy arrays of identical shape", varname="data",
                                    actual="NumPy arrays with mismatching shapes")
            n = data[0].shape[0]
            if trialdefinition is None:
                starts = np.arange(len(data)) * n
                trialdefinition = np.stack([starts, starts + n, np.zeros(len(data))], axis=1)
            data = np.concatenate([np.asarray(t) for t in data], axis=0)
        super().__init__(None, samplerate, None, dimord)
        if data is not None:
            data = np.asarray(data)
            if data.ndim != 2:
                raise SPYValueError("2-dimensional (time x channel) array", varname="data", actual=f"{data.ndim}d")
            self.data = data
            if trialdefinition is None:
                n = data.shape[self.dimord.index("time")]
                trialdefinition = np.array([[0, n, 0]])
            self.trialdefinition = trialdefinition
        nchan = 0 if self.data is None else self.data.shape[self.dimord.index("channel")]
        if channel is None:
            channel = ["channel" + str(i + 1).zfill(len(str(nchan))) for i in range(nchan)]
        self.channel = np.array(channel)

    def shard_span(self):
        """[lo, hi) of the host rows this rank's trials live in: the whole recording without a process group, else the
        rows from the first to the last sample of the rank's contiguous trial shard (parallel.my_shard over the selected
        trials, exactly the shard every compute_hip works on) - a worker of the reference reads only its own slab too
        (shared/kwarg_decorators.py:684-735).  Trials that overlap or are listed out of order make the span cover rows
        nobody on this rank needs; nothing outside it is ever staged."""
        from .. import parallel
        ntime = self.data.shape[self.dimord.index("time")]
        if parallel.world()[1] == 1:
            return 0, int(ntime)
        rows = trial_rows(self)
        lo, hi = parallel.my_shard(len(rows))
        mine = rows[lo:hi]
        if not mine:
            return 0, 0
        return int(min(a for a, _ in mine)), int(min(ntime, max(b for _, b in mine)))

    def device_data(self, device=None, partial=False):
        """This rank's part of the (time x channel) float32 matrix in HBM (uploaded once, C-order, channel fastest): the
        rows `shard_span()`; `device_rows(data)` gives the trials in ITS row coordinates.
        `partial=True`: do not wait for an upload in flight - the caller walks the trials in order and asks
        `upload_in_flight().wait_rows(row_end)` before it touches rows (backend.Upload): kernels on the first trials
        overlap the PCIe copy of the later ones."""
        import torch
        from .. import backend
        backend.require_gpu()              # loud failure: there is no CPU path
        dev = torch.device("cuda" if device is None else device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        span = self.shard_span()
        # the copy belongs to one host array in one orientation and one row span: a new array object, shape, dimord or
        # shard uploads again
        key = (id(self._data), self._data.shape, tuple(self.dimord), str(dev), span)
        if self._device is None or self._device_key != key:
            if getattr(self, "_upload", None) is not None:
                self._upload.finish()      # a copy thread still writing into the tensor that is about to be dropped
                self._upload = None
            self._device, self._upload = backend.to_device(self.data, dev, time_axis=self.dimord.index("time"),
                                                           background=True, rows=span)
            self._device_key = key
            self._row_origin = span[0]
            self.staged_rows = span        # (what tests and tests/nccl_worker.py report)
        if not partial and getattr(self, "_upload", None) is not None:
            self._upload.finish()
            self._upload = None
        return self._device

    def upload_in_flight(self):
        """backend.Upload of a device copy that is still being filled, or None."""
        up = getattr(self, "_upload", None)
        if up is not None and up.complete:
            up.finish()
            self._upload = up = None
        return up

    def selectdata(self, select=None):
        self.selection = None if select is None else Selection(self, select)
        return self


class SpectralData(_Base):
    _defaultDimord = ["time", "taper", "freq", "channel"]

    def __init__(self, data=None, samplerate=None, trialdefinition=None, dimord=None):
        super().__init__(data, samplerate, trialdefinition, dimord)
        self.freq = None
        self.taper = None
        self.channel = None

    def selectdata(self, select=None):
        self.selection = None if select is None else Selection(self, select)
        return self


class CrossSpectralData(_Base):
    _defaultDimord = ["time", "freq", "channel_i", "channel_j"]

    def __init__(self, data=None, samplerate=None, trialdefinition=None, dimord=None):
        super().__init__(data, samplerate, trialdefinition, dimord)
        self.freq = None
        self.channel_i = None
        self.channel_j = None
        self._acc_raw = None          # raw lower-triangle accumulator still on the device (+ its scale), see
        self._acc_scale = None        # connectivity/ST_compRoutines.py: CrossSpectra.compute_hip
        self._dev_thunk = None
        self._dev_value = None

    @property
    def _dev(self):
        """Device copy of `.data` if there is one (finalised on first access)."""
        if self._dev_value is None and self._dev_thunk is not None:
            self._dev_value = self._dev_thunk()
        return self._dev_value

    @_dev.setter
    def _dev(self, value):
        self._dev_value = value
        self._dev_thunk = None


class FauxTrial:
    """Shape/dtype stand-in of a trial for the dry run (datatype/base_data.py:1458)."""

    def __init__(self, shape, dtype):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    @property
    def T(self):
        return FauxTrial(self.shape[::-1], self.dtype)


class Selection:
    """Resolved in-place selection: which rows / columns form output trial k.

    trials : any order, repeats allowed, kept as given (selector.py:217-250)
    channel: indices, names or slice
    latency: [t0, t1] -> inclusive sample window best_match(time[trial], [t0,t1], span=True)
             (selector.py:958-972), or 'maxperiod' / 'minperiod' / 'prestim' / 'poststim'
    """

    def __init__(self, data, select):
        if not isinstance(select, dict):
            raise SPYTypeError(select, "select", "dict")
        unknown = set(select) - {"trials", "channel", "latency"}
        if unknown:
            raise SPYValueError("keys 'trials', 'channel', 'latency'", varname="select", actual=str(sorted(unknown)))
        ntr = data.trialdefinition.shape[0]
        tr = select.get("trials")
        if tr is None or (isinstance(tr, str) and tr == "all"):
            self.trial_ids = list(range(ntr))
        else:
            self.trial_ids = [int(t) for t in np.atleast_1d(tr)]
            if any(t < 0 or t >= ntr for t in self.trial_ids):
                raise SPYValueError(f"trial indices in [0, {ntr})", varname="select: trials", actual=str(tr))
        ch = select.get("channel")
        names = list(getattr(data, "channel", []))
        if ch is None or (isinstance(ch, str) and ch == "all"):
            self.channel = list(range(len(names)))
        elif isinstance(ch, slice):
            self.channel = list(range(len(names)))[ch]
        else:
            idx = []
            for c in np.atleast_1d(ch):
                if isinstance(c, (str, np.str_)):
                    if c not in names:
                        raise SPYValueError("existing channel names", varname="select: channel", actual=str(c))
                    idx.append(names.index(c))
                else:
                    idx.append(int(c))
            if any(c < 0 or c >= len(names) for c in idx):
                raise SPYValueError(f"channel indices in [0, {len(names)})", varname="select: channel", actual=str(ch))
            self.channel = idx
        self.time = {}     # trial id -> (start, stop) relative to the trial
        lat = select.get("latency")
        si = data.sampleinfo
        times = data.time
        if isinstance(lat, str):
            iv = data.trialintervals[self.trial_ids]
            if lat == "maxperiod":
                lat = [iv[:, 0].min(), iv[:, 1].max()]
            elif lat == "minperiod":
                lat = [iv[:, 0].max(), iv[:, 1].min()]
            elif lat == "prestim":
                lat = [iv[:, 0].min(), 0.0]
            elif lat == "poststim":
                lat = [0.0, iv[:, 1].max()]
            else:
                raise SPYValueError("'maxperiod', 'minperiod', 'prestim', 'poststim' or [t0, t1]", "latency", lat)
        for t in set(self.trial_ids):
            n = int(si[t, 1] - si[t, 0])
            if lat is None:
                self.time[t] = (0, n)
            else:
                _, sel = best_match(times[t], lat, span=True)
                self.time[t] = (int(sel[0]), int(sel[-1]) + 1) if sel.size else (0, 0)
        trl = data.trialdefinition
        out = np.zeros((len(self.trial_ids), trl.shape[1]))
        counter = 0
        for k, t in enumerate(self.trial_ids):
            a, b = self.time[t]
            out[k, :3] = [counter, counter + (b - a), a + trl[t, 2] if b > a else 0]
            out[k, 3:] = trl[t, 3:]
            counter += b - a
        self.trialdefinition = out

    def rows(self, k):
        """Absolute [start, stop) rows of output trial k in the data matrix (needs the parent's sampleinfo)."""
        raise NotImplementedError


def trial_rows(data):
    """[(start, stop)] absolute row ranges, in output-trial order, honouring an active selection."""
    si = data.sampleinfo
    if data.selection is None:
        return list(zip(si[:, 0].astype(np.int64).tolist(), si[:, 1].astype(np.int64).tolist()))
    out = []
    for t in data.selection.trial_ids:
        a, b = data.selection.time[t]
        out.append((int(si[t, 0] + a), int(si[t, 0] + b)))
    return out


def device_rows(data):
    """trial_rows in the row coordinates of `data.device_data()` (which must have been called: it decides the span)."""
    o = getattr(data, "_row_origin", 0) or 0
    return [(a - o, b - o) for a, b in trial_rows(data)] if o else trial_rows(data)


def selected_channels(data):
    return None if data.selection is None else list(data.selection.channel)


def selected_trialdefinition(data):
    return data.trialdefinition.copy() if data.selection is None else data.selection.trialdefinition.copy()
