"""NumPy + ctypes host of libspyhip.so - no PyTorch anywhere (the reference depends on NumPy/SciPy only,
pyproject.toml:28-29).  What a Syncopy maintainer would put behind `ComputationalRoutine.compute_hip`
(INTEGRATION.md): the library allocates HBM, holds the trial queue, runs the kernels, sums over ranks with RCCL and
hands back NumPy arrays.

    dev = abi.Device(0)
    q = dev.upload_trials(data, sampleinfo)                    # (rows x channels) float32 + trial row ranges -> HBM
    pow_ = dev.mtmfft(q, tapers, scale, nfft, output="pow", keeptapers=False)
    coh = dev.coherence(q, tapers, scale, nfft, output="abs")

The PyTorch-backed front ends (`spy.freqanalysis`, `spy.connectivityanalysis`) call the very same entry points with
tensor pointers; nothing here is a second implementation of any arithmetic.

A process that really is NumPy-only sets SPY_NO_TORCH=1 before the first use: otherwise `_lib.load()` imports an
installed torch FIRST, so that a later `import torch` in the same process does not bind to the system libamdhip64 this
library brings in (same SONAME as the runtime bundled with the torch wheel).

`reference_mean` (the per-channel mean in the reference's float32 summation order, K0) reproduces NumPy bit for bit
for dimord time x channel; for channel x time data the reference averages a transposed view with pairwise summation
and the result agrees to float32 rounding only.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SpyHipError, check  # noqa: F401

OUTPUT_KIND = {"pow": 0, "abs": 1, "fourier": 2, "complex": 2, "real": 3, "imag": 4, "angle": 5,
               "absreal": 6, "absimag": 7}
DETREND = {None: -1, False: -1, 0: 0, 1: 1}


class Buffer:
    """A block of HBM owned by the library (spyhip_alloc / spyhip_free)."""

    def __init__(self, dev, shape, dtype, zero=False):
        self.dev, self.shape, self.dtype = dev, tuple(int(s) for s in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(dev.lib.spyhip_alloc(dev.handle, self.nbytes, C.byref(p)), "spyhip_alloc")
        self.ptr = p
        if zero:
            check(dev.lib.spyhip_memset(dev.handle, self.ptr, 0, self.nbytes), "spyhip_memset")

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(self.dev.lib.spyhip_download(self.dev.handle, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes),
              "spyhip_download")
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            self.dev.lib.spyhip_free(self.dev.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TrialQueue:
    """spyhip_queue: the (rows x channels) float32 matrix of an AnalogData object in HBM with its trials' row
    ranges (`sampleinfo`, datatype/base_data.py:993) - uploaded once, indexed exactly."""

    def __init__(self, dev, data, sampleinfo):
        data = np.ascontiguousarray(data, dtype=np.float32)
        si = np.ascontiguousarray(sampleinfo, dtype=np.int64)
        if data.ndim != 2 or si.ndim != 2 or si.shape[1] != 2:
            raise ValueError("data must be (rows, channels), sampleinfo (trials, 2)")
        self.dev = dev
        self.nrows, self.nchan = data.shape
        self.ntrials = si.shape[0]
        self.lengths = (si[:, 1] - si[:, 0]).astype(np.int64)
        h = C.c_void_p()
        check(dev.lib.spyhip_queue_upload(dev.handle, data.ctypes.data_as(C.c_void_p), self.nrows, self.nchan,
                                          si.ctypes.data_as(_lib.c_i64p), self.ntrials, C.byref(h)),
              "spyhip_queue_upload")
        self.handle = h
        self.data_d = C.c_void_p(dev.lib.spyhip_queue_data(h))
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_int()
        check(dev.lib.spyhip_queue_segments(h, C.byref(a), C.byref(b), C.byref(n)), "spyhip_queue_segments")
        self.start_d, self.stop_d = a, b

    def free(self):
        if self.handle is not None:
            self.dev.lib.spyhip_queue_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Device:
    """One spyhip_ctx; every call is complete when it returns a NumPy array."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.spyhip_ctx_create(int(device), C.byref(h)), "spyhip_ctx_create")
        self.handle = h
        self.rank, self.nranks = 0, 1

    # ---- data
    def upload_trials(self, data, sampleinfo):
        return TrialQueue(self, data, sampleinfo)

    def synchronize(self):
        check(self.lib.spyhip_ctx_synchronize(self.handle), "spyhip_ctx_synchronize")

    # ---- ranks (RCCL inside the library; the 128-byte id travels by whatever means the host has)
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        check(self.lib.spyhip_comm_unique_id(buf), "spyhip_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        check(self.lib.spyhip_comm_init(self.handle, C.c_char_p(unique_id), int(rank), int(nranks)), "spyhip_comm_init")
        self.rank, self.nranks = int(rank), int(nranks)

    def comm_destroy(self):
        check(self.lib.spyhip_comm_destroy(self.handle), "spyhip_comm_destroy")
        self.rank, self.nranks = 0, 1

    @property
    def has_comm(self):
        return self.lib.spyhip_comm_info(self.handle, None, None) == 0

    # ---- K1
    def _plan(self, nsig, nfft, nchan, tapers, scale, detrend, demean_taper, freq_idx, output, keeptapers):
        tapers = np.ascontiguousarray(np.atleast_2d(tapers), dtype=np.float64)
        assert tapers.shape[1] == nsig
        fi = None if freq_idx is None else np.ascontiguousarray(freq_idx, dtype=np.int32)
        nfsel = nfft // 2 + 1 if fi is None else int(fi.size)
        h = C.c_void_p()
        check(self.lib.spyhip_fft_plan_create(
            self.handle, int(nsig), int(nfft), int(nchan), tapers.shape[0], tapers.ctypes.data_as(_lib.c_f64p),
            float(scale), DETREND[detrend], int(bool(demean_taper)),
            None if fi is None else fi.ctypes.data_as(_lib.c_i32p), nfsel, OUTPUT_KIND[output], int(bool(keeptapers)),
            C.byref(h)), "spyhip_fft_plan_create")
        if DETREND[detrend] == 0:      # whole trials: the reference's float32 row-order mean (compRoutines.py:169-170)
            check(self.lib.spyhip_fft_plan_set_reference_mean(h, 1), "spyhip_fft_plan_set_reference_mean")
        return h, tapers.shape[0], nfsel

    def mtmfft(self, queue, tapers, scale, nfft=None, detrend=0, demean_taper=False, freq_idx=None, output="pow",
               keeptapers=False, device_result=False):
        """mtmfft_cF for all trials of the queue (equal length): (T, Kout, F', C) float32 / complex64."""
        nsig = int(queue.lengths[0])
        if np.any(queue.lengths != nsig):
            raise ValueError("one call serves trials of one length (group them by length)")
        nfft = nsig if nfft is None else int(nfft)
        plan, K, nfsel = self._plan(nsig, nfft, queue.nchan, tapers, scale, detrend, demean_taper, freq_idx, output,
                                    keeptapers)
        try:
            kout = K if keeptapers else 1
            out = Buffer(self, (queue.ntrials, kout, nfsel, queue.nchan),
                         np.complex64 if OUTPUT_KIND[output] == 2 else np.float32)
            check(self.lib.spyhip_fft_exec(plan, queue.data_d, queue.nchan, None, queue.start_d, queue.start_d,
                                           queue.stop_d, queue.ntrials, out.ptr), "spyhip_fft_exec")
            if device_result:
                self.synchronize()
                return out
            res = out.numpy()
            out.free()
            return res
        finally:
            self.synchronize()
            self.lib.spyhip_fft_plan_destroy(plan)

    # ---- K1 + K4 (+ C1) + K5
    def csd_accumulator(self, queue, tapers, scale, nfft=None, detrend=0, demean_taper=False, batch_bytes=8 << 30):
        """Raw lower-triangle accumulator sum_t sum_k X X^H of this rank's trials, summed over ranks if a
        communicator exists: (Buffer (F, C, C) complex64, number of tapers)."""
        spec = self.mtmfft(queue, tapers, scale, nfft, detrend, demean_taper, None, "fourier", True,
                           device_result=True)
        T, K, F, Cn = spec.shape
        acc = Buffer(self, (F, Cn, Cn), np.complex64, zero=True)
        check(self.lib.spyhip_csd_accumulate(self.handle, spec.ptr, T * K, F, Cn, acc.ptr), "spyhip_csd_accumulate")
        self.synchronize()
        spec.free()
        if self.has_comm:
            check(self.lib.spyhip_allreduce_csd(self.handle, acc.ptr, F, Cn), "spyhip_allreduce_csd")
        return acc, K

    def csd(self, queue, tapers, scale, nfft=None, detrend=0, demean_taper=False, ntrials_total=None):
        """Trial-averaged cross-spectral density (F, C, C) complex64 (cross_spectra_cF + trial mean)."""
        acc, K = self.csd_accumulator(queue, tapers, scale, nfft, detrend, demean_taper)
        T = queue.ntrials if ntrials_total is None else int(ntrials_total)
        F, Cn, _ = acc.shape
        check(self.lib.spyhip_csd_finalize(self.handle, acc.ptr, F, Cn, 1.0 / (K * T)), "spyhip_csd_finalize")
        res = acc.numpy()
        acc.free()
        return res

    def coherence(self, queue, tapers, scale, nfft=None, detrend=0, output="abs", ntrials_total=None):
        """connectivityanalysis(method='coh'): (F, C, C) float32 (complex64 for output='complex')."""
        acc, K = self.csd_accumulator(queue, tapers, scale, nfft, detrend, False)
        T = queue.ntrials if ntrials_total is None else int(ntrials_total)
        F, Cn, _ = acc.shape
        kind = OUTPUT_KIND[output]
        out = Buffer(self, (F, Cn, Cn), np.complex64 if kind == 2 else np.float32)
        check(self.lib.spyhip_coh_from_accumulator(self.handle, acc.ptr, F, Cn, 1.0 / (K * T), kind, out.ptr),
              "spyhip_coh_from_accumulator")
        res = out.numpy()
        out.free()
        acc.free()
        return res

    def close(self):
        if self.handle is not None:
            self.lib.spyhip_ctx_destroy(self.handle)
        self.handle = None
