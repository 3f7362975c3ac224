"""Thin Python layer over the C ABI of libspyhip.so (include/spyhip.h).

PyTorch is used for device memory and streams only: tensors are allocated by
torch, their raw device pointers are handed to the HIP library, and all kernels
are enqueued on torch's current stream.  There is no CPU code path here.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import SpyHipError, check

OUTPUT_KIND = {"pow": 0, "abs": 1, "fourier": 2, "complex": 2, "real": 3, "imag": 4, "angle": 5,
               "absreal": 6, "absimag": 7}
DETREND = {None: -1, False: -1, 0: 0, 1: 1}

_contexts = {}
_NP_DTYPE = {torch.float32: np.float32, torch.float64: np.float64, torch.complex64: np.complex64,
             torch.complex128: np.complex128, torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8}


def require_gpu():
    if not torch.cuda.is_available():
        raise SpyHipError("no HIP device visible: syncopy_amd has no CPU fallback (torch.cuda.is_available() is False)")


class Context:
    """One spyhip_ctx per device, bound to torch's current stream at every call."""

    def __init__(self, device):
        require_gpu()
        self.lib = _lib.load()
        self.device = int(device)
        h = C.c_void_p()
        check(self.lib.spyhip_ctx_create(self.device, C.byref(h)), "spyhip_ctx_create")
        self.handle = h

    def bind_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.spyhip_ctx_set_stream(self.handle, C.c_void_p(s)), "spyhip_ctx_set_stream")

    def synchronize(self):
        check(self.lib.spyhip_ctx_synchronize(self.handle), "spyhip_ctx_synchronize")


def context(device=None):
    require_gpu()
    if device is None:
        device = torch.cuda.current_device()
    device = torch.device(device).index if not isinstance(device, int) else device
    if device is None:
        device = torch.cuda.current_device()
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


_PIN_BYTES = 256 << 20
_D2H_CHUNK = 64 << 20
_pin = {}
_copy_pool = None


_COPY_THREADS = 8


def _pool():
    global _copy_pool
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _copy_pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="spyhip-copy")
    return _copy_pool


def _host_copy(dst, src):
    """dst[:] = src for large byte arrays, split over the copy threads (NumPy releases the GIL in its copy loops): one
    thread moves ~10 GB/s into freshly allocated pageable memory (page faults included), which made the drain of the
    pinned staging buffer - not the bus - the slow half of a large result's way to the host."""
    n = dst.shape[0]
    if n < (8 << 20):
        dst[:] = src
        return
    pool = _pool()
    step = (n + _COPY_THREADS - 1) // _COPY_THREADS
    futs = [pool.submit(np.copyto, dst[o:o + step], src[o:o + step]) for o in range(0, n, step)]
    for f in futs:
        f.result()


_RESULT_PIN_CAP = 8 << 30       # page-locked bytes results may hold (live arrays + free blocks); beyond: the staged path
_RESULT_PIN_GRAIN = 64 << 20
_result_free = {}               # rounded size -> [pinned uint8 tensors]
_result_bytes = [0]             # page-locked bytes of the pool, live and free
import threading as _threading_mod
_result_lock = _threading_mod.Lock()
_result_zombies = []            # (copy-finished event, block) of HostLandings dropped unread while their copy was under way


def _result_block(nbytes):
    """A page-locked block for a large result, from the free list or fresh (cudaHostAlloc of 0.5 GB costs ~0.1 s once; the
    block returns to the list when the array the caller got - and every view of it - is gone).  None when the pool is at
    its cap."""
    size = (nbytes + _RESULT_PIN_GRAIN - 1) // _RESULT_PIN_GRAIN * _RESULT_PIN_GRAIN
    with _result_lock:
        # blocks of landings nobody read, whose copy has finished meanwhile
        for item in [z for z in _result_zombies if z[0].query()]:
            _result_zombies.remove(item)
            _result_free.setdefault(item[1].numel(), []).append(item[1])
        lst = _result_free.get(size)
        if lst:
            return lst.pop()
        if _result_bytes[0] + size > _RESULT_PIN_CAP:
            # free blocks of other sizes are the first to go
            for k in list(_result_free):
                while _result_free[k] and _result_bytes[0] + size > _RESULT_PIN_CAP:
                    _result_free[k].pop()
                    _result_bytes[0] -= k
            if _result_bytes[0] + size > _RESULT_PIN_CAP:
                return None
        _result_bytes[0] += size
    return torch.empty(size, dtype=torch.uint8, pin_memory=True)


def _result_release(block):
    with _result_lock:
        _result_free.setdefault(block.numel(), []).append(block)


_prewarm_threads = []


def prewarm_landing(nbytes):
    """Make sure the result pool holds a free page-locked block for a result of `nbytes` - in a background thread: the
    first analysis of a process page-locks 0.54 GB for its coherence result (~8 ms), which this moves under the upload
    and the transforms instead of in front of the first host copy.  No-op when a free block of that size exists."""
    import threading
    size = (nbytes + _RESULT_PIN_GRAIN - 1) // _RESULT_PIN_GRAIN * _RESULT_PIN_GRAIN
    with _result_lock:
        if _result_free.get(size):
            return
    dev = torch.cuda.current_device()

    def work():
        torch.cuda.set_device(dev)
        block = _result_block(nbytes)
        if block is not None:
            _result_release(block)

    th = threading.Thread(target=work, name="spyhip-prewarm", daemon=True)
    th.start()
    _prewarm_threads[:] = [t for t in _prewarm_threads if t.is_alive()] + [th]


def _prewarm_join():
    for th in _prewarm_threads:
        th.join()
    del _prewarm_threads[:]


def to_host(t):
    """Device tensor -> NumPy array.  Large results land in ONE asynchronous copy in a page-locked block of the result pool
    (the bus gives ~57 GB/s to pinned memory, ~6 GB/s to pageable memory) and the array handed out IS that block: no
    second pass through host memory (rounds 2-5 staged through a pinned pair and copied out into pageable memory, 26 GB/s
    end to end for the 0.54 GB of a 256-channel coherence).  The block goes back to the pool when the array and all its
    views are garbage; the pool holds at most 8 GiB, beyond which results take the staged path; small results take the
    plain path."""
    nbytes = t.numel() * t.element_size()
    if not t.is_cuda or nbytes < (8 << 20):
        return t.cpu().numpy()
    t = t.contiguous()
    flat = t.view(-1).view(torch.uint8) if not t.is_complex() else torch.view_as_real(t).view(-1).view(torch.uint8)
    block = _result_block(nbytes)
    if block is not None:
        import weakref
        block[:nbytes].copy_(flat, non_blocking=True)
        root = block.numpy()                         # every view handed out keeps `root` alive through .base
        weakref.finalize(root, _result_release, block)
        out = root[:nbytes].view(_NP_DTYPE[t.dtype]).reshape(tuple(t.shape))
        torch.cuda.current_stream(t.device).synchronize()
        return out
    out = np.empty(t.shape, dtype=_NP_DTYPE[t.dtype])
    dst = out.reshape(-1).view(np.uint8)
    stage = _staging()
    stream = torch.cuda.current_stream(t.device)
    pending = None                                   # (event, staging buffer, offset, length) of the copy in flight
    for k, off in enumerate(range(0, nbytes, _D2H_CHUNK)):
        n = min(_D2H_CHUNK, nbytes - off)
        buf = stage[k & 1]
        buf[:n].copy_(flat[off:off + n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        if pending is not None:                      # drain the previous chunk while this one is on the bus
            pending[0].synchronize()
            _host_copy(dst[pending[2]:pending[2] + pending[3]], pending[1][:pending[3]].numpy())
        pending = (ev, buf, off, n)
    pending[0].synchronize()
    _host_copy(dst[pending[2]:pending[2] + pending[3]], pending[1][:pending[3]].numpy())
    return out


def _staging(name="buf", nbytes=_PIN_BYTES):
    stage = _pin.get(name)
    if stage is None:
        stage = _pin[name] = (torch.empty(nbytes, dtype=torch.uint8, pin_memory=True),
                              torch.empty(nbytes, dtype=torch.uint8, pin_memory=True))
    return stage


_H2D_CHUNK = 128 << 20          # bytes per staging buffer of the ingress (two of them alternate; its own pair: a result
                                # may be on its way to the host through to_host's buffers at the same time)


def _staged_fill(view, src):
    """view[...] = src (2-D, float32 view of a pinned buffer; src any dtype / orientation), rows split over the copy
    threads: one thread converts / copies ~10 GB/s, the bus takes 57."""
    n = view.shape[0]
    if view.nbytes < (8 << 20):
        np.copyto(view, src, casting="unsafe")
        return
    pool = _pool()
    step = (n + _COPY_THREADS - 1) // _COPY_THREADS
    futs = [pool.submit(np.copyto, view[o:o + step], src[o:o + step], "unsafe") for o in range(0, n, step)]
    for f in futs:
        f.result()


import threading as _threading
_upload_lock = _threading.Lock()        # the pinned staging pair of the ingress serves one upload at a time


class Upload:
    """A recording on its way into the in-HBM trial queue (the reference streams trial by
    trial, computational_routine.py:1001-1032).  A background thread fills two alternating pinned staging buffers
    (rows split over a thread pool, converted to float32 / transposed while staging) and enqueues the copies on a
    stream of its own; after every chunk it leaves a mark (last row, event).  Consumers that walk the trials in order
    call `wait_rows(row_end)` before they launch work on rows < row_end: the calling thread waits until that chunk has
    been ENQUEUED, then torch's current stream waits for its event - kernels on the first trials run while the later
    ones are still on the bus.  `finish()` waits for everything (what `AnalogData.device_data()` does for callers that
    do not know about uploads in flight)."""

    def __init__(self, host, dev, time_axis, row0=0):
        import threading
        self.dev, self.host, self.time_axis = dev, host, time_axis
        self.row0 = int(row0)                # host row of the tensor's first row (a rank stages only its own span)
        self.nrows = dev.shape[0]
        # `dev` was allocated on the caller's current stream: the caching allocator may have handed out a block whose
        # previous owner still has kernels queued there - the copy stream must not overtake them
        self.alloc_stream = torch.cuda.current_stream(dev.device)
        self.marks = []                      # [(row_end, event)] in row order
        self.cond = threading.Condition()
        self.error = None
        self.complete = False
        self.thread = threading.Thread(target=self._run, name="spyhip-upload", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            with _upload_lock:
                self._copy()
        except BaseException as exc:                       # noqa: BLE001 - handed to the waiting thread
            self.error = exc
        finally:
            with self.cond:
                self.complete = True
                self.cond.notify_all()

    def _copy(self):
        dev, host = self.dev, self.host
        torch.cuda.set_device(dev.device)
        ntime, nchan = dev.shape
        stage = _staging("h2d", _H2D_CHUNK)
        rows_per = self.chunk_rows()
        stream = torch.cuda.Stream(device=dev.device)
        stream.wait_stream(self.alloc_stream)
        dev.record_stream(stream)
        busy = [None, None]
        h0 = self.row0
        for k, r0 in enumerate(range(0, ntime, rows_per)):
            r1 = min(ntime, r0 + rows_per)
            buf = stage[k & 1]
            if busy[k & 1] is not None:
                busy[k & 1].synchronize()              # the copy that last used this buffer has left it
            pinned = buf[:(r1 - r0) * nchan * 4].view(torch.float32).view(r1 - r0, nchan)
            _staged_fill(pinned.numpy(), host[h0 + r0:h0 + r1] if self.time_axis == 0 else host[:, h0 + r0:h0 + r1].T)
            with torch.cuda.stream(stream):
                dev[r0:r1].copy_(pinned, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
            busy[k & 1] = ev
            with self.cond:
                self.marks.append((r1, ev))
                self.cond.notify_all()
        for ev in busy:
            if ev is not None:
                ev.synchronize()

    def chunk_rows(self):
        return max(1, _H2D_CHUNK // (4 * max(self.dev.shape[1], 1)))

    def wait_rows(self, row_end, stream=None):
        """Rows [0, row_end) are in HBM as far as `stream` (default: torch's current stream) is concerned."""
        row_end = min(int(row_end), self.nrows)
        with self.cond:
            while not self.complete and not (self.marks and self.marks[-1][0] >= row_end):
                self.cond.wait()
            if self.error is not None:
                raise self.error
            ev = next((e for r, e in self.marks if r >= row_end), self.marks[-1][1] if self.marks else None)
        if ev is not None:
            (stream or torch.cuda.current_stream(self.dev.device)).wait_event(ev)

    def finish(self):
        self.thread.join()
        if self.error is not None:
            raise self.error
        self.wait_rows(self.nrows)
        self.host = None


def to_device(host, device, time_axis=0, background=False, rows=None):
    """Host recording -> (time x channel) float32 matrix in HBM: the ingress of the in-HBM trial queue.
    `rows` = (lo, hi): only that span of the recording's rows (a rank's trial shard, AnalogData.shard_span) - the
    tensor's row 0 is then host row lo.
    `host` may be an np.memmap onto a `.spy` data file (io/spy_container.py) of any size: blocks of rows are read
    (and, if needed, converted to float32 / transposed - dimord ["channel", "time"], compRoutines.py:143-146) straight
    into two alternating pinned staging buffers and copied to the device asynchronously, so the file is never held in
    host memory as a whole and the disk read of block k+1 overlaps the PCIe copy of block k.
    `background=True` returns (tensor, Upload or None) at once: the copy proceeds on a thread and a stream of its own
    (class Upload) - None when the recording is small enough for one plain copy."""
    ntime, nchan = (host.shape if time_axis == 0 else host.shape[::-1])
    lo, hi = (0, ntime) if rows is None else (int(rows[0]), int(rows[1]))
    assert 0 <= lo <= hi <= ntime, (lo, hi, ntime)
    dev = torch.empty((hi - lo, nchan), dtype=torch.float32, device=device)
    nbytes = (hi - lo) * nchan * 4
    if nbytes == 0:
        return (dev, None) if background else dev
    direct = (time_axis == 0 and host.dtype == np.float32 and host.flags["C_CONTIGUOUS"]
              and not isinstance(host, np.memmap))
    if direct and nbytes < (64 << 20):
        dev.copy_(torch.from_numpy(host[lo:hi]))
        return (dev, None) if background else dev
    up = Upload(host, dev, time_axis, row0=lo)
    if background:
        return dev, up
    up.finish()
    return dev


# reusable (rows, F, C) spectra buffers of the coherence path, one per (shape, device): the FFT -> CSD hand-over
# never leaves the device, and re-allocating tens of GB per call costs more than the kernels
_handover = {}


def release_buffers():
    """Give back what the library keeps alive between calls: the spectra hand-over buffer (tens of GB at production
    shapes - kept because re-allocating it per call costs more than the kernels) and the pinned staging buffers.
    Exposed as `syncopy_amd.release_device_buffers()`."""
    _handover.clear()
    _pin.clear()
    _prewarm_join()
    if True:
        with _result_lock:
            for k, lst in _result_free.items():
                _result_bytes[0] -= k * len(lst)
            _result_free.clear()
            for ev, blk in _result_zombies:
                ev.synchronize()
                _result_bytes[0] -= blk.numel()
            del _result_zombies[:]
    for ctx in _contexts.values():
        ctx.lib.spyhip_ctx_trim(ctx.handle)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def handover_buffer(shape, device, dtype=torch.complex64):
    key = (tuple(shape), str(device), dtype)
    buf = _handover.get(key)
    if buf is None:
        _handover.clear()                       # keep at most one (they are large)
        buf = torch.empty(shape, dtype=dtype, device=device)
        _handover[key] = buf
    return buf


# FFT -> CSD hand-over layout of the coherence path: the channel-blocked layout makes the FFT's stores coalesced
# (13.1 -> 9.7 us/trial at 256 ch x 4096) but the CSD kernel's fetch scattered (35.4 -> 37.8 us/trial, ~1.7x the
# algorithmic HBM reads by the PMC counters); the standard (rows, F, C) layout stays the default.
USE_BLOCKED_HANDOVER = False


class FFTPlan:
    """spyhip_fft_plan: tapered FFT of segments of a (rows x ld) float32 matrix."""

    def __init__(self, nsig, nfft, nchan, tapers, scale, detrend=None, demean_taper=False, freq_idx=None,
                 output="pow", keeptapers=True, device=None, reference_mean=False):
        """`reference_mean`: constant detrending (detrend=0) subtracts the per-channel mean in the reference's own
        float32 row-order summation (spyhip_fft_plan_set_reference_mean) - for whole trials."""
        self.ctx = context(device)
        tapers = np.ascontiguousarray(np.atleast_2d(tapers), dtype=np.float64)
        assert tapers.shape[1] == nsig, (tapers.shape, nsig)
        self.nsig, self.nfft, self.nchan, self.ntaper = int(nsig), int(nfft), int(nchan), tapers.shape[0]
        self.output = output
        self.kind = OUTPUT_KIND[output]
        self.keeptapers = bool(keeptapers)
        nf = self.nfft // 2 + 1
        if freq_idx is None:
            fi_p, self.nfsel = None, nf
        else:
            fi = np.ascontiguousarray(freq_idx, dtype=np.int32)
            fi_p, self.nfsel = fi.ctypes.data_as(_lib.c_i32p), int(fi.size)
        h = C.c_void_p()
        self.ctx.bind_stream()
        check(self.ctx.lib.spyhip_fft_plan_create(
            self.ctx.handle, self.nsig, self.nfft, self.nchan, self.ntaper,
            tapers.ctypes.data_as(_lib.c_f64p), float(scale), DETREND[detrend], int(bool(demean_taper)),
            fi_p, self.nfsel, self.kind, int(self.keeptapers), C.byref(h)), "spyhip_fft_plan_create")
        self.handle = h
        self.kout = self.ntaper if self.keeptapers else 1
        self.out_dtype = torch.complex64 if self.kind == 2 else torch.float32
        self.blocked = False
        self.reference_mean = int(reference_mean)      # 0 | 1 (True): float32 arrays, reference-order mean | 2: float64 segments
        if self.reference_mean:
            check(self.ctx.lib.spyhip_fft_plan_set_reference_mean(self.handle, self.reference_mean),
                  "spyhip_fft_plan_set_reference_mean")

    def set_precision(self, reference=True):
        """`reference=True`: float64 taper product and FFT, rounded to complex64 where the reference rounds
        (spyhip_fft_plan_set_precision; power-of-two nfft 256 ... 4096).  Returns False if this plan cannot."""
        return self.ctx.lib.spyhip_fft_plan_set_precision(self.handle, int(bool(reference))) == 0

    def set_blocked(self, on=True):
        """Channel-quad-blocked hand-over layout (nseg*ntaper, ceil(nchan/4), nfsel, 4) for csd_accumulate(...,
        blocked=True); returns False (layout unchanged) when the plan cannot use it."""
        rc = self.ctx.lib.spyhip_fft_plan_set_blocked(self.handle, int(bool(on)))
        if rc == 0:
            self.blocked = bool(on)
        return rc == 0

    @property
    def kernel_name(self):
        return self.ctx.lib.spyhip_fft_plan_kernel_name(self.handle).decode()

    def out_shape(self, nseg):
        if self.blocked:
            return (nseg * self.kout, (self.nchan + 3) // 4, self.nfsel, 4)
        return (nseg, self.kout, self.nfsel, self.nchan)

    def execute(self, data, seg_start, seg_lo=None, seg_hi=None, chan_idx=None, out=None, absmax=None):
        """data: (rows, ld) float32 cuda tensor; seg_*: int64 cuda tensors of equal length.
        `absmax`: (nchan,) float32 cuda tensor that every call RAISES to the largest |re|, |im| written per channel - the
        range csd_accumulate(..., absmax=) scales by (spyhip_fft_plan_set_absmax); the returned tensor's
        `spyhip_absmax_tracked` says whether this plan's kernel delivered it (False: `absmax` is untouched, pass
        absmax=None to csd_accumulate).  `self.tracked_absmax` repeats the answer of the LAST call for single-threaded
        callers; the tensor attribute is the one that stays right when a cached plan serves several threads."""
        assert data.is_cuda and data.dtype == torch.float32 and data.dim() == 2 and data.is_contiguous()
        dev = data.device
        nseg = int(seg_start.numel())
        if seg_lo is None:
            seg_lo = seg_start
        if seg_hi is None:
            seg_hi = seg_start + self.nsig
        for t in (seg_start, seg_lo, seg_hi):
            assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() == nseg
        if chan_idx is not None:
            assert chan_idx.is_cuda and chan_idx.dtype == torch.int32 and chan_idx.numel() == self.nchan
        else:
            assert data.shape[1] >= self.nchan
        if out is None:
            out = torch.empty(self.out_shape(nseg), dtype=self.out_dtype, device=dev)
        else:
            assert out.is_cuda and out.is_contiguous() and out.dtype == self.out_dtype
            assert tuple(out.shape) == self.out_shape(nseg), (tuple(out.shape), self.out_shape(nseg))
        self.ctx.bind_stream()
        tracked = False
        if absmax is not None:
            assert absmax.is_cuda and absmax.dtype == torch.float32 and absmax.numel() == self.nchan and absmax.is_contiguous()
            tracked = self.ctx.lib.spyhip_fft_plan_set_absmax(self.handle, _ptr(absmax)) == 0      # (-3: not tracked)
        self.tracked_absmax = tracked
        try:
            check(self.ctx.lib.spyhip_fft_exec(self.handle, _ptr(data), int(data.shape[1]), _ptr(chan_idx),
                                               _ptr(seg_start), _ptr(seg_lo), _ptr(seg_hi), nseg, _ptr(out)),
                  "spyhip_fft_exec")
        finally:
            if self.tracked_absmax:
                self.ctx.lib.spyhip_fft_plan_set_absmax(self.handle, None)
        out.spyhip_absmax_tracked = self.tracked_absmax
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.ctx.lib.spyhip_fft_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class CWTPlan:
    """spyhip_cwt_plan: Morlet CWT (overlap-save FFT convolution) of segments of the trial matrix."""

    FAMILY = {"Morlet": 0, "MorletSL": 1, "Paul": 2, "DOG": 3}

    def __init__(self, nsig, nchan, scales, dt, w0=6.0, detrend=None, output="pow", tpos=None, ntime_out=None,
                 device=None, sl_cycles=None, k_sd=5.0, family=None, order=None):
        """`sl_cycles`: superlet formulation MorletSL with that many cycles (spyhip_cwt_plan_create_sl) instead of
        Morlet(w0).  `family` = "Paul" | "DOG" with `order` m: the other wavelet functions of the reference
        (spyhip_cwt_plan_create_family; Ricker / Marr / Mexican_hat = DOG with m = 2)."""
        self.ctx = context(device)
        scales = np.ascontiguousarray(scales, dtype=np.float64)
        self.nsig, self.nchan, self.nscales = int(nsig), int(nchan), int(scales.size)
        self.kind = OUTPUT_KIND[output]
        if tpos is None:
            tp, self.ntime_out = None, self.nsig
        else:
            tpa = np.ascontiguousarray(tpos, dtype=np.int32)
            assert tpa.size == self.nsig
            tp, self.ntime_out = tpa.ctypes.data_as(_lib.c_i32p), int(ntime_out)
        h = C.c_void_p()
        self.ctx.bind_stream()
        if family in ("Paul", "DOG"):
            check(self.ctx.lib.spyhip_cwt_plan_create_family(
                self.ctx.handle, self.nsig, self.nchan, self.nscales, scales.ctypes.data_as(_lib.c_f64p), float(dt),
                self.FAMILY[family], float(order), 0.0, DETREND[detrend], self.kind, tp, self.ntime_out, C.byref(h)),
                "spyhip_cwt_plan_create_family")
        elif sl_cycles is not None:
            check(self.ctx.lib.spyhip_cwt_plan_create_sl(
                self.ctx.handle, self.nsig, self.nchan, self.nscales, scales.ctypes.data_as(_lib.c_f64p), float(dt),
                float(sl_cycles), float(k_sd), DETREND[detrend], self.kind, tp, self.ntime_out, C.byref(h)),
                "spyhip_cwt_plan_create_sl")
        else:
            check(self.ctx.lib.spyhip_cwt_plan_create(
                self.ctx.handle, self.nsig, self.nchan, self.nscales, scales.ctypes.data_as(_lib.c_f64p), float(dt),
                float(w0), DETREND[detrend], self.kind, tp, self.ntime_out, C.byref(h)), "spyhip_cwt_plan_create")
        self.handle = h
        self.out_dtype = torch.complex64 if self.kind == 2 else torch.float32

    def set_precision(self, reference=True):
        """`reference=True`: float64 FFT convolutions rounded to complex64 where the reference rounds
        (spyhip_cwt_plan_set_precision).  Returns False if this plan cannot."""
        return self.ctx.lib.spyhip_cwt_plan_set_precision(self.handle, int(bool(reference))) == 0

    def set_direct(self, on=True):
        """Scales on short blocks written by their transform kernel in the output layout (default) or, `on=False`, every
        scale through the staging buffer and the transposition pass (spyhip_cwt_plan_set_direct)."""
        check(self.ctx.lib.spyhip_cwt_plan_set_direct(self.handle, int(bool(on))), "spyhip_cwt_plan_set_direct")

    def out_shape(self, nseg):
        return (nseg, self.ntime_out, self.nscales, self.nchan)

    def execute(self, data, seg_start, trial_lo, trial_hi, chan_idx=None, out=None, accumulate=False):
        """accumulate: False/0 store, True/1 out[b] += segment b, 2: out[0] += sum over all segments."""
        assert data.is_cuda and data.dtype == torch.float32 and data.dim() == 2 and data.is_contiguous()
        nseg = int(seg_start.numel())
        for t in (seg_start, trial_lo, trial_hi):
            assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() == nseg
        if chan_idx is not None:
            assert chan_idx.is_cuda and chan_idx.dtype == torch.int32 and chan_idx.numel() == self.nchan
        if out is None:
            assert not accumulate
            out = torch.zeros(self.out_shape(nseg), dtype=self.out_dtype, device=data.device)
        else:
            assert out.is_cuda and out.is_contiguous() and out.dtype == self.out_dtype
            assert tuple(out.shape) == self.out_shape(1 if int(accumulate) == 2 else nseg)
        self.ctx.bind_stream()
        check(self.ctx.lib.spyhip_cwt_exec(self.handle, _ptr(data), int(data.shape[1]), _ptr(chan_idx),
                                           _ptr(seg_start), _ptr(trial_lo), _ptr(trial_hi), nseg, _ptr(out),
                                           int(accumulate)), "spyhip_cwt_exec")
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.ctx.lib.spyhip_cwt_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class csd_phase_exact:
    """`with backend.csd_phase_exact(on):` - K4 arithmetic of the current device's context for the block
    (include/spyhip.h: spyhip_csd_set_phase_exact).  The 3-multiplication kernels keep complex values, moduli and real
    parts within rtol 1e-5 but subtract three independently rounded row sums for the imaginary part; coherence
    outputs that ARE the imaginary part or the phase ("imag", "angle") take the 4-multiplication kernels, whose
    imaginary part is summed directly like the reference's complex64 products (connectivity/csd.py:98-102)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            ctx = context()
            check(ctx.lib.spyhip_csd_set_phase_exact(ctx.handle, 1), "spyhip_csd_set_phase_exact")
        return self

    def __exit__(self, *exc):
        if self.on:
            ctx = context()
            check(ctx.lib.spyhip_csd_set_phase_exact(ctx.handle, 0), "spyhip_csd_set_phase_exact")
        return False


def csd_accumulate(spec, acc, blocked=False, absmax=None, split=True, ranges=None):
    """acc[f,i,j] += sum_r spec[r,f,i] conj(spec[r,f,j]) on the lower triangle (MFMA).
    spec: (..., F, C) complex64 (leading dims flattened to rows), or with blocked=True the hand-over layout
    (rows, ceil(C/4), F, 4) of FFTPlan.set_blocked; acc: (F, C, C) complex64.
    256 channels in the standard layout run on the half-precision matrix cores with split operands
    (spyhip_csd_accumulate_split); `absmax`: (256,) float32 bound of |re|, |im| per channel as FFTPlan.execute(...,
    absmax=) leaves it, or None: the library takes its own pass over the spectra first.  `split=False` keeps 256 channels
    on the float32 matrix instructions (spyhip_csd_accumulate: float32 operands, the reference's operand precision).
    `ranges` = [(f0, f1), ...] (frequency_ranges): the K4h update launched range by range, `acc.spyhip_range_events` =
    [(f0, f1, event)] of THIS call for coh_pipeline (None after a call in one piece)."""
    assert spec.is_cuda and spec.dtype == torch.complex64 and spec.is_contiguous()
    assert acc.is_cuda and acc.dtype == torch.complex64 and acc.is_contiguous()
    ctx = context(spec.device)
    ctx.bind_stream()
    if blocked:
        F, Cn = acc.shape[0], acc.shape[1]
        assert spec.dim() == 4 and tuple(spec.shape[1:]) == ((Cn + 3) // 4, F, 4) and acc.shape[2] == Cn, \
            (tuple(spec.shape), tuple(acc.shape))
        check(ctx.lib.spyhip_csd_accumulate_blocked(ctx.handle, _ptr(spec), int(spec.shape[0]), F, Cn, _ptr(acc)),
              "spyhip_csd_accumulate_blocked")
        return acc
    F, Cn = spec.shape[-2], spec.shape[-1]
    assert tuple(acc.shape) == (F, Cn, Cn), (tuple(acc.shape), (F, Cn, Cn))
    nrows = spec.numel() // (F * Cn)
    if Cn == 256 and nrows > 0 and split:
        if absmax is not None:
            assert absmax.is_cuda and absmax.dtype == torch.float32 and absmax.numel() == 256 and absmax.is_contiguous()
        if ranges and absmax is not None and nrows >= 1024 and not os.environ.get("SPYHIP_CSD_F32"):
            # (small batches - a recording consumed chunk by chunk behind its upload - stay in one piece: eight launches of
            # a few hundred rows each cost more than the pipeline gives a call that waits for the bus anyway)
            # frequency range by frequency range, an event behind each: whoever turns the accumulator into a result can
            # start on range r while range r + 1 is still being accumulated (coh_pipeline)
            stream = torch.cuda.current_stream(spec.device)
            events = []
            for f0, f1 in ranges:
                check(ctx.lib.spyhip_csd_accumulate_split_range(ctx.handle, _ptr(spec), nrows, F, Cn, _ptr(acc), _ptr(absmax),
                                                                int(f0), int(f1 - f0)), "spyhip_csd_accumulate_split_range")
                ev = torch.cuda.Event()
                ev.record(stream)
                events.append((int(f0), int(f1), ev))
            acc.spyhip_range_events = events
            return acc
        acc.spyhip_range_events = None
        check(ctx.lib.spyhip_csd_accumulate_split(ctx.handle, _ptr(spec), nrows, F, Cn, _ptr(acc), _ptr(absmax)),
              "spyhip_csd_accumulate_split")
        return acc
    check(ctx.lib.spyhip_csd_accumulate(ctx.handle, _ptr(spec), nrows, F, Cn, _ptr(acc)), "spyhip_csd_accumulate")
    return acc


class HostLanding:
    """A page-locked block of the result pool that an asynchronous device-to-host copy is filling.  `array()` waits for
    the copy and hands the block out as a NumPy array (it returns to the pool when the array and its views are garbage);
    a landing nobody reads gives its block back when it dies - once the copy has finished (until then the block waits in
    `_result_zombies`), so that the next user of the block is not overwritten by it."""

    def __init__(self, block, nbytes, shape, np_dtype):
        self.block, self.nbytes, self.shape, self.np_dtype, self.done = block, nbytes, tuple(shape), np_dtype, None

    def tensor(self, torch_dtype):
        return self.block[:self.nbytes].view(torch_dtype).reshape(self.shape)

    def array(self):
        import weakref
        if self.block is None:                       # handed out before: the same memory again
            return self._array
        self.done.synchronize()
        block, self.block = self.block, None
        root = block.numpy()
        weakref.finalize(root, _result_release, block)
        self._array = root[:self.nbytes].view(self.np_dtype).reshape(self.shape)
        return self._array

    def __del__(self):
        try:
            if self.block is not None:
                if self.done is not None and not self.done.query():
                    with _result_lock:               # still being written: parked until the copy is through
                        _result_zombies.append((self.done, self.block))
                else:
                    _result_release(self.block)
        except Exception:
            pass


_side_streams = {}


def _side_stream(device):
    key = str(device)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def coh_pipeline(acc, scale, output, events):
    """The coherence result of an accumulator whose last update came range by range (csd_accumulate(ranges=)): on a side
    stream, for every range as soon as its event has passed, the fused normalisation (coh_from_accumulator) and the
    asynchronous copy of that range into a page-locked landing block - under the matrix products of the next range.  The
    0.54 GB of a 256-channel coherence then cost the caller the copy of the LAST range (~2.5 ms) instead of 9.5 ms behind
    the kernels.  Returns (device result (F, C, C), HostLanding or None when the result pool is full).  The caller's stream
    waits for the normalisation, not for the copies."""
    F, Cn, _ = acc.shape
    kind = OUTPUT_KIND[output]
    odt = torch.complex64 if kind == 2 else torch.float32
    res = torch.empty((F, Cn, Cn), dtype=odt, device=acc.device)
    nbytes = res.numel() * res.element_size()
    _prewarm_join()
    block = _result_block(nbytes)
    landing = HostLanding(block, nbytes, (F, Cn, Cn), _NP_DTYPE[odt]) if block is not None else None
    host = landing.tensor(odt) if landing is not None else None
    main, side = torch.cuda.current_stream(acc.device), _side_stream(acc.device)
    acc.record_stream(side)
    res.record_stream(side)
    normalised = torch.cuda.Event()
    with torch.cuda.stream(side):
        last = len(events) - 1
        for k, (f0, f1, ev) in enumerate(events):
            side.wait_event(ev)
            coh_from_accumulator(acc[f0:f1], scale, output, out=res[f0:f1])
            if k == last:
                normalised.record(side)
            if host is not None:
                # in pieces of ~16 MB: the small host-to-device copies of the NEXT analysis (segment tables) queue behind
                # whatever piece is on the bus, not behind a whole range
                step = max(1, (16 << 20) // (Cn * Cn * res.element_size()))
                for g0 in range(f0, f1, step):
                    g1 = min(f1, g0 + step)
                    host[g0:g1].copy_(res[g0:g1], non_blocking=True)
        if landing is not None:
            landing.done = torch.cuda.Event()
            landing.done.record(side)
    main.wait_event(normalised)
    return res, landing


def csd_split_fallbacks(device=None):
    """Frequencies the last 256-channel csd_accumulate of this device's context left to the float32 kernels
    (spyhip_csd_split_fallbacks; synchronises)."""
    ctx = context(device)
    n = C.c_int(0)
    check(ctx.lib.spyhip_csd_split_fallbacks(ctx.handle, C.byref(n)), "spyhip_csd_split_fallbacks")
    return int(n.value)


def csd_kernel_name(nchan, blocked=False):
    """Name of the csd_accum_kernel instance spyhip_csd_accumulate launches for `nchan` channels (launch policy of
    csrc/csd.hip), for matching rocprofv3 rows."""
    nt = (nchan + 31) // 32
    ntiles = nt * (nt + 1) // 2
    import os
    if nchan == 256 and not blocked and not os.environ.get("SPYHIP_CSD_F32"):
        return "spycsd::csdh_kernel"
    if nchan == 256:
        return "spycsd::csd3m_kernel<256, 8, true, false, false>"
    if nchan <= 512 and not blocked:
        return "spycsd::csd3m_kernel<%d, 8, false>" % ((nchan + 15) // 16 * 16)
    if nchan > 512 and not blocked:
        return "spycsd::csd3m_kernel<512, 8, false, true> (+ csd3m_kernel<256, 8, false> per 256-channel block)"
    if not blocked and nchan <= 256:
        return "spycsd::csd_accum_kernel<5, 4, %d>" % (1 if nchan == 256 else 2)
    if not blocked and nchan <= 512:
        return "spycsd::csd_accum_kernel<5, 4, 3>"
    if ntiles >= 21:
        return "spycsd::csd_accum_kernel<5, 4, 0>"
    return "spycsd::csd_accum_kernel<3, 2, 0>" if ntiles >= 6 else "spycsd::csd_accum_kernel<1, 1, 0>"


_lib_comm = {}          # device index -> identity of the process group the library's RCCL communicator was built under
_lib_comm_failed = {}   # device index -> True once the communicator could not be created on every rank (torch route from then on)


def _group_identity():
    """What makes the communicator stale: another WORLD group object, world size or rank (destroy_process_group +
    init_process_group between two analyses)."""
    import torch.distributed as dist
    return (id(dist.group.WORLD), dist.get_world_size(), dist.get_rank())


def _library_comm(ctx):
    """The library's RCCL communicator on this context (include/spyhip.h: spyhip_comm_*), created once per process
    group: rank 0 draws the unique id, the existing process group carries its 128 bytes to the others (the only use
    of torch.distributed on this path), every rank joins.  Collective: all ranks arrive here together - at the first
    sum over ranks of an analysis.  A communicator built under an earlier process group (other size / rank / group
    object) is destroyed and rebuilt - reusing it would hang or sum over the wrong ranks.  The communicator spans the
    WORLD group."""
    import torch.distributed as dist
    ident = _group_identity()
    have = _lib_comm.get(ctx.device)
    if have == ident:
        return True
    if have is not None:
        check(ctx.lib.spyhip_comm_destroy(ctx.handle), "spyhip_comm_destroy")
        _lib_comm.pop(ctx.device, None)
    rank, size = dist.get_rank(), dist.get_world_size()
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_ubyte * 128)()
        check(ctx.lib.spyhip_comm_unique_id(buf), "spyhip_comm_unique_id")
        uid = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
    uid = uid.cuda(ctx.device)
    dist.broadcast(uid, src=0)
    raw = (C.c_ubyte * 128).from_buffer_copy(uid.cpu().numpy().tobytes())
    ok, err = 1, None
    try:
        check(ctx.lib.spyhip_comm_init(ctx.handle, raw, rank, size), "spyhip_comm_init")
    except SpyHipError as exc:                   # (librccl not found, an initialisation error on this rank ...)
        ok, err = 0, exc
    # every rank must take the same route from here on: one that failed while the others joined would leave them waiting
    # in the first collective - the (working) process group carries the verdict
    flag = torch.tensor([ok], dtype=torch.int32, device=uid.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if ok:
            ctx.lib.spyhip_comm_destroy(ctx.handle)
        _lib_comm_failed[ctx.device] = True
        if rank == 0:
            import sys
            sys.stderr.write("syncopy_amd: the library's RCCL communicator could not be created on every rank (%s); sums over "
                             "ranks go through torch.distributed instead\n" % (err if err is not None else "another rank failed"))
        return False
    _lib_comm[ctx.device] = ident
    return True


def shutdown_library_comm():
    """Destroy the library's RCCL communicators (before the process group that bootstrapped them goes away).  Also
    registered with atexit (below)."""
    for dev in list(_lib_comm):
        ctx = _contexts.get(dev)
        if ctx is not None:
            try:
                check(ctx.lib.spyhip_comm_destroy(ctx.handle), "spyhip_comm_destroy")
            except Exception:                       # noqa: BLE001 - teardown: the device may be gone already
                pass
        _lib_comm.pop(dev, None)


import atexit as _atexit
_atexit.register(shutdown_library_comm)


def csd_allreduce_(acc):
    """Sum the (lower-triangle) CSD accumulator over all ranks in place: the ONE collective of the coherence path
    (the reference's mutex-guarded `+=`, kwarg_decorators.py:723-735).  Only the lower triangle carries data before
    csd_finalize: `spyhip_allreduce_csd` packs it to (F, C(C+1)/2), sums it with the library's own RCCL communicator
    over xGMI in a fixed rank order and unpacks it (half the bytes).  No-op for a single process; under a gloo group
    (the CPU tests never get here) the packed triangle goes through torch.distributed instead."""
    from . import parallel
    if not parallel.collective_active():
        return acc
    assert acc.is_cuda and acc.dtype == torch.complex64 and acc.is_contiguous() and acc.dim() == 3
    F, Cn, _ = acc.shape
    ctx = context(acc.device)
    ctx.bind_stream()
    import torch.distributed as dist
    if (dist.get_backend() == "nccl" and not os.environ.get("SPY_TORCH_COLLECTIVE") and not _lib_comm_failed.get(ctx.device)
            and _library_comm(ctx)):
        ctx.bind_stream()
        check(ctx.lib.spyhip_allreduce_csd(ctx.handle, _ptr(acc), F, Cn), "spyhip_allreduce_csd")
        return acc
    packed = torch.empty((F, Cn * (Cn + 1) // 2), dtype=torch.complex64, device=acc.device)
    check(ctx.lib.spyhip_csd_tril_pack(ctx.handle, _ptr(acc), F, Cn, _ptr(packed)), "spyhip_csd_tril_pack")
    parallel.allreduce_sum_(packed)
    check(ctx.lib.spyhip_csd_tril_unpack(ctx.handle, _ptr(packed), F, Cn, _ptr(acc)), "spyhip_csd_tril_unpack")
    return acc


def csd_tril_pack(acc):
    """(F, C, C) complex64 -> packed lower triangle (F, C(C+1)/2)."""
    F, Cn, _ = acc.shape
    ctx = context(acc.device)
    ctx.bind_stream()
    packed = torch.empty((F, Cn * (Cn + 1) // 2), dtype=torch.complex64, device=acc.device)
    check(ctx.lib.spyhip_csd_tril_pack(ctx.handle, _ptr(acc), F, Cn, _ptr(packed)), "spyhip_csd_tril_pack")
    return packed


def csd_tril_unpack(packed, acc):
    """Write the packed lower triangle back into acc (F, C, C); the upper triangle is left alone."""
    F, Cn, _ = acc.shape
    ctx = context(acc.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_csd_tril_unpack(ctx.handle, _ptr(packed), F, Cn, _ptr(acc)), "spyhip_csd_tril_unpack")
    return acc


def csd_finalize(acc, scale):
    """Scale the accumulated lower triangle and mirror it: acc becomes the full Hermitian CSD."""
    assert acc.is_cuda and acc.dtype == torch.complex64 and acc.is_contiguous() and acc.dim() == 3
    F, Cn, _ = acc.shape
    ctx = context(acc.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_csd_finalize(ctx.handle, _ptr(acc), F, Cn, float(scale)), "spyhip_csd_finalize")
    return acc


def coh_normalize(csd, output="abs"):
    """Coherency csd_ij / sqrt(csd_ii csd_jj) + output conversion; csd: (F, C, C) complex64, full Hermitian."""
    assert csd.is_cuda and csd.dtype == torch.complex64 and csd.is_contiguous() and csd.dim() == 3
    F, Cn, _ = csd.shape
    kind = OUTPUT_KIND[output]
    out = torch.empty((F, Cn, Cn), dtype=torch.complex64 if kind == 2 else torch.float32, device=csd.device)
    ctx = context(csd.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_coh_normalize(ctx.handle, _ptr(csd), F, Cn, kind, _ptr(out)), "spyhip_coh_normalize")
    return out


def frequency_ranges(nfreq, device=None, parts=8):
    """[(f0, f1)] cutting `nfreq` frequencies into (at most) `parts` runs of whole rounds of workgroups (one K4h workgroup
    per frequency and CU), the remainder with the last; None below two rounds.  256 channels x 2049 frequencies, result
    read on the host: 27.8 / 25.3 / 24.0 ms per call with 2 / 4 / 8 ranges (32.0 in one piece)."""
    ncu = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
    rounds = nfreq // ncu
    parts = min(parts, rounds)
    if parts < 2:
        return None
    per = rounds // parts
    edges = [k * per * ncu for k in range(parts)] + [nfreq]
    return [(edges[k], edges[k + 1]) for k in range(parts)]


def coh_from_accumulator(acc, scale, output="abs", out=None):
    """Coherence straight from the RAW lower-triangle accumulator of csd_accumulate (scale = 1/(tapers*trials)):
    csd_finalize + coh_normalize fused, bit-identical, a third of the traffic.  acc is left untouched.  Every frequency
    is on its own: `acc[f0:f1]` with `out=res[f0:f1]` does a range."""
    assert acc.is_cuda and acc.dtype == torch.complex64 and acc.is_contiguous() and acc.dim() == 3
    F, Cn, _ = acc.shape
    kind = OUTPUT_KIND[output]
    odt = torch.complex64 if kind == 2 else torch.float32
    if out is None:
        out = torch.empty((F, Cn, Cn), dtype=odt, device=acc.device)
    else:
        assert out.is_cuda and out.is_contiguous() and out.dtype == odt and tuple(out.shape) == (F, Cn, Cn)
    ctx = context(acc.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_coh_from_accumulator(ctx.handle, _ptr(acc), F, Cn, float(scale), kind, _ptr(out)),
          "spyhip_coh_from_accumulator")
    return out


def ppc_accumulate(spec, ntaper, acc):
    """acc (F, C, C) complex64 += sum over trials of the unit phasors of the single-trial cross spectra, straight
    from tapered spectra spec (ntrials * ntaper, F, C) complex64 (K7; lower triangle maintained)."""
    assert spec.is_cuda and spec.dtype == torch.complex64 and spec.is_contiguous() and spec.dim() == 3
    assert acc.is_cuda and acc.dtype == torch.complex64 and acc.is_contiguous()
    R, F, Cn = spec.shape
    assert R % ntaper == 0 and tuple(acc.shape) == (F, Cn, Cn)
    ctx = context(spec.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_ppc_accumulate(ctx.handle, _ptr(spec), R // ntaper, int(ntaper), F, Cn, _ptr(acc)),
          "spyhip_ppc_accumulate")


def ppc_accumulate_csd(csd, acc):
    """acc (...) complex64 += unit phasors of the single-trial cross spectra csd (ntrials, ...) complex64."""
    assert csd.is_cuda and csd.dtype == torch.complex64 and csd.is_contiguous()
    assert acc.is_cuda and acc.dtype == torch.complex64 and acc.is_contiguous() and tuple(csd.shape[1:]) == tuple(acc.shape)
    ctx = context(csd.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_ppc_accumulate_csd(ctx.handle, _ptr(csd), csd.shape[0], acc.numel(), _ptr(acc)),
          "spyhip_ppc_accumulate_csd")


def ppc_finalize(acc, ntrials, lower_only):
    """Pairwise phase consistency (F, ni, nj) float32 from the accumulated phasor sums of `ntrials` trials."""
    assert acc.is_cuda and acc.dtype == torch.complex64 and acc.is_contiguous() and acc.dim() == 3
    F, ni, nj = acc.shape
    out = torch.empty((F, ni, nj), dtype=torch.float32, device=acc.device)
    ctx = context(acc.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_ppc_finalize(ctx.handle, _ptr(acc), F, ni, nj, int(bool(lower_only)), int(ntrials),
                                      _ptr(out)), "spyhip_ppc_finalize")
    return out


def jack_coh_accumulate(spec, ntaper, csd, direct, output, ntrials_total, sum_d, sum_d2):
    """Streaming jackknife of the coherence (K9): for every trial of spec (ntrials * ntaper, F, C) the leave-one-out
    coherence replicate minus `direct` is summed into sum_d (float64 / complex128) and its squared modulus into
    sum_d2 (float64); csd = finalised trial average (F, C, C) complex64."""
    assert spec.is_cuda and spec.dtype == torch.complex64 and spec.is_contiguous() and spec.dim() == 3
    R, F, Cn = spec.shape
    kind = OUTPUT_KIND[output]
    assert R % ntaper == 0 and tuple(csd.shape) == (F, Cn, Cn) and csd.dtype == torch.complex64 and csd.is_contiguous()
    assert tuple(direct.shape) == (F, Cn, Cn) and direct.is_contiguous()
    assert direct.dtype == (torch.complex64 if kind == 2 else torch.float32)
    assert sum_d.dtype == (torch.complex128 if kind == 2 else torch.float64) and sum_d.is_contiguous()
    assert sum_d2.dtype == torch.float64 and sum_d2.is_contiguous() and tuple(sum_d2.shape) == (F, Cn, Cn)
    ctx = context(spec.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_jack_coh_accumulate(ctx.handle, _ptr(spec), R // ntaper, int(ntaper), F, Cn, _ptr(csd),
                                             _ptr(direct), kind, int(ntrials_total), _ptr(sum_d), _ptr(sum_d2)),
          "spyhip_jack_coh_accumulate")


def ccov_nfft(nsamples):
    """Transform length K8 needs for trials of `nsamples` samples (no GPU involved)."""
    from ._lib import load
    n = load().spyhip_ccov_nfft(int(nsamples))
    if n < 0:
        raise ValueError(f"cross-covariance: trials of {nsamples} samples exceed the supported 349525")
    return n


def ccov_from_accumulator(acc, nsamples, scale, norm=0):
    """Cross-covariance lags (nlag, C, C) float32 from the raw accumulator (nfft/2+1, C, C) of csd_accumulate over
    the spectra of zero-padded trials (K8).  norm: 0 none, 1 zero-lag auto-covariances, 2 np.std products."""
    assert acc.is_cuda and acc.dtype == torch.complex64 and acc.is_contiguous() and acc.dim() == 3
    F, Cn, _ = acc.shape
    nlag = nsamples // 2 + (nsamples & 1)
    out = torch.empty((nlag, Cn, Cn), dtype=torch.float32, device=acc.device)
    ctx = context(acc.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_ccov_from_accumulator(ctx.handle, _ptr(acc), 2 * (F - 1), Cn, int(nsamples), float(scale),
                                               int(norm), _ptr(out)), "spyhip_ccov_from_accumulator")
    return out


def ccov_normalize_(cc):
    """In place: cc (nlag, C, C) float32 /= sqrt(cc[0,a,a] cc[0,b,b]) (normalize_ccov_cF)."""
    assert cc.is_cuda and cc.dtype == torch.float32 and cc.is_contiguous() and cc.dim() == 3
    ctx = context(cc.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_ccov_normalize(ctx.handle, _ptr(cc), cc.shape[0], cc.shape[1]), "spyhip_ccov_normalize")
    return cc


def slt_combine(acc, spec, s0, expo, init, modulus_only=False, square=False):
    """One factor of the superlet geometric mean: acc[..., s0+q, :] = (init ? 1 : acc) * spec[..., q, :] ** expo[q].
    acc (nseg, ntime, nscales, C), spec (nseg, ntime, nsub, C) on the device, complex64 - or float32 moduli (ABS output
    of the plan; `square` stores the squared product: last factor of a POW output); expo: nsub host floats;
    modulus_only (complex arrays): fold |spec| ** expo."""
    real = acc.dtype == torch.float32
    assert acc.is_cuda and acc.is_contiguous() and acc.dim() == 4 and acc.dtype in (torch.complex64, torch.float32)
    assert spec.is_cuda and spec.dtype == acc.dtype and spec.is_contiguous() and spec.dim() == 4
    assert acc.shape[:2] == spec.shape[:2] and acc.shape[3] == spec.shape[3]
    assert real or not square
    expo = np.ascontiguousarray(expo, dtype=np.float64)
    assert expo.size == spec.shape[2] and s0 + expo.size <= acc.shape[2]
    mode = (2 | (4 if square else 0)) if real else int(bool(modulus_only))
    ctx = context(acc.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_slt_combine(ctx.handle, _ptr(acc), _ptr(spec), acc.shape[0] * acc.shape[1], acc.shape[2],
                                     spec.shape[2], int(s0), acc.shape[3], expo.ctypes.data_as(_lib.c_f64p),
                                     int(bool(init)), mode), "spyhip_slt_combine")
    return acc


def spec_convert(spec, output):
    """spectralConversions of a complex64 device tensor: real kinds -> float32 tensor, 'fourier'/'complex' -> itself."""
    kind = OUTPUT_KIND[output]
    if kind == 2:
        return spec
    assert spec.is_cuda and spec.dtype == torch.complex64 and spec.is_contiguous()
    out = torch.empty(spec.shape, dtype=torch.float32, device=spec.device)
    ctx = context(spec.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_spec_convert(ctx.handle, _ptr(spec), spec.numel(), kind, _ptr(out)), "spyhip_spec_convert")
    return out


def granger(csd, rtol=5e-6, niter=100, cond_max=1e4, eps_max=1e-1, want_factors=False):
    """Wilson spectral factorisation + Granger causality of a trial-averaged CSD (F, C, C) complex64.
    Returns (granger float32 (F,C,C), info dict[, H complex128 (F,C,C), Sigma complex128 (C,C)])."""
    assert csd.is_cuda and csd.dtype == torch.complex64 and csd.is_contiguous() and csd.dim() == 3
    F, Cn, _ = csd.shape
    out = torch.empty((F, Cn, Cn), dtype=torch.float32, device=csd.device)
    H = torch.empty((F, Cn, Cn), dtype=torch.complex128, device=csd.device) if want_factors else None
    Sig = torch.empty((Cn, Cn), dtype=torch.complex128, device=csd.device) if want_factors else None
    info = (C.c_double * 4)()
    ctx = context(csd.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_granger(ctx.handle, _ptr(csd), F, Cn, float(rtol), int(niter), float(cond_max),
                                 float(eps_max), _ptr(out), _ptr(H), _ptr(Sig), info), "spyhip_granger")
    meta = {"converged": bool(info[0]), "max rel. err": float(info[1]), "reg. factor": float(info[2]),
            "initial cond. num": float(info[3])}
    if meta["reg. factor"] == 0.0:
        meta["reg. factor"] = 0
    elif meta["reg. factor"] == -1.0:
        meta["reg. factor"] = -1
    return (out, meta, H, Sig) if want_factors else (out, meta)


def granger_stats(device=None):
    """Diagnostics of the last granger() call on the device: {'iterations': Wilson iterations run}."""
    ctx = context(device)
    return {"iterations": int(ctx.lib.spyhip_granger_last_iterations(ctx.handle))}


def axis_nanmean(x, axis):
    """np.nanmean(x, axis, keepdims=True) of one trial array (float32 / complex64) on the device, in NumPy's summation
    order (statistics/compRoutines.py:22-57)."""
    assert x.is_cuda and x.dtype in (torch.float32, torch.complex64) and x.is_contiguous()
    axis = axis % x.dim()
    outer = int(np.prod(x.shape[:axis], dtype=np.int64))
    n = int(x.shape[axis])
    inner = int(np.prod(x.shape[axis + 1:], dtype=np.int64))
    shape = list(x.shape)
    shape[axis] = 1
    out = torch.empty(shape, dtype=x.dtype, device=x.device)
    ctx = context(x.device)
    ctx.bind_stream()
    check(ctx.lib.spyhip_axis_nanmean(ctx.handle, _ptr(x), outer, n, inner, int(x.is_complex()), _ptr(out)),
          "spyhip_axis_nanmean")
    return out


def trial_mean(x):
    """Sequential float32 sum over axis 0 followed by one division (trial averaging order of the reference)."""
    assert x.is_cuda and x.dtype in (torch.float32, torch.complex64) and x.is_contiguous()
    T = x.shape[0]
    n = x.numel() // T
    out = torch.empty(x.shape[1:], dtype=x.dtype, device=x.device)
    ctx = context(x.device)
    ctx.bind_stream()
    if x.is_complex():          # complex64: the reference's complex division by the count (1/T as a float32 factor)
        check(ctx.lib.spyhip_trial_mean_c64(ctx.handle, _ptr(x), _ptr(out), T, n), "spyhip_trial_mean_c64")
    else:
        check(ctx.lib.spyhip_trial_mean_f32(ctx.handle, _ptr(x), _ptr(out), T, n), "spyhip_trial_mean_f32")
    return out
