"""Device-side execution of the (multi-)tapered FFT family: host bookkeeping only
(taper tables, plan cache, segment lists); every number is produced by libspyhip.

Used by the compute functions in compRoutines.py both for one trial at a time
(the reference's cF contract) and for all trials at once (in-HBM trial queue).
"""
import numpy as np
import torch
from .. import backend
from ..shared.errors import SPYValueError
from .tapers import spec_scale, taper_table  # noqa: F401

_plan_cache = {}
MAX_CACHED_PLANS = 64


def _bounded_put(cache, key, value):
    """Insert into a plan cache that keeps the MAX_CACHED_PLANS most recently USED entries (plans own device tables:
    a long session over many shapes must not accumulate them); `_cache_hit` keeps the order up to date."""
    while len(cache) >= MAX_CACHED_PLANS:
        cache.pop(next(iter(cache)))            # dicts iterate in insertion order: the first key is the least recent
    cache[key] = value


def _cache_hit(cache, key):
    """Look up `key`; a hit moves the entry to the most-recent end."""
    value = cache.get(key)
    if value is not None:
        cache[key] = cache.pop(key)
    return value


_precision = ["float32"]        # arithmetic of the tapered FFT plans created inside `with precision(...)`
_advice = None                  # a list while connectivityanalysis(precision="auto") runs its float32 attempt: the
                                # coherence stage records here that the data's dynamic range asks for float64 transforms


class precision:
    """`with hs.precision("reference"):` - tapered-FFT plans requested inside the block transform in float64 and round
    to complex64 where the reference does (mtmfft.py:96-127; spyhip_fft_plan_set_precision).  Default "float32"."""

    def __init__(self, kind):
        if kind not in ("float32", "reference"):
            raise SPYValueError("'float32' or 'reference'", varname="precision", actual=str(kind))
        self.kind = kind

    def __enter__(self):
        _precision.append(self.kind)
        return self

    def __exit__(self, *exc):
        _precision.pop()
        return False


def get_plan(nsig, nfft, nchan, taper, taper_opt, nnorm, scale, detrend, demean_taper, freq_idx, output, keeptapers,
             device, blocked=False, whole_trials=True, float32_frames=False):
    """Cached FFTPlan; `blocked` asks for the channel-blocked hand-over layout of the CSD path (the plan's
    `.blocked` tells whether the kernel serving this length supports it).  `whole_trials`: the segments are trials
    that the reference detrends as float32 arrays (compRoutines.py:169-170, ST_compRoutines.py:405-409), so the
    mean is taken in its float32 row order; sliding-window frames are detrended in float64 there (stft.py:112-132)
    and keep the kernels' float64 block sums - unless the frames stay float32 (`float32_frames`: stft.py:101-132 with
    boundary=None, padded=False, the `toi` array case): they are strided views of the (time x channel) trial, and
    np.mean over their last axis walks the samples of a frame in order with one float32 accumulator per channel
    (pairwise for a single channel) - the same rounding sequence as the mean of a whole trial."""
    fkey = None if freq_idx is None else np.asarray(freq_idx, dtype=np.int32).tobytes()
    key = (int(nsig), int(nfft), int(nchan), taper, tuple(sorted((taper_opt or {}).items())), int(nnorm),
           float(scale), detrend, bool(demean_taper), fkey, output, bool(keeptapers), str(device), bool(blocked),
           bool(whole_trials), bool(float32_frames), _precision[-1])
    plan = _cache_hit(_plan_cache, key)
    if plan is None:
        tp = taper_table(taper, nsig, nnorm, taper_opt)
        plan = backend.FFTPlan(nsig, nfft, nchan, tp, scale, detrend, demean_taper, freq_idx, output,
                               keeptapers, device=device,
                               reference_mean=(1 if detrend == 0 else 0) if (whole_trials or float32_frames) else 2)
        if _precision[-1] == "reference":
            if not plan.set_precision(True):
                raise SPYValueError("a transform length up to 2^20 without a prime factor above 61 (e.g. "
                                    "pad='nextpow2') for precision='reference'", varname="precision",
                                    actual=f"nfft = {int(nfft)}")
        elif blocked:
            plan.set_blocked(True)
        _bounded_put(_plan_cache, key, plan)
    return plan


def full_freq_idx(freq_idx, nfft):
    """None if the selection is the identity (saves the index lookup in the kernel)."""
    if freq_idx is None:
        return None
    freq_idx = np.asarray(freq_idx)
    nf = nfft // 2 + 1
    if freq_idx.size == nf and np.array_equal(freq_idx, np.arange(nf)):
        return None
    return freq_idx.astype(np.int32)


def run_mtmfft(dev_data, rows, chan_idx, nfft, taper, taper_opt, demean_taper, ft_compat, polyremoval, freq_idx,
               output, keeptapers):
    """Tapered FFT of the trials `rows` = [(start, stop)] of the device matrix.
    Returns a list of (Kout, F, C) device tensors views into per-length batches, one per trial."""
    device = dev_data.device
    nchan = dev_data.shape[1] if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else torch.tensor(np.asarray(chan_idx), dtype=torch.int32, device=device)
    lengths = np.array([b - a for a, b in rows])
    results = [None] * len(rows)
    for n in np.unique(lengths):
        which = np.nonzero(lengths == n)[0]
        n = int(n)
        N = n if nfft is None else int(nfft)
        plan = get_plan(n, N, nchan, taper, taper_opt, N, spec_scale(n, N, ft_compat), polyremoval, demean_taper,
                        full_freq_idx(freq_idx, N), output, keeptapers, device)
        starts = torch.tensor([rows[i][0] for i in which], dtype=torch.int64, device=device)
        out = plan.execute(dev_data, starts, chan_idx=ci)
        for k, i in enumerate(which):
            results[i] = out[k]
    return results


def run_stft(dev_data, row0, soi_start, soi_stop, frames, nperseg, step, boundary, chan_idx, taper, taper_opt,
             polyremoval, freq_idx, output, keeptapers):
    """Sliding-window tapered FFT (stft.py:16-159 / mtmconvol.py:136-150) of one trial.
    frames: indices of the STFT frames to evaluate; frame s starts at
    soi_start + s*step - (nperseg//2 if boundary), samples outside [soi_start, soi_stop) are zero."""
    device = dev_data.device
    nchan = dev_data.shape[1] if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else torch.tensor(np.asarray(chan_idx), dtype=torch.int32, device=device)
    opt = dict(taper_opt or {})
    if taper == "dpss":
        opt["sym"] = False          # mtmconvol.py:110-111
    plan = get_plan(nperseg, nperseg, nchan, taper, opt, nperseg, np.sqrt(2) / nperseg, polyremoval, False,
                    full_freq_idx(freq_idx, nperseg), output, keeptapers, device, whole_trials=False,
                    float32_frames=not boundary)
    frames = np.asarray(frames, dtype=np.int64)
    lead = nperseg // 2 if boundary else 0
    starts = torch.from_numpy(row0 + soi_start + frames * step - lead).to(device)
    lo = torch.full_like(starts, row0 + soi_start)
    hi = torch.full_like(starts, row0 + soi_stop)
    return plan.execute(dev_data, starts, lo, hi, chan_idx=ci)


def run_mtmfft_batches(dev_data, rows, chan_idx, nfft, taper, taper_opt, demean_taper, ft_compat, polyremoval,
                       freq_idx, output, keeptapers, max_bytes=32 << 30, blocked=False, reuse=False):
    """Generator over (trial indices, (B, Kout, F, C) device tensor) batches: trials of equal length share a
    plan; a batch is bounded by `max_bytes` of spectra so the intermediate stays a small part of HBM.
    With `blocked` the tensor is in the plan's hand-over layout whenever `tensor.dim() == 4 and
    tensor.shape[-1] == 4 and plan supports it` - callers check `spyhip_blocked` on the yielded tensor."""
    device = dev_data.device
    nchan = dev_data.shape[1] if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else torch.tensor(np.asarray(chan_idx), dtype=torch.int32, device=device)
    lengths = np.array([b - a for a, b in rows])
    for n in np.unique(lengths):
        which = np.nonzero(lengths == n)[0]
        n = int(n)
        N = n if nfft is None else int(nfft)
        plan = get_plan(n, N, nchan, taper, taper_opt, N, spec_scale(n, N, ft_compat), polyremoval, demean_taper,
                        full_freq_idx(freq_idx, N), output, keeptapers, device, blocked=blocked)
        per_trial = int(np.prod(plan.out_shape(1))) * (8 if plan.kind == 2 else 4)
        bmax = max(1, int(max_bytes // per_trial))
        for i in range(0, which.size, bmax):
            sel = which[i:i + bmax]
            starts = torch.tensor([rows[j][0] for j in sel], dtype=torch.int64, device=device)
            # reuse=True: the consumer is done with a batch before it asks for the next one (stream order), so all
            # batches - and all later calls of the same shape - share one device buffer
            buf = backend.handover_buffer(plan.out_shape(len(sel)), device) if reuse and plan.kind == 2 else None
            spec = plan.execute(dev_data, starts, chan_idx=ci, out=buf)
            spec.spyhip_blocked = plan.blocked
            spec.spyhip_ntaper = plan.kout
            yield sel, spec
