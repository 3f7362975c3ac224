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


_rows_cache = {}


def _device_rows(host_rows, device):
    """Segment start rows (int64) on the device.  Repeated analyses of one recording ask for the same tables: a small
    cache saves the host-to-device copy - which otherwise queues behind whatever the copy engine is busy with (the result
    of the previous analysis on its way to the host)."""
    if host_rows.nbytes > (1 << 20):
        return torch.from_numpy(host_rows).to(device)
    key = (str(device), host_rows.tobytes())
    t = _cache_hit(_rows_cache, key)
    if t is None:
        t = torch.from_numpy(host_rows).to(device)
        _bounded_put(_rows_cache, key, t)
    return t


def _cache_hit(cache, key):
    """Look up `key`; a hit moves the entry to the most-recent end."""
    value = cache.get(key)
    if value is not None:
        cache[key] = cache.pop(key)
    return value


# Arithmetic of the tapered-FFT plans requested from here on - a stack of "float32" | "reference" | None.
#   None (the default, precision="auto" of the front ends): the ROUTE decides -
#     * compute functions called one trial at a time (the reference's own trial loop, INTEGRATION.md route B) transform in
#       float64: every call moves a trial over PCIe both ways, the transform's cost is invisible next to that, and the
#       result then carries the reference's own rounding (mtmfft.py:96-127) whatever the data;
#     * the batched routes transform in float32 unless the front end decided otherwise for this call (`freqanalysis`:
#       outputs that isolate a part of the complex spectrum; `connectivityanalysis`: `needs_float64` below).
_precision = [None]


class PrecisionUnavailable(SPYValueError):
    """No float64 kernel serves this transform length (beyond 2^20 points)."""


class precision:
    """`with hs.precision("reference"):` - tapered-FFT plans requested inside the block transform in float64 and round
    to complex64 where the reference does (mtmfft.py:96-127; spyhip_fft_plan_set_precision); "float32": the fast
    kernels; "auto" / None: the route's default (above)."""

    def __init__(self, kind):
        if kind not in ("float32", "reference", "auto", None):
            raise SPYValueError("'float32', 'reference' or 'auto'", varname="precision", actual=str(kind))
        self.kind = None if kind == "auto" else kind

    def __enter__(self):
        _precision.append(self.kind)
        return self

    def __exit__(self, *exc):
        _precision.pop()
        return False


def requested_precision():
    """What the innermost `with precision(...)` asked for: "float32", "reference" or None (the route decides)."""
    return _precision[-1]


def per_trial_route():
    """Context of a compute function that serves ONE trial per call: float64 transforms unless the caller fixed the
    precision."""
    return _Pushed("reference?" if _precision[-1] is None else _precision[-1])


def soft_reference():
    """Context of a batched call whose front end chose float64 transforms on its own (precision="auto"): as
    "reference", but a length no float64 kernel serves falls back to float32 instead of raising."""
    return _Pushed("reference?")


class _Pushed:
    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        _precision.append(self.kind)
        return self

    def __exit__(self, *exc):
        _precision.pop()
        return False


def plan_precision():
    return "reference" if _precision[-1] in ("reference", "reference?") else "float32"


def dynamic_range(spec, kept=None, few_products=False):
    """How far the weakest part of a channel's KEPT spectrum sits below its mean power, judged on tapered spectra
    (B, K, F, C) complex64 of a few trials over the WHOLE frequency axis: max over channels of mean_f P / q_2%(P[kept])
    with P = the trial- and taper-averaged power (`spec`: the power spectra of every taper, or complex spectra).  The mean runs over every bin - an offset nobody removed or a line
    outside the kept band raises the float32 transform's error in the kept bins just the same; the two kept bins next
    to DC are left out of the quantile (they belong to the detrending).  A low percentile, not the minimum: one empty
    bin (a notch, the Nyquist bin of an even filter) is not what the spectrum is like.
    `few_products`: the caller averages so few (trial, taper) products (<= 16) that nothing averages the float32 error of ONE
    weak bin away - then the weakest kept bin counts, the bins next to DC included (a coherence of 0.003 at the DC bin of
    four demeaned Hann-tapered trials sat at 1.13 x the criterion, tests/test_gpu_fuzz.py family 500000 seed 284; float64
    transforms of a handful of trials cost nothing)."""
    from .. import backend
    if spec.is_complex():
        spec = spec.abs().square()
    # power spectra (B, K, F, C) float32: the mean over trials and tapers with the library's own reduction, the rest on
    # the host on 2 MB (no torch kernels: the first use of each costs 0.1-0.4 s of code-object loading in a fresh process)
    F, C = spec.shape[-2], spec.shape[-1]
    p = backend.trial_mean(spec.reshape(-1, F, C).contiguous()).cpu().numpy().astype(np.float64)
    num = p.mean(axis=0)
    if kept is not None:
        kept = np.asarray(kept)
        if not few_products:
            kept = kept[kept >= 2] if (kept >= 2).sum() >= 4 else kept
        p = p[kept]
    elif p.shape[0] >= 6 and not few_products:
        p = p[2:]
    q = np.maximum(p.min(axis=0) if few_products else np.quantile(p, 0.02, axis=0), 1e-38)
    return float((num / q).max())


def selection_hides_peak(dev_data, rows, chan_idx, nfft, taper, taper_opt, polyremoval, freq_idx, nfull):
    """precision="auto" for 'pow' / 'abs' spectra on the batched route.  The float32 transform's error is ~1.5e-7 of the
    LARGEST bin of a spectrum; the parity criterion's floor is 1e-6 of the largest bin that is KEPT.  With the whole
    frequency axis kept the first is always inside the second.  A `foi` / `foilim` selection that leaves the dominant
    bins out - an offset nobody removed (polyremoval=None; the un-detrended windows of a `toi` array), line noise below
    a high-pass band - turns that error into 1.5e-7 x (largest bin / largest kept bin) of the output's scale: float64
    transforms from a factor 3 on.  Judged on the float32 power spectra of the segments `rows` (equal length, a
    handful), the largest ratio over the ranks decides."""
    from .. import parallel
    # only rank-INDEPENDENT early-outs ahead of the collective: a rank whose shard is empty (more ranks than trials, a
    # first trial shorter than the window) contributes the ratio 0 and still takes part, so that every rank enters the
    # same all-reduce and picks the same precision
    if freq_idx is None or len(freq_idx) >= nfull:
        return False
    ratio = 0.0
    if len(rows):
        n = rows[0][1] - rows[0][0]
        rows = [r for r in rows if r[1] - r[0] == n][:16]
        N = n if nfft is None else int(nfft)
        with precision("float32"):
            spec = run_mtmfft(dev_data, rows, chan_idx, N, taper, taper_opt, False, False, polyremoval, None, "pow", False)
        p = torch.stack(spec, dim=0)[:, 0]                                    # (B, F, C)
        kept = p.index_select(1, torch.as_tensor(np.asarray(freq_idx), device=p.device))
        ratio = float((p.amax(dim=(0, 1)) / kept.amax(dim=(0, 1)).clamp_min(1e-38)).max()) if kept.numel() else 0.0
    ratio = parallel.allreduce_max(ratio)
    return bool(np.sqrt(ratio) > 3.0)


_SOFT_FALLBACK = "float32 (no float64 kernel for this length)"


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
           bool(whole_trials), bool(float32_frames), plan_precision())
    plan = _cache_hit(_plan_cache, key)
    if plan is None and plan_precision() == "reference" and _precision[-1] != "reference":
        # a soft request that fell back to float32 earlier left its plan under the fall-back key (below)
        plan = _cache_hit(_plan_cache, key[:-1] + (_SOFT_FALLBACK,))
        if plan is not None:
            return plan
    if plan is None:
        tp = taper_table(taper, nsig, nnorm, taper_opt)
        plan = backend.FFTPlan(nsig, nfft, nchan, tp, scale, detrend, demean_taper, freq_idx, output,
                               keeptapers, device=device,
                               reference_mean=(1 if detrend == 0 else 0) if (whole_trials or float32_frames) else 2)
        if plan_precision() == "reference":
            if not plan.set_precision(True):
                if _precision[-1] == "reference":
                    raise PrecisionUnavailable("a transform length up to 2^20 for float64 transforms", varname="precision",
                                               actual=f"nfft = {int(nfft)}")
                # the soft request ("reference?") falls back to float32: that plan must not sit under the "reference" key,
                # where a later explicit precision="reference" would take it for a float64 plan - nor under "float32",
                # where it would be found but never looked for by the next soft request (a plan rebuilt per call)
                key = key[:-1] + (_SOFT_FALLBACK,)
                if blocked:
                    plan.set_blocked(True)
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


def run_stft_trials(dev_data, trials, nperseg, step, boundary, chan_idx, taper, taper_opt, polyremoval, freq_idx, output,
                    keeptapers, max_bytes=8 << 30):
    """run_stft for MANY trials in as few launches as `max_bytes` of spectra allow: trials = [(row0, soi_start, soi_stop,
    frames)]; returns one device tensor (n_frames, Kout, F, C) per trial (views of the batches)."""
    device = dev_data.device
    nchan = dev_data.shape[1] if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else torch.tensor(np.asarray(chan_idx), dtype=torch.int32, device=device)
    opt = dict(taper_opt or {})
    if taper == "dpss":
        opt["sym"] = False          # mtmconvol.py:110-111
    plan = get_plan(nperseg, nperseg, nchan, taper, opt, nperseg, np.sqrt(2) / nperseg, polyremoval, False,
                    full_freq_idx(freq_idx, nperseg), output, keeptapers, device, whole_trials=False,
                    float32_frames=not boundary)
    lead = nperseg // 2 if boundary else 0
    per_frame = int(np.prod(plan.out_shape(1))) * (8 if plan.kind == 2 else 4)
    fmax = max(1, int(max_bytes // per_frame))
    out, k = [None] * len(trials), 0
    while k < len(trials):
        k1, nfr = k, 0
        while k1 < len(trials) and (k1 == k or nfr + len(trials[k1][3]) <= fmax):
            nfr += len(trials[k1][3])
            k1 += 1
        if nfr == 0:                  # (no window fits these trials: empty results - the caller's shape check speaks, as before)
            for i in range(k, k1):
                out[i] = torch.empty((0,) + tuple(plan.out_shape(1)[1:]), device=device,
                                     dtype=torch.complex64 if plan.kind == 2 else torch.float32)
            k = k1
            continue
        st = np.concatenate([r0 + a + np.asarray(fr, dtype=np.int64) * step - lead for r0, a, _, fr in trials[k:k1]])
        lo = np.concatenate([np.full(len(fr), r0 + a, dtype=np.int64) for r0, a, _, fr in trials[k:k1]])
        hi = np.concatenate([np.full(len(fr), r0 + b, dtype=np.int64) for r0, _, b, fr in trials[k:k1]])
        res = plan.execute(dev_data, torch.from_numpy(st).to(device), torch.from_numpy(lo).to(device),
                           torch.from_numpy(hi).to(device), chan_idx=ci)
        off = 0
        for i in range(k, k1):
            n = len(trials[i][3])
            out[i] = res[off:off + n]
            off += n
        k = k1
    return out


def run_mtmfft_batches(dev_data, rows, chan_idx, nfft, taper, taper_opt, demean_taper, ft_compat, polyremoval,
                       freq_idx, output, keeptapers, max_bytes=32 << 30, blocked=False, reuse=False, upload=None):
    """Generator over (trial indices, (B, Kout, F, C) device tensor) batches: trials of equal length share a
    plan; a batch is bounded by `max_bytes` of spectra so the intermediate stays a small part of HBM.
    With `blocked` the tensor is in the plan's hand-over layout whenever `tensor.dim() == 4 and
    tensor.shape[-1] == 4 and plan supports it` - callers check `spyhip_blocked` on the yielded tensor."""
    device = dev_data.device
    nchan = dev_data.shape[1] if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else torch.tensor(np.asarray(chan_idx), dtype=torch.int32, device=device)
    rows_arr = np.asarray(rows, dtype=np.int64).reshape(-1, 2)        # (trials, 2): no per-trial Python work below
    lengths = rows_arr[:, 1] - rows_arr[:, 0]
    for n in np.unique(lengths):
        which = np.nonzero(lengths == n)[0]
        n = int(n)
        N = n if nfft is None else int(nfft)
        plan = get_plan(n, N, nchan, taper, taper_opt, N, spec_scale(n, N, ft_compat), polyremoval, demean_taper,
                        full_freq_idx(freq_idx, N), output, keeptapers, device, blocked=blocked)
        per_trial = int(np.prod(plan.out_shape(1))) * (8 if plan.kind == 2 else 4)
        bmax = max(1, int(max_bytes // per_trial))
        if upload is not None and not upload.complete:
            # the recording is still on its way into HBM (backend.Upload): batches of about one upload chunk, each
            # launched as soon as its rows have arrived - the transforms of chunk k run under the copy of chunk k + 1
            bmax = max(1, min(bmax, 2 * upload.chunk_rows() // max(n, 1)))
        for i in range(0, which.size, bmax):
            sel = which[i:i + bmax]
            if upload is not None:
                upload.wait_rows(int(rows_arr[sel, 1].max()))
            starts = _device_rows(np.ascontiguousarray(rows_arr[sel, 0]), device)
            # reuse=True: the consumer is done with a batch before it asks for the next one (stream order), so all
            # batches - and all later calls of the same shape - share one device buffer
            buf = backend.handover_buffer(plan.out_shape(len(sel)), device) if reuse and plan.kind == 2 else None
            # complex all-taper spectra of 256 channels are what K4h consumes (backend.csd_accumulate): let the transform
            # kernel leave the per-channel range it scales by (fresh per batch; None where the kernel family cannot)
            am = None
            if nchan == 256 and plan.kind == 2 and keeptapers and not plan.blocked:
                am = torch.zeros(256, dtype=torch.float32, device=device)
            spec = plan.execute(dev_data, starts, chan_idx=ci, out=buf, absmax=am)
            spec.spyhip_blocked = plan.blocked
            spec.spyhip_ntaper = plan.kout
            spec.spyhip_absmax = am if getattr(spec, "spyhip_absmax_tracked", plan.tracked_absmax) else None
            yield sel, spec
