"""K4h - the 256-channel cross-spectral update on the half-precision matrix cores with split float32 operands
(csrc/csdh_kernel.h, spyhip_csd_accumulate_split) - against complex128 products of the same spectra
(connectivity/csd.py:94-102 + the trial sum of shared/computational_routine.py:1022-1032): ragged row counts, frequency
counts around a round of workgroups, channels 18 decades apart, spectral dynamic range up to and beyond what an fp16 pair
holds (the kernel then hands the frequency to the float32 kernels by itself), Inf / NaN, and the range the transform
kernel delivers for it (spyhip_fft_plan_set_absmax)."""
import numpy as np
import pytest

from oracle import spy_oracle as O
from parity import assert_parity

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

C = 256


@pytest.fixture(scope="module")
def be():
    from syncopy_amd import backend
    backend.require_gpu()
    return backend


def _spectra(R, F, seed, gains=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.view_as_complex(torch.randn((R, F, C, 2), generator=g, device="cuda", dtype=torch.float32))
    if gains:       # fT-scale MEG next to volts: the per-channel power-of-two scale must take all of it out
        x = x * torch.logspace(-14, 4, C, device="cuda", dtype=torch.float32)[torch.randperm(C, generator=g, device="cuda")]
    return x.contiguous()


def _check(spec, acc, freqs, what):
    il = np.tril_indices(C)
    for f in freqs:
        x = spec[:, f, :].to(torch.complex128)
        ref = (x.T @ x.conj()).cpu().numpy()
        got = acc[f].cpu().numpy()
        # channel by channel the scales differ by 18 decades: compare the coherency-normalised matrices
        d = np.sqrt(np.real(np.diag(ref)))
        d[d == 0] = 1.0
        n = np.outer(d, d)
        assert_parity((got / n)[il].astype(np.complex64), (ref / n)[il].astype(np.complex64), what=f"{what} f={f}")


@pytest.mark.parametrize("R,F", [(1, 5), (7, 40), (31, 7), (32, 3), (33, 300), (64, 256), (95, 259), (700, 513)])
def test_rows_and_frequencies(be, R, F):
    """Ragged last chunk (rows not a multiple of 32, a single row), fewer / more frequencies than compute units (259 and
    513: the re-cut float32 tail takes the frequencies beyond the last full round), two calls adding up."""
    spec = _spectra(R, F, seed=R * 1000 + F)
    acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(spec, acc)
    assert be.csd_split_fallbacks() == 0
    _check(spec, acc, sorted({0, 1, F // 2, F - 2, F - 1} & set(range(F))), f"R={R} F={F}")
    if R >= 33:
        acc2 = torch.zeros_like(acc)
        be.csd_accumulate(spec[:20].contiguous(), acc2)
        be.csd_accumulate(spec[20:].contiguous(), acc2)
        _check(spec, acc2, [0, F - 1], f"two calls R={R} F={F}")
    again = torch.zeros_like(acc)
    be.csd_accumulate(spec, again)
    assert torch.equal(torch.view_as_real(acc), torch.view_as_real(again)), "deterministic"


def test_spectral_dynamic_range_and_fallback(be):
    """An 80 dB tilt across frequency stays on the half-precision path (the scale is per channel, the fp16 pair spans
    ~2^18 between a channel's peak and its rms at a frequency); a 140 dB line on one channel pushes every other
    frequency of that channel beyond it: the kernel flags those frequencies and the float32 kernels add them - same
    criterion either way."""
    R, F = 200, 300
    spec = _spectra(R, F, seed=4)
    tilt = torch.logspace(0, -4, F, device="cuda", dtype=torch.float32)[None, :, None]
    a = (spec * tilt).contiguous()
    acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(a, acc)
    assert be.csd_split_fallbacks() == 0
    _check(a, acc, [0, 1, 150, 298, 299], "80 dB tilt")
    b = a.clone()
    b[:, 17, 5] *= 1e7
    acc = torch.zeros_like(acc)
    be.csd_accumulate(b, acc)
    nfb = be.csd_split_fallbacks()
    assert 200 <= nfb <= 255, nfb          # of the 256 frequencies of the half-precision launch; 17 and the loud ones stay
    _check(b, acc, [0, 16, 17, 18, 150, 255, 256, 299], "140 dB line")
    # phase-exact contexts: the stand-in for flagged frequencies is the 4-multiplication float32 kernel
    with be.csd_phase_exact(True):
        acc4 = torch.zeros_like(acc)
        be.csd_accumulate(b, acc4)
    _check(b, acc4, [0, 17, 150, 255], "140 dB line, phase-exact")


def test_dead_tiny_and_subnormal_channels(be):
    """A channel of zeros (no error to make), one at 1e-18 (scaled up by 2^73; its auto-spectrum 1e-36 R is about the
    smallest a complex64 accumulator - the reference's own - still holds) and one of subnormal float32 values (cannot be
    scaled into the fp16 pair's range: flagged, float32 kernels)."""
    R, F = 64, 9
    spec = _spectra(R, F, seed=9, gains=False)
    spec[:, :, 3] = 0
    spec[:, :, 8] *= 1e-18
    acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(spec, acc)
    assert be.csd_split_fallbacks() == 0
    _check(spec, acc, range(F), "zero + tiny channels")
    assert float(acc[:, 3, :].abs().max()) == 0.0 and float(acc[:, :, 3].abs().max()) == 0.0
    spec[:, :, 200] *= 1e-41
    acc.zero_()
    be.csd_accumulate(spec, acc)
    assert be.csd_split_fallbacks() == F
    x = spec[:, 4, :].to(torch.complex128)
    ref = (x.T @ x.conj()).cpu().numpy()
    keep = [i for i in range(C) if i not in (3, 8, 200)]
    assert_parity(acc[4].cpu().numpy()[np.ix_(keep, keep)][np.tril_indices(len(keep))],
                  ref[np.ix_(keep, keep)][np.tril_indices(len(keep))].astype(np.complex64), what="subnormal channel present")


def test_nonfinite_values_propagate(be):
    """Inf / NaN in the spectra end up in the accumulator rows / columns of their channel at their frequency, as in the
    reference's own products - through the float32 stand-in; every other frequency is untouched by them."""
    R, F = 70, 12
    spec = _spectra(R, F, seed=2, gains=False)
    spec[5, 7, 9] = float("nan")
    acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(spec, acc)
    assert be.csd_split_fallbacks() == 1           # the range pass skips a NaN: only its own frequency is handed over
    assert bool(torch.isnan(acc[7, 9, :10].real).all()) and bool(torch.isnan(acc[7, 200, 9].real))
    spec[69, 3, 11] = complex(float("inf"), 0.0)   # in the ragged last chunk; the range of channel 11 is now Inf:
    acc.zero_()                                    # no scale exists for it, every frequency goes to the float32 kernels
    be.csd_accumulate(spec, acc)
    assert be.csd_split_fallbacks() == F
    assert bool(torch.isnan(acc[7, 9, :10].real).all()) and bool(torch.isnan(acc[7, 200, 9].real))
    assert not bool(torch.isfinite(acc[3, 11, 11].real))
    for f in (0, 2, 4, 6, 8, 11):
        assert bool(torch.isfinite(torch.view_as_real(acc[f])).all())
    clean = spec.clone()
    clean[5, 7, 9] = 0
    clean[69, 3, 11] = 0
    _check(clean, acc, [0, 6, 8, 11], "frequencies without Inf / NaN")


@pytest.mark.parametrize("N,nchan,K", [(256, 256, 2), (1024, 37, 3), (4096, 256, 7), (8192, 8, 1)])
def test_fft_plan_delivers_the_range(be, N, nchan, K):
    """spyhip_fft_plan_set_absmax: every exec raises absmax[c] to a bound of the |re|, |im| it wrote for channel c
    (taper norm x 2-norm of the detrended segment): never below the true maximum, within sqrt(N) of it on noise, for
    every power-of-two length of the packed kernel."""
    B = 5
    rng = np.random.default_rng(N + nchan)
    data = rng.normal(size=(B * N, nchan)).astype(np.float32) * np.logspace(-3, 3, nchan).astype(np.float32)
    d = torch.from_numpy(data).cuda()
    starts = torch.arange(B, device="cuda", dtype=torch.int64) * N
    tapers = O.taper_table("dpss", N, N, {"NW": 4.0, "Kmax": K}) if K > 1 else O.taper_table("hann", N, N)
    plan = be.FFTPlan(N, N, nchan, tapers, np.sqrt(2) / N, 0, False, None, "fourier", True)
    am = torch.zeros(nchan, dtype=torch.float32, device="cuda")
    spec = plan.execute(d, starts[:3], absmax=am)
    assert plan.tracked_absmax
    ref = torch.view_as_real(spec).abs().amax(dim=(0, 1, 2, 4))
    assert bool((am >= ref).all()) and bool((am <= ref * np.sqrt(N)).all()), (am / ref).cpu().numpy()
    first = am.clone()
    spec2 = plan.execute(d, starts[3:], absmax=am)          # a second call only raises it
    ref2 = torch.maximum(ref, torch.view_as_real(spec2).abs().amax(dim=(0, 1, 2, 4)))
    assert bool((am >= first).all()) and bool((am >= ref2).all()) and bool((am <= ref2 * np.sqrt(N)).all())
    # a channel riding on an offset nobody removes: the DC bin IS the bound (tight where the bits are needed)
    if K == 1:
        off = be.FFTPlan(N, N, nchan, tapers, np.sqrt(2) / N, None, False, None, "fourier", True)
        am0 = torch.zeros(nchan, dtype=torch.float32, device="cuda")
        s0 = off.execute(d + 1000.0 * float(np.abs(data).max()), starts[:2], absmax=am0)
        r0 = torch.view_as_real(s0).abs().amax(dim=(0, 1, 2, 4))
        assert bool((am0 >= r0).all()) and bool((am0 <= 1.5 * r0).all()), (am0 / r0).cpu().numpy()
    plain = plan.execute(d, starts[:3])                     # and without it the spectra are the same bits
    assert torch.equal(torch.view_as_real(plain), torch.view_as_real(spec))
    if nchan == 256:
        F = N // 2 + 1
        acc, own = (torch.zeros((F, C, C), dtype=torch.complex64, device="cuda") for _ in range(2))
        be.csd_accumulate(spec, acc, absmax=am)
        be.csd_accumulate(spec, own)
        # (the library's own pass finds the exact maximum of these three trials, `am` covers five: scales may differ by a
        # power of two, results only in the last bits)
        _check(spec.reshape(-1, F, C), acc, [0, 1, F // 2, F - 1], "range from the transform kernel")
        _check(spec.reshape(-1, F, C), own, [0, 1, F // 2, F - 1], "range from the library's own pass")


@pytest.mark.parametrize("N,reference", [(2000, False), (3000, False), (12000, False), (4096, True), (2000, True)])
def test_range_for_every_kernel_family(be, N, reference):
    """Lengths without an in-kernel bound (decimal schedules, N = P M through HBM) and float64 transforms take a pass over
    the segments ahead of the transform: still a bound, still within sqrt(N) of the truth, with a mean removed or not."""
    nchan, B = 12, 3
    rng = np.random.default_rng(N)
    data = (rng.normal(size=(B * N, nchan)) * np.logspace(-2, 2, nchan) + 50.0).astype(np.float32)
    d = torch.from_numpy(data).cuda()
    starts = torch.arange(B, device="cuda", dtype=torch.int64) * N
    tapers = O.taper_table("hann", N, N)
    for detrend in (0, None):
        plan = be.FFTPlan(N, N, nchan, tapers, np.sqrt(2) / N, detrend, False, None, "fourier", True)
        if reference:
            assert plan.set_precision(True)
        am = torch.zeros(nchan, dtype=torch.float32, device="cuda")
        spec = plan.execute(d, starts, absmax=am)
        assert plan.tracked_absmax
        ref = torch.view_as_real(spec).abs().amax(dim=(0, 1, 2, 4))
        assert bool((am >= ref).all()) and bool((am <= ref * np.sqrt(N)).all()), (detrend, (am / ref).cpu().numpy())


def test_plans_without_complex_spectra_say_so(be):
    """Plans that do not write complex all-taper spectra leave the tensor alone and report it."""
    d = torch.randn((1024, 8), device="cuda")
    am = torch.zeros(8, dtype=torch.float32, device="cuda")
    pow_plan = be.FFTPlan(1024, 1024, 8, O.taper_table("hann", 1024, 1024), 1.0, 0, False, None, "pow", True)
    pow_plan.execute(d, torch.zeros(1, device="cuda", dtype=torch.int64), absmax=am)
    assert not pow_plan.tracked_absmax and float(am.max()) == 0.0


def test_ranges_equal_one_piece_and_pipeline_result():
    """The cross-spectral update launched frequency range by frequency range (spyhip_csd_accumulate_split_range) is the same
    update - bit for bit, flagged frequencies and the re-cut float32 tail included - and the pipelined coherence
    (normalisation and host copy of range r under the products of range r + 1, backend.coh_pipeline) lands the same array
    on the host as the fused kernel followed by a plain copy."""
    import torch
    from syncopy_amd import backend as be
    be.require_gpu()
    g = torch.Generator(device="cuda").manual_seed(21)
    R, F, C = 1024, 2049, 256                 # (batches below 1024 rows stay in one piece: backend.csd_accumulate)
    x = torch.randn((R, F, C, 2), generator=g, device="cuda", dtype=torch.float32)
    x[:, 700, 5] *= 1e7                       # a line 140 dB above one channel's floor in range 2: left to the float32 kernel
    x[:, 1800, 9] *= 1e7                      # ... and one in the last range
    spec = torch.view_as_complex(x)
    am = torch.view_as_real(spec).abs().amax(dim=(0, 1, 3)).contiguous()
    ranges = be.frequency_ranges(F)
    assert ranges == [(256 * k, 256 * (k + 1) if k < 7 else 2049) for k in range(8)]
    one = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(spec, one, absmax=am)
    flagged_one = be.csd_split_fallbacks()
    assert one.spyhip_range_events is None
    parts = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(spec, parts, absmax=am, ranges=ranges)
    assert be.csd_split_fallbacks() == flagged_one >= 2
    assert torch.equal(torch.view_as_real(one), torch.view_as_real(parts))
    # in any order
    rev = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(spec, rev, absmax=am, ranges=ranges[::-1])
    assert torch.equal(torch.view_as_real(one), torch.view_as_real(rev))
    for output in ("abs", "complex"):
        ref = be.to_host(be.coh_from_accumulator(one, 1.0 / R, output))
        res, landing = be.coh_pipeline(parts, 1.0 / R, output, parts.spyhip_range_events)
        assert landing is not None
        got = landing.array()
        assert got.shape == ref.shape and got.dtype == ref.dtype and np.array_equal(got, ref, equal_nan=True)
        assert np.array_equal(res.cpu().numpy(), ref, equal_nan=True)
        del got, landing                      # the block goes back to the pool ...
    res, landing = be.coh_pipeline(parts, 1.0 / R, "abs", parts.spyhip_range_events)
    del landing                               # ... also when nobody reads it
    torch.cuda.synchronize()


def test_front_end_coherence_through_the_pipeline():
    """spy.connectivityanalysis(method='coh') on 256 channels x 4096 samples (2049 frequencies: eight ranges): the pipelined
    result against the same analysis with the pipeline switched off (a process group of one rank keeps the update in one
    piece) - the same kernels on the same data, so equal bit for bit - and `.data` readable twice."""
    import torch
    import syncopy_amd as spy
    from syncopy_amd import backend as be
    T, N, C = 160, 4096, 256                  # 1120 rows of spectra: above the 1024 from which the update goes by ranges
    x = spy.synthdata.ar2_uncoupled_fast(C, N, T, seed=3).cpu().numpy()
    trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
    data = spy.AnalogData(x, samplerate=1000.0, trialdefinition=trl)
    seen = []
    keep_pipe = be.coh_pipeline
    be.coh_pipeline = lambda *a, **k: (seen.append(1), keep_pipe(*a, **k))[1]
    try:
        got = spy.connectivityanalysis(data, method="coh", tapsmofrq=1)
    finally:
        be.coh_pipeline = keep_pipe
    assert seen, "the pipelined path did not run"
    a = np.array(got.data)
    assert a.shape == (1, 2049, 256, 256) and np.isfinite(a).all() and np.allclose(a[0, :, np.arange(256), np.arange(256)], 1, atol=1e-5)
    assert np.array_equal(a, np.array(got.data))
    keep = be.frequency_ranges
    be.frequency_ranges = lambda *args, **kw: None
    try:
        plain = spy.connectivityanalysis(data, method="coh", tapsmofrq=1)
    finally:
        be.frequency_ranges = keep
    assert np.array_equal(a, np.array(plain.data))
