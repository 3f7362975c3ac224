"""Seeded random sweep over the front-end options (GPU kernels vs the oracle bound to the same front ends): trial
lengths 3 ... 6000 (every transform-length class: powers of two, decimal, 5-smooth, Bluestein, primes, long), channel
counts 1 ... 40, ragged trials, every padding / taper / output / detrending combination the path supports.  A case
that the front end itself rejects must be rejected identically by both sides."""
import os

import numpy as np
import pytest
import scipy.signal as sps

import syncopy_amd as spy
from oracle import spy_oracle as O
from oracle_routines import ORACLE_CONN, ORACLE_FREQ
from parity import ATOL_REL, RTOL

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from syncopy_amd import backend
    backend.require_gpu()


SCALE = int(os.environ.get("SPY_FUZZ_SCALE", "1"))      # SPY_FUZZ_SCALE=10: ten times the seeds (exploration runs)
OFFSET = int(os.environ.get("SPY_FUZZ_OFFSET", "0"))     # SPY_FUZZ_OFFSET=100000: a different family of cases
LENGTHS = [3, 7, 16, 30, 64, 100, 127, 128, 200, 250, 256, 257, 360, 500, 512, 729, 1000, 1009, 1024, 1500, 2000, 2048,
           2500, 3000, 3001, 4096, 4100, 5000, 6000]


def _make(rng, ragged, offsets=True, min_trials=1):
    nchan = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 16, 33, 40]))
    ntr = max(int(rng.integers(1, 6)), min_trials)
    n0 = int(rng.choice(LENGTHS))
    lengths = [n0] * ntr
    if ragged:
        lengths = [max(3, int(n0 * f)) for f in rng.uniform(0.5, 1.0, size=ntr)]
    x = rng.normal(size=(sum(lengths), nchan)).astype(np.float32)
    off = rng.normal(size=nchan).astype(np.float32)[None, :] * 3       # (drawn in any case: same data stream per seed)
    if offsets:
        x += off
    edges = np.concatenate([[0], np.cumsum(lengths)])
    trl = np.stack([edges[:-1], edges[1:], np.zeros(ntr)], axis=1)
    return spy.AnalogData(x, samplerate=1000.0, trialdefinition=trl), lengths


def _detrend_exact(x, polyremoval):
    """O.detrend with the FIT in float64 and everything else as scipy.signal.detrend does it on a float32 trial: the
    trend `A @ coef` is rounded to float32 and subtracted in float32 (signaltools.py: `newdata - A @ coef` in the data's
    dtype).  What is left between this and the reference is the float32 least-squares solve alone (LAPACK sgelsd on
    OpenBLAS kernels: implementation-defined rounding, ~1e-7 of a channel's offset in the two coefficients)."""
    if polyremoval in (0, 1) and np.asarray(x).dtype == np.float32:
        x64 = np.asarray(x, dtype=np.float64)
        trend = x64 - sps.detrend(x64, type="constant" if polyremoval == 0 else "linear", axis=0)
        return np.asarray(x) - trend.astype(np.float32)
    if polyremoval in (0, 1):
        return sps.detrend(np.asarray(x, dtype=np.float64), type="constant" if polyremoval == 0 else "linear",
                           axis=0).astype(np.float32)
    return x


def _detrend_frames_exact(frames, kind):
    """O.detrend_frames with the FIT in float64 and everything else as scipy.signal.detrend does it on the float32 frames
    of a `toi`-array mtmconvol (stft.py:101-132): the trend is rounded to the frames' dtype and subtracted in it
    (signaltools.py: `newdata - A @ coef`) - with channels riding on offsets of 20 standard deviations that rounding is
    1.9e-6 per sample, the same on both sides, and not part of what the float64 fit changes (family 1700000, seed 187:
    the Nyquist bin of 32-sample windows sat at 1.11 x while the float64-rounded variant of this helper was 1.8 x away
    from BOTH sides)."""
    f64 = np.asarray(frames, dtype=np.float64)
    trend = f64 - sps.detrend(f64, type=kind)
    if frames.dtype == np.float32:
        return frames - trend.astype(np.float32)
    return (f64 - trend).astype(frames.dtype)


def _modulus_scale(fn, data, classes, kw):
    """Largest MODULUS behind an output that is a projection of a complex quantity (real / imag): the absolute floor of the
    criterion refers to that scale, not to the largest projection (tests/test_gpu_depth.py: IMAG_ATOL).  Three samples minus
    their regression line under a Hann window leave ONE non-zero sample: every cross spectrum is real, the reference's
    'imag' output is rounding residue of 0 (1e-8) of coherencies of modulus 1, and the real part of the one kept bin of
    such a trial's 4-point transform is residue of a spectrum of modulus 0.4 (families 500000 / 900000: seeds 23, 668,
    1624).  The oracle's own complex result gives the scale."""
    if kw.get("output") not in ("real", "imag"):
        return None
    full = dict(kw, output="complex" if kw.get("method") == "coh" else "fourier")
    if full["output"] == "fourier" and kw.get("method") == "mtmfft":
        full.update(keeptapers=True, keeptrials=True)
    try:
        return float(np.abs(np.asarray(fn(data, **full, compute_method="sequential", routine_classes=classes).data)).max())
    except Exception:                                 # noqa: BLE001 - no scale, the plain criterion
        return None


def _check(got, ref, exact, what, atol_rel=ATOL_REL, scale=None):
    """The shared criterion, widened element by element by twice the reference's OWN detrending noise where there is
    any: scipy.signal.detrend(type="linear") fits a float32 trial with a float32 least-squares solve, which leaves a
    coherent ramp of ~1e-7 of a channel's offset in the data - the bins next to DC of such a channel (and every
    normalised quantity formed there) are made of it.  The kernels fit in float64; `exact` is the oracle with a float64
    fit, |ref - exact| is therefore the reference's rounding, not ours."""
    if getattr(got, "per_trial_route", None) is not None:
        _check(got.per_trial_route, ref, exact, what + " [per-trial route]", atol_rel, scale)
    a, b = np.asarray(got.data), np.asarray(ref.data)
    if np.abs(b).max() < 1e-12:          # (three samples minus their own regression line: the reference result IS rounding
        assert np.abs(a).max() < 1e-6, what     # residue of zero - nothing to compare but the magnitude)
        return
    # `scale`: the largest modulus of the complex quantity a real / imag output is a projection of (_modulus_scale)
    tol = RTOL * np.abs(b) + atol_rel * max(float(np.abs(b).max()), scale or 0.0)
    err = np.abs(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64) - b)
    if exact is not None:
        if "polyremoval': 1" in what or "polyremoval=1" in what:
            _log_detrend_gap(what, float((err / np.where(tol == 0, np.finfo(np.float32).tiny, tol)).max()),
                             float((np.abs(b - np.asarray(exact.data)) / np.where(tol == 0, np.finfo(np.float32).tiny, tol)).max()))
        tol = tol + 2.0 * np.abs(b - np.asarray(exact.data))
    assert np.isfinite(err).all(), what
    r = err / np.where(tol == 0, np.finfo(np.float32).tiny, tol)
    assert r.max() <= 1.0, f"{what}: max err/tol = {r.max():.3g} at {np.unravel_index(r.argmax(), r.shape)}"


def _log_detrend_gap(what, unwidened, ref_vs_exact):
    """the linear-detrending cases WITHOUT the widening - max err/tol of (kernels - reference) under the
    plain criterion and, beside it, the reference's own distance from the float64-fit oracle.  One line per case into
    $SPY_FUZZ_GAP_LOG (tools/final_bench.sh summarises it into profiles/)."""
    path = os.environ.get("SPY_FUZZ_GAP_LOG")
    if path:
        with open(path, "a") as fh:
            fh.write(f"{unwidened:.4g}\t{ref_vs_exact:.4g}\t{what[:160]}\n")


def _run_both(fn, data, classes, kw):
    def call(**extra):
        try:
            return fn(data, **kw, **extra), None
        except Exception as exc:                      # noqa: BLE001 - compared below
            return None, exc
    got, e1 = call()
    ref, e2 = call(compute_method="sequential", routine_classes=classes)
    if e1 is not None or e2 is not None:
        assert e1 is not None and e2 is not None and type(e1) is type(e2), (kw, repr(e1), repr(e2))
        return None, None, None
    assert got.data.shape == ref.data.shape and got.data.dtype == ref.data.dtype, (kw, got.data.shape, ref.data.shape)
    # the per-trial route of INTEGRATION.md (B): the reference's own trial loop around the GPU compute functions
    seq, e3 = call(compute_method="sequential")
    assert e3 is None, (kw, repr(e3))
    assert seq.data.shape == ref.data.shape and seq.data.dtype == ref.data.dtype, (kw, seq.data.shape, ref.data.shape)
    got.per_trial_route = seq
    exact = None
    wav = kw.get("method") in ("wavelet", "superlet")
    if kw.get("polyremoval") == 1 or wav:
        # what the reference's own float32 arithmetic deviates from: the least-squares fit of a linear trend in float64,
        # and - wavelet methods - scipy.signal.fftconvolve fed a float64 trial (it transforms float32 input in SINGLE
        # precision: 1.5e-7 of the largest coefficient, which the roots of superlet products amplify)
        keep, keepf, keepc = O.detrend, O.detrend_frames, sps.fftconvolve
        if kw.get("polyremoval") == 1:
            O.detrend, O.detrend_frames = _detrend_exact, _detrend_frames_exact
        if wav:
            sps.fftconvolve = lambda in1, in2, mode="full", axes=None: keepc(np.asarray(in1, dtype=np.float64), in2, mode=mode, axes=axes)
        try:
            exact, _ = call(compute_method="sequential", routine_classes=classes)
        finally:
            O.detrend, O.detrend_frames, sps.fftconvolve = keep, keepf, keepc
    return got, ref, exact


@pytest.mark.parametrize("seed", range(48 * SCALE))
def test_mtmfft_random_options(seed):
    rng = np.random.default_rng(1000 + OFFSET + seed)
    ragged, polyremoval = bool(rng.integers(0, 2)), [None, 0, 1][int(rng.integers(0, 3))]
    # channels ride on offsets of ~3 standard deviations whenever the call removes them; without detrending an offset
    # is a DC line 60 dB above the spectrum, i.e. the float32 / precision="reference" question of
    # test_gpu_kernels.py::test_reference_precision_resolves_60dB, not this sweep's
    data, lengths = _make(rng, ragged, offsets=polyremoval is not None)
    kw = dict(method="mtmfft")
    kw["output"] = str(rng.choice(["pow", "abs", "fourier", "real", "imag"]))
    kw["polyremoval"] = polyremoval
    kw["pad"] = ["maxperlen", "nextpow2"][int(rng.integers(0, 2))]
    kw["keeptrials"] = bool(rng.integers(0, 2)) or kw["output"] == "fourier"
    tap = int(rng.integers(0, 4))
    if tap == 0:
        kw["taper"] = "hann"
    elif tap == 1:
        kw["taper"] = None
    elif tap == 2:
        kw["taper"] = "hamming"
    else:
        kw["tapsmofrq"] = float(rng.choice([2.0, 5.0, 10.0]))
        kw["keeptapers"] = bool(rng.integers(0, 2)) and kw["keeptrials"]
    if kw["output"] == "fourier":
        kw["keeptapers"] = True
    if rng.integers(0, 3) == 0:
        kw["foilim"] = [float(rng.uniform(0, 100)), float(rng.uniform(150, 500))]
    got, ref, exact = _run_both(spy.freqanalysis, data, ORACLE_FREQ, kw)
    if got is not None:
        _check(got, ref, exact, f"seed {seed}: {kw} lengths {lengths} ch {data.data.shape[1]}",
               scale=_modulus_scale(spy.freqanalysis, data, ORACLE_FREQ, kw))


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_connectivity_random_options(seed):
    rng = np.random.default_rng(2000 + OFFSET + seed)
    polyremoval = [None, 0, 1][int(rng.integers(0, 3))]
    method = str(rng.choice(["coh", "csd", "corr", "ppc"]))
    # ppc of T trials averages cos(phase difference) over T(T-1)/2 trial pairs; where a single-trial cross spectrum
    # passes near zero its phase - in the reference's complex64 arithmetic as much as here - is rounding noise, and with
    # two or three trials nothing averages that out: ppc cases draw at least five trials
    data, lengths = _make(rng, ragged=False, offsets=polyremoval is not None, min_trials=5 if method == "ppc" else 1)
    kw = dict(method=method)
    if method in ("coh", "csd", "ppc"):
        if rng.integers(0, 2):
            kw["tapsmofrq"] = float(rng.choice([3.0, 8.0]))
        else:
            kw["taper"] = "hann"
        kw["pad"] = ["maxperlen", "nextpow2"][int(rng.integers(0, 2))]
    if method == "coh":
        kw["output"] = str(rng.choice(["abs", "pow", "complex", "real", "imag"]))
    kw["polyremoval"] = polyremoval
    got, ref, exact = _run_both(spy.connectivityanalysis, data, ORACLE_CONN, kw)
    if got is not None:
        floor = {"ppc": 5e-6, "corr": 1e-5}.get(method, 1e-6)       # (the floors of these two methods)
        if method == "ppc":
            # the worst of M = (frequencies x channel pairs x trials) single-trial cross spectra comes within ~1/sqrt(M) of
            # zero (|S| is Rayleigh-like: P(|S| < r sigma) ~ r^2).  The float32 transform leaves an ABSOLUTE error of
            # ~5e-7 of the rms bin in every spectrum (the reference transforms in float64 and is exact there), so the
            # phase of that element is off by ~1e-6 sqrt(M), and a trial weighs 2/T in the pair average
            T = len(lengths)
            floor = max(floor, 1e-6 * np.sqrt(ref.data.size / 2 * T) * 2 / T)
        if method == "coh" and kw.get("output") == "imag":
            # Im of a channel's coherence with itself: exactly 0 here, rounding residue of 0 in the reference's
            # complex64 arithmetic (with ONE channel the whole reference result is that residue): not compared
            for arr in (got.data, ref.data) + ((exact.data,) if exact is not None else ()):
                idx = np.arange(arr.shape[-1])
                arr[..., idx, idx] = 0
        _check(got, ref, exact, f"seed {seed}: {kw} lengths {lengths} ch {data.data.shape[1]}", atol_rel=floor,
               scale=_modulus_scale(spy.connectivityanalysis, data, ORACLE_CONN, kw))


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_timefrequency_random_options(seed):
    rng = np.random.default_rng(3000 + OFFSET + seed)
    ragged, polyremoval = bool(rng.integers(0, 2)), [None, 0, 1][int(rng.integers(0, 3))]
    data, lengths = _make(rng, ragged, offsets=polyremoval is not None)
    nmin = min(lengths)
    if rng.integers(0, 2):
        win = int(rng.choice([8, 16, 50, 64, 100, 128, 250, 256]))
        win = max(4, min(win, nmin))
        kw = dict(method="mtmconvol", taper="hann", t_ftimwin=win / 1000.0,
                  toi=[0.0, 0.5, 0.75, "all"][int(rng.integers(0, 4))], output=str(rng.choice(["pow", "abs", "fourier"])))
    else:
        kw = dict(method="wavelet", wavelet="Morlet", width=6, foi=np.sort(rng.uniform(5, 300, size=int(rng.integers(1, 6)))),
                  toi="all", output=str(rng.choice(["pow", "abs", "fourier"])))
    kw["keeptrials"] = bool(rng.integers(0, 2)) and len(set(lengths)) == 1 or True
    kw["polyremoval"] = polyremoval
    got, ref, exact = _run_both(spy.freqanalysis, data, ORACLE_FREQ, kw)
    if got is not None:
        _check(got, ref, exact, f"seed {seed}: {kw} lengths {lengths} ch {data.data.shape[1]}")


@pytest.mark.parametrize("seed", range(32 * SCALE))
def test_mtmfft_selections_and_window_options(seed):
    """foi lists, in-place selections (trials / channels / latency), explicit taper counts, Kaiser windows,
    demean_taper, ft_compat, padding in seconds, and channels whose offset is 100 x their fluctuations (constant
    detrending then lives or dies by the ORDER of the float32 mean)."""
    rng = np.random.default_rng(4000 + OFFSET + seed)
    ragged = bool(rng.integers(0, 2))
    polyremoval = [0, 0, 1, None][int(rng.integers(0, 4))]
    data, lengths = _make(rng, ragged, offsets=False)
    if polyremoval == 0:
        data.data[...] += (rng.normal(size=data.data.shape[1]) * 100).astype(np.float32)[None, :]
    elif polyremoval == 1:
        data.data[...] += (rng.normal(size=data.data.shape[1]) * 3).astype(np.float32)[None, :]
    nchan, ntr, nmin = data.data.shape[1], len(lengths), min(lengths)
    kw = dict(method="mtmfft", polyremoval=polyremoval, output=str(rng.choice(["pow", "abs", "fourier"])))
    kw["keeptrials"] = bool(rng.integers(0, 2)) or kw["output"] == "fourier"
    w = int(rng.integers(0, 4))
    if w == 0:
        kw["taper"] = "kaiser"
        kw["taper_opt"] = {"beta": float(rng.choice([2.0, 8.6, 14.0]))}
    elif w == 1:
        kw["tapsmofrq"] = float(rng.choice([4.0, 12.0]))
        kw["nTaper"] = int(rng.integers(1, 5))
        kw["keeptapers"] = bool(rng.integers(0, 2)) and kw["keeptrials"]
    elif w == 2:
        kw["taper"] = str(rng.choice(["hann", "hamming", "blackman", "bartlett"]))
    else:
        kw["taper"] = None
    if kw["output"] == "fourier":
        kw["keeptapers"] = True
    kw["demean_taper"] = bool(rng.integers(0, 2))
    kw["ft_compat"] = bool(rng.integers(0, 2))
    pad = int(rng.integers(0, 3))
    kw["pad"] = ["maxperlen", "nextpow2", float(np.ceil(max(lengths) * 1.25) / 1000.0)][pad]
    if rng.integers(0, 2):
        nyq = 500.0
        kw["foi"] = np.sort(rng.uniform(0, nyq, size=int(rng.integers(1, 12))))
    sel = {}
    if rng.integers(0, 2) and ntr > 1:
        sel["trials"] = sorted(rng.choice(ntr, size=int(rng.integers(1, ntr + 1)), replace=False).tolist())
    if rng.integers(0, 2) and nchan > 1:
        sel["channel"] = rng.choice(nchan, size=int(rng.integers(1, nchan + 1)), replace=False).tolist()
    if rng.integers(0, 3) == 0 and nmin >= 40:
        t0 = float(rng.uniform(0, 0.3 * nmin)) / 1000.0
        sel["latency"] = [t0, t0 + float(rng.uniform(0.3 * nmin, 0.6 * nmin)) / 1000.0]
    if sel:
        kw["select"] = sel
    got, ref, exact = _run_both(spy.freqanalysis, data, ORACLE_FREQ, kw)
    if got is not None:
        _check(got, ref, exact, f"seed {seed}: {kw} lengths {lengths} ch {nchan}")


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_welch_and_superlet_random_options(seed):
    rng = np.random.default_rng(5000 + OFFSET + seed)
    polyremoval = [None, 0, 1][int(rng.integers(0, 3))]
    data, lengths = _make(rng, ragged=False, offsets=polyremoval is not None)
    n = lengths[0]
    if rng.integers(0, 2):
        win = max(8, min(int(rng.choice([32, 100, 128, 256, 500])), n))
        kw = dict(method="welch", t_ftimwin=win / 1000.0, toi=float(rng.choice([0.0, 0.25, 0.5])),
                  keeptrials=bool(rng.integers(0, 2)))
        if rng.integers(0, 2):
            kw["taper"] = "hann"
        else:
            kw["tapsmofrq"] = float(rng.choice([10.0, 25.0]))
    else:
        kw = dict(method="superlet", order_max=int(rng.integers(2, 6)), order_min=1, c_1=int(rng.integers(1, 4)),
                  adaptive=bool(rng.integers(0, 2)), foi=np.sort(rng.uniform(20, 300, size=int(rng.integers(2, 5)))),
                  toi="all", output=str(rng.choice(["pow", "abs"])), keeptrials=True)
    kw["polyremoval"] = polyremoval
    got, ref, exact = _run_both(spy.freqanalysis, data, ORACLE_FREQ, kw)
    if got is not None:
        # superlets: a root of a small modulus amplifies the transform's absolute error (floor 5e-6)
        _check(got, ref, exact, f"seed {seed}: {kw} lengths {lengths} ch {data.data.shape[1]}",
               atol_rel=5e-6 if kw["method"] == "superlet" else ATOL_REL)


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_connectivity_selections_and_spectral_input(seed):
    """coh / csd / ppc with in-place selections, foi / foilim, keeptrials, offsets of 100 standard deviations under
    constant detrending, and the SpectralData route (freqanalysis(output='fourier', keeptapers=True) chained into
    connectivityanalysis, optionally with channelcmb=[senders, receivers])."""
    rng = np.random.default_rng(6000 + OFFSET + seed)
    polyremoval = [0, 0, 1, None][int(rng.integers(0, 4))]
    method = str(rng.choice(["coh", "csd", "ppc"]))
    # coherency and ppc are RATIOS: where the power of a channel is small at some frequency (with two trials and one
    # taper that happens by chance somewhere on a 3000-bin axis) the float32 transform's absolute error - 5e-7 of the rms
    # bin - becomes a relative one the reference's float64 transform does not have; at least four trials keep the
    # smallest auto-spectrum of a case within the criterion (the production shapes average 7000 products)
    data, lengths = _make(rng, ragged=False, offsets=False, min_trials=5 if method == "ppc" else 4)
    nchan, ntr, n = data.data.shape[1], len(lengths), lengths[0]
    if polyremoval is not None:
        data.data[...] += (rng.normal(size=nchan) * (100 if polyremoval == 0 else 3)).astype(np.float32)[None, :]
    taperkw = dict(tapsmofrq=float(rng.choice([4.0, 10.0]))) if rng.integers(0, 2) else dict(taper="hann")
    kw = dict(method=method)
    if method == "coh":
        kw["output"] = str(rng.choice(["abs", "pow", "complex", "real"]))
    if rng.integers(0, 2):
        kw["foilim"] = [float(rng.uniform(0, 100)), float(rng.uniform(120, 500))]
    spectral = bool(rng.integers(0, 2)) and n >= 16
    floor = 5e-6 if method == "ppc" else 1e-6
    if method == "ppc":
        floor = max(floor, 1e-6 * np.sqrt((n // 2 + 1) * nchan * nchan / 2 * ntr) * 2 / ntr)
    what = f"seed {seed}: {kw} {taperkw} spectral={spectral} polyremoval={polyremoval} lengths {lengths} ch {nchan}"
    if spectral:
        fkw = dict(method="mtmfft", output="fourier", keeptapers=True, keeptrials=True, polyremoval=polyremoval, **taperkw)
        spec_g = spy.freqanalysis(data, **fkw)
        spec_o = spy.freqanalysis(data, compute_method="sequential", routine_classes=ORACLE_FREQ, **fkw)
        if rng.integers(0, 2) and nchan >= 3 and method != "ppc":
            k = int(rng.integers(1, nchan))
            perm = rng.permutation(nchan)
            kw["channelcmb"] = [perm[:k].tolist(), perm[k:].tolist()]
        got = spy.connectivityanalysis(spec_g, **kw)
        ref = spy.connectivityanalysis(spec_o, compute_method="sequential", routine_classes=ORACLE_CONN, **kw)
        assert got.data.shape == ref.data.shape and got.data.dtype == ref.data.dtype, what
        exact = None
        if polyremoval == 1:
            keep = O.detrend
            O.detrend = _detrend_exact
            try:
                spec_x = spy.freqanalysis(data, compute_method="sequential", routine_classes=ORACLE_FREQ, **fkw)
            finally:
                O.detrend = keep
            exact = spy.connectivityanalysis(spec_x, compute_method="sequential", routine_classes=ORACLE_CONN, **kw)
        _check(got, ref, exact, what, atol_rel=floor)
        return
    kw.update(taperkw)
    kw["polyremoval"] = polyremoval
    kw["pad"] = ["maxperlen", "nextpow2"][int(rng.integers(0, 2))]
    sel = {}
    if rng.integers(0, 2) and method != "ppc":
        sel["trials"] = sorted(rng.choice(ntr, size=int(rng.integers(1, ntr + 1)), replace=False).tolist())
    if rng.integers(0, 2) and nchan > 2:
        sel["channel"] = sorted(rng.choice(nchan, size=int(rng.integers(2, nchan + 1)), replace=False).tolist())
    if sel:
        kw["select"] = sel
    got, ref, exact = _run_both(spy.connectivityanalysis, data, ORACLE_CONN, kw)
    if got is not None:
        _check(got, ref, exact, what, atol_rel=floor)


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_granger_random_networks(seed):
    """Granger causality of random AR(2) networks (2 ... 6 channels, 30 ... 60 trials): against the oracle's Wilson
    factorisation at the tolerance of tests/test_gpu_golden.py::test_conn5_granger (the reference's own: atol 1e-2)."""
    rng = np.random.default_rng(7000 + OFFSET + seed)
    nchan = int(rng.integers(2, 7))
    adj = np.zeros((nchan, nchan))
    for _ in range(int(rng.integers(1, nchan + 1))):
        i, j = rng.choice(nchan, size=2, replace=False)
        adj[i, j] = float(rng.uniform(0.1, 0.3))
    data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=int(rng.choice([500, 1000, 1024])), nTrials=int(rng.integers(30, 60)),
                                     seed=int(rng.integers(1, 10000)))
    kw = dict(method="granger", tapsmofrq=float(rng.choice([3.0, 5.0])))
    if rng.integers(0, 2):
        kw["pad"] = "nextpow2"
    got, ref, _ = _run_both(spy.connectivityanalysis, data, ORACLE_CONN, kw)
    if got is None:                            # both sides refused with the same error (a CSD that is not positive definite:
        return                                 # np.linalg.LinAlgError on both, wilson_sf.py:76)
    # Both sides stop iterating when the reconstruction error max|S - psi psi^H| / |S| falls below rtol = 5e-6; at that
    # point the estimate still sits 1e-3 ... 3e-3 from the fully converged factorisation (measured: the oracle's loop
    # run to 1e-12 on the same cross-spectral matrix), and the two sides - complex128 kernels on an exactly Hermitian
    # accumulator vs NumPy on a complex64 matrix that is Hermitian only to rounding, whose error stalls at ~1e-5 -
    # leave the loop in different states: over 32 random networks they agree to 5.6e-3 at worst (the reference's own acceptance tolerance is 1e-2,
    # tests/test_connectivity.py:149).  The two bins next to DC belong to a detrended spectrum (S(0) ~ 0), where the
    # factorisation converges last: held to 0.1 (0.064 at worst over 160 networks).
    assert got.info["reg. factor"] == ref.info["reg. factor"]
    if ref.info["reg. factor"] == -1:          # no regularisation brings the condition number under cond_max: both sides
        assert not got.info["converged"]       # say so (wilson_sf.py:197-254) and neither result means anything
        return
    if not ref.info["converged"] and not got.info["converged"]:
        return                                 # the error test passes on neither side within nIter = 100 (a regularised matrix
                                               # whose DC bin is all regularisation: family 0, seed 274) and both say so
    # (the ORACLE's loop alone failing to report convergence is common - its error stalls just above rtol on a complex64
    # matrix that is Hermitian only to rounding, about a quarter of these networks - and its estimate is compared as it is)
    assert got.info["converged"]
    # The DC bin is not compared: method="granger" demeans every TAPERED trial (demean_taper, connectivity_analysis.py:864),
    # so X_k(0) = 0 and S(0) = 0 in exact arithmetic - what either side factorises there is the rounding residue
    # of its own transform (1e-9 of the spectrum in the reference's float64 FFT rounded to complex64, 1e-8 in float32
    # kernels), and ln(S_ii / (S_ii - ...)) of a residue matrix is a different number on each side, up to ~0.15 (family
    # 500000, seed 94: 0.152 against 0.050 for one pair, the oracle run on to rtol / 100 keeps its 0.050 - it is not the
    # stopping point).  Finite is all that can be said (family 0, seed 274: above 1 on one side).
    assert np.isfinite(got.data).all()
    # End to end the two sides may differ by more than the fixed tolerances without either stage being off: the Wilson
    # factorisation couples all frequencies, S(0) is a zero matrix made of each side's own rounding residue, and the
    # estimate at the bins next to DC moves with it (family 3300000, seed 261: 0.1015 against 0.0891 at 6 Hz for cross-
    # spectral matrices that agree to 0.035 of the criterion; the product's AV kernels on the ORACLE's matrix give the
    # oracle's 0.08911, the oracle's factorisation run to 1e-13 as well).  So the stages are held to parity one by one -
    #   ST: the product's cross-spectral matrix against the oracle's under the shared criterion;
    #   AV: the product's Granger kernels against the oracle's factorisation ON THE SAME (the product's) matrix;
    # - and the end-to-end difference to what the reference's OWN estimate moves by when it is handed the product's
    # matrix instead of its own (`moved`), on top of the fixed tolerances.
    csd_got = _product_granger_csd(data, kw)
    csd_ref = _oracle_granger_csd(data, kw)
    _assert = np.testing.assert_allclose
    tolc = RTOL * np.abs(csd_ref) + ATOL_REL * np.abs(csd_ref).max()
    assert (np.abs(csd_got - csd_ref) <= tolc).all(), f"seed {seed} {kw}: ST stage {float((np.abs(csd_got - csd_ref) / tolc).max()):.3g}"
    on_got, _ = O.granger_cF(csd_got[None])
    on_ref, _ = O.granger_cF(csd_ref[None])
    # (both loops stop at the first error below rtol = 5e-6 and may leave one iteration apart: the bin next to DC, where the
    # factorisation converges last, is held to 0.1, the others to the reference's own acceptance tolerance, atol 1e-2,
    # tests/test_connectivity.py:149)
    _assert(got.data[:, 1:], on_got[:, 1:], atol=0.1, err_msg=f"seed {seed} {kw}: AV stage on the same matrix")
    _assert(got.data[:, 2:], on_got[:, 2:], rtol=2e-3, atol=1e-2, err_msg=f"seed {seed} {kw}: AV stage on the same matrix")
    moved = np.abs(on_got - on_ref)
    err = np.abs(np.asarray(got.data, dtype=np.float64) - np.asarray(ref.data, dtype=np.float64))
    assert (err[:, 1:] <= 0.1 + 2 * moved[:, 1:]).all(), f"seed {seed} {kw}"
    tol = 2e-3 * np.abs(ref.data) + 1e-2 + 2 * moved
    assert (err[:, 2:] <= tol[:, 2:]).all(), f"seed {seed} {kw}: {float((err - tol)[:, 2:].max()):.3g} over"


def _granger_st_options(data, kw):
    """nSamples, taper options of the ST stage of method='granger' as connectivityanalysis derives them
    (connectivity_analysis.py:540-575: process_padding, process_taper on the mean trial length)."""
    from syncopy_amd.shared.input_processors import process_padding, process_taper
    fs = float(data.samplerate)
    lens = np.diff(data.trialdefinition[:, :2]).squeeze(axis=1)
    nS = process_padding(kw.get("pad", "maxperlen"), lens, fs)
    freqs = np.fft.rfftfreq(nS, 1 / fs)
    taper, topt = process_taper("hann", None, kw["tapsmofrq"], None, keeptapers=False, foimax=freqs.max(), samplerate=fs,
                                nSamples=lens.mean(), output="pow")
    return int(nS), taper, topt, fs


def _product_granger_csd(data, kw):
    """The product's trial-averaged cross-spectral matrix as its batched route forms it for method='granger'
    (demean_taper, polyremoval=0): transforms, K4, scale + mirror - (F, C, C) complex64 on the host."""
    import torch
    from syncopy_amd import backend as be
    from syncopy_amd.specest import hip_spectral as hs
    from syncopy_amd.datatype import device_rows
    nS, taper, topt, _ = _granger_st_options(data, kw)
    dev, rows = data.device_data(), device_rows(data)
    C = dev.shape[1]
    acc = torch.zeros((nS // 2 + 1, C, C), dtype=torch.complex64, device=dev.device)
    K = 1
    for _, spec in hs.run_mtmfft_batches(dev, rows, None, nS, taper, topt, True, False, 0, None, "fourier", True, reuse=True):
        be.csd_accumulate(spec, acc)
        K = spec.shape[1]
    be.csd_finalize(acc, 1.0 / (K * len(rows)))
    return acc.cpu().numpy()


def _oracle_granger_csd(data, kw):
    """The same matrix in the reference's arithmetic: cross_spectra_cF per trial, complex64 sequential trial sum."""
    nS, taper, topt, fs = _granger_st_options(data, kw)
    acc = None
    for t in data.trials:
        r, _ = O.cross_spectra_cF(np.array(t), samplerate=fs, nSamples=nS, foi=None, taper=taper, taper_opt=topt, demean_taper=True,
                                  polyremoval=0)
        acc = r if acc is None else acc.__iadd__(r)
    return (acc / np.float32(len(data.trials)))[0].astype(np.complex64)


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_timefrequency_toi_foi_offsets(seed):
    """mtmconvol / wavelet with time-of-interest arrays (regular and irregular, relative to a trial offset), foi lists,
    multitaper windows with keeptapers, trial averages of equal-length trials, trials that start before time zero."""
    rng = np.random.default_rng(8000 + OFFSET + seed)
    polyremoval = [None, 0, 1][int(rng.integers(0, 3))]
    nchan = int(rng.choice([1, 2, 5, 8, 17]))
    ntr = int(rng.integers(2, 5))
    n = int(rng.choice([300, 512, 1000, 1500, 2048]))
    x = rng.normal(size=(ntr * n, nchan)).astype(np.float32)
    if polyremoval is not None:
        x += (rng.normal(size=nchan) * 20).astype(np.float32)[None, :]
    pre = int(rng.choice([0, n // 4, n // 2]))                        # samples before time zero
    trl = np.stack([np.arange(ntr) * n, np.arange(1, ntr + 1) * n, np.full(ntr, -pre)], axis=1)
    data = spy.AnalogData(x, samplerate=1000.0, trialdefinition=trl)
    t0, t1 = -pre / 1000.0, (n - pre - 1) / 1000.0
    kind = int(rng.integers(0, 3))
    if kind == 0:
        toi = "all"
    elif kind == 1:
        step = float(rng.choice([0.005, 0.01, 0.05]))
        toi = np.arange(t0 + 0.02, t1 - 0.02, step)
    else:
        toi = np.sort(rng.uniform(t0, t1, size=int(rng.integers(1, 9))))
    if rng.integers(0, 2):
        win = int(rng.choice([32, 64, 100, 200, 256]))
        kw = dict(method="mtmconvol", t_ftimwin=win / 1000.0, toi=toi if kind else float(rng.choice([0.0, 0.5, 0.8])))
        if rng.integers(0, 2):
            kw["taper"] = "hann"
        else:
            kw["tapsmofrq"] = float(rng.choice([10.0, 20.0]))
            kw["keeptapers"] = bool(rng.integers(0, 2))
        kw["output"] = "fourier" if kw.get("keeptapers") else str(rng.choice(["pow", "abs"]))
        if rng.integers(0, 2):
            kw["foi"] = np.sort(rng.uniform(5, 450, size=int(rng.integers(1, 7))))
    else:
        kw = dict(method="wavelet", wavelet="Morlet", width=float(rng.choice([4, 6, 8])), toi=toi,
                  foi=np.sort(rng.uniform(8, 350, size=int(rng.integers(1, 7)))), output=str(rng.choice(["pow", "abs", "fourier"])))
    kw["keeptrials"] = bool(rng.integers(0, 2)) or kw.get("keeptapers", False)
    kw["polyremoval"] = polyremoval
    got, ref, exact = _run_both(spy.freqanalysis, data, ORACLE_FREQ, kw)
    if got is not None:
        _check(got, ref, exact, f"seed {seed}: {kw} n {n} trials {ntr} ch {nchan} pre {pre}")
        np.testing.assert_allclose(got.trialdefinition, ref.trialdefinition)
        np.testing.assert_allclose(got.freq, ref.freq)


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_corr_and_jackknife_random_options(seed):
    rng = np.random.default_rng(9000 + OFFSET + seed)
    polyremoval = [None, 0, 1][int(rng.integers(0, 3))]
    nchan = int(rng.choice([2, 3, 6, 11]))
    n = int(rng.choice([100, 257, 500, 1000, 1024]))
    ntr = int(rng.integers(6, 12))
    x = rng.normal(size=(ntr * n, nchan)).astype(np.float32)
    x[:, 1] += 0.5 * np.roll(x[:, 0], 3)                                   # a lagged coupling
    if polyremoval is not None:
        x += (rng.normal(size=nchan) * 10).astype(np.float32)[None, :]
    trl = np.stack([np.arange(ntr) * n, np.arange(1, ntr + 1) * n, np.zeros(ntr)], axis=1)
    data = spy.AnalogData(x, samplerate=1000.0, trialdefinition=trl)
    if rng.integers(0, 2):
        kw = dict(method="corr", keeptrials=bool(rng.integers(0, 2)), polyremoval=polyremoval)
        got, ref, exact = _run_both(spy.connectivityanalysis, data, ORACLE_CONN, kw)
        if got is not None:
            _check(got, ref, exact, f"seed {seed}: {kw} n {n} trials {ntr} ch {nchan}", atol_rel=1e-5)
        return
    kw = dict(method="coh", tapsmofrq=float(rng.choice([5.0, 10.0])), jackknife=True, polyremoval=polyremoval,
              output=str(rng.choice(["abs", "pow"])))
    got, ref, exact = _run_both(spy.connectivityanalysis, data, ORACLE_CONN, kw)
    if got is None:
        return
    from parity import jackknife_tolerances
    _check(got, ref, exact, f"seed {seed}: {kw} n {n} trials {ntr} ch {nchan}")
    if exact is None:                   # (a float32 least-squares fit moves every replicate: compared for 0 / None only)
        tol_var, tol_bias = jackknife_tolerances(ref.data, ref.jack_var, T=ntr)
        ev = np.abs(np.asarray(got.jack_var, dtype=np.float64) - ref.jack_var) / tol_var
        eb = np.abs(np.asarray(got.jack_bias, dtype=np.float64) - ref.jack_bias) / (1e-5 * np.abs(ref.jack_bias) + tol_bias)
        assert ev.max() <= 1.0 and eb.max() <= 1.0, (seed, kw, float(ev.max()), float(eb.max()))


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_reference_precision_random_lengths(seed):
    """precision="reference" (float64 taper product and transform, complex64 rounding where mtmfft.py:104-127 rounds) at
    random lengths - powers of two (radix-16 register kernel up to 4096) and anything else without a prime factor above 61
    (generic Stockham passes) - on data with 60 dB of dynamic range: the criterion everywhere, and bin by bin PURE
    rtol 1e-5 on at least 99 % of the bins, which the float32 kernels cannot give (~5 % there)."""
    rng = np.random.default_rng(10000 + OFFSET + seed)
    n = int(rng.choice([100, 250, 256, 360, 500, 729, 1000, 1024, 1500, 2000, 2048, 2500, 3000, 4096, 5000, 6000, 8192]))
    nchan, ntr = int(rng.choice([1, 2, 3, 8])), int(rng.integers(1, 4))
    t = np.arange(n * ntr) / 1000.0
    x = rng.normal(size=(n * ntr, nchan)) + 1000.0 * np.sin(2 * np.pi * 40.0 * t)[:, None] + 50.0
    trl = np.stack([np.arange(ntr) * n, np.arange(1, ntr + 1) * n, np.zeros(ntr)], axis=1)
    data = spy.AnalogData(x.astype(np.float32), samplerate=1000.0, trialdefinition=trl)
    kw = dict(method="mtmfft", output="fourier", keeptapers=True, keeptrials=True, polyremoval=0)
    if rng.integers(0, 2):
        kw["taper"] = "hann"
    else:
        kw["tapsmofrq"] = float(rng.choice([3.0, 8.0]))
    got = spy.freqanalysis(data, precision="reference", **kw)
    ref = spy.freqanalysis(data, compute_method="sequential", routine_classes=ORACLE_FREQ, **kw)
    _check(got, ref, None, f"seed {seed}: n {n} ch {nchan} {kw}")
    err = np.abs(got.data.astype(np.complex128) - ref.data)
    frac = float((err <= 1e-5 * np.abs(ref.data)).mean())
    assert frac >= 0.99, (seed, n, frac)
