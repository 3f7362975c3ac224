"""Edge cases through the front ends (GPU kernels vs the oracle bound to the same front ends): trials of a handful of
samples, one channel, one trial, windows and kernels at the limits of a trial - the shapes a Syncopy user can pass
and the reference's own tests probe (tests/test_specest.py, tests/test_connectivity.py: short trials, single channels,
selections), far from the benchmark shapes the kernels are tuned for."""
import numpy as np
import pytest

import syncopy_amd as spy
from oracle_routines import ORACLE_CONN, ORACLE_FREQ
from parity import assert_parity

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from syncopy_amd import backend
    backend.require_gpu()


def _data(nsamp, nchan, ntrials, seed=0, lengths=None):
    rng = np.random.default_rng(seed)
    lengths = lengths or [nsamp] * ntrials
    x = rng.normal(size=(sum(lengths), nchan)).astype(np.float32)
    x += np.linspace(-1, 2, nchan, dtype=np.float32)[None, :]            # offsets: detrending matters
    edges = np.concatenate([[0], np.cumsum(lengths)])
    trl = np.stack([edges[:-1], edges[1:], np.zeros(len(lengths))], axis=1)
    return spy.AnalogData(x, samplerate=1000.0, trialdefinition=trl)


def _both(fn, data, classes, **kw):
    got = fn(data, **kw)
    ref = fn(data, compute_method="sequential", routine_classes=classes, **kw)
    assert got.data.shape == ref.data.shape and got.data.dtype == ref.data.dtype
    return got, ref


@pytest.mark.parametrize("nsamp", [3, 5, 8, 16, 17, 31, 64, 100, 128, 200, 255, 257])
@pytest.mark.parametrize("output", ["pow", "fourier"])
def test_mtmfft_tiny_trials(nsamp, output):
    """mtmfft on trials of 3 ... 257 samples (specest/mtmfft.py:80-99 takes any nSamples), Hann window, 3 channels."""
    data = _data(nsamp, 3, 4, seed=nsamp)
    got, ref = _both(spy.freqanalysis, data, ORACLE_FREQ, method="mtmfft", taper="hann", output=output, keeptrials=True)
    assert got.data.shape[2] == nsamp // 2 + 1
    assert_parity(got.data, ref.data, what=f"mtmfft {nsamp} samples {output}")


@pytest.mark.parametrize("nchan,ntrials", [(1, 1), (1, 5), (2, 1), (3, 2)])
def test_mtmfft_one_channel_one_trial(nchan, ntrials):
    data = _data(1000, nchan, ntrials, seed=7)
    got, ref = _both(spy.freqanalysis, data, ORACLE_FREQ, method="mtmfft", tapsmofrq=3, output="pow")
    assert_parity(got.data, ref.data, what=f"mtmfft {nchan} ch {ntrials} trials")


def test_mtmfft_ragged_trials_every_padding():
    """trials of 90 ... 1000 samples in one call: pad='maxperlen' / 'nextpow2' / seconds (process_padding)."""
    data = _data(0, 4, 0, seed=3, lengths=[90, 1000, 333, 512, 257])
    for pad in ("maxperlen", "nextpow2", 1.5):
        got, ref = _both(spy.freqanalysis, data, ORACLE_FREQ, method="mtmfft", taper="hann", pad=pad, keeptrials=True)
        assert_parity(got.data, ref.data, what=f"ragged trials pad={pad}")


@pytest.mark.parametrize("nchan", [1, 2])
def test_coherence_and_csd_of_one_and_two_channels(nchan):
    data = _data(500, nchan, 6, seed=11)
    for method, output in (("coh", "abs"), ("coh", "complex"), ("csd", "abs")):
        kw = dict(method=method, tapsmofrq=4)
        if method == "coh":
            kw["output"] = output
        got, ref = _both(spy.connectivityanalysis, data, ORACLE_CONN, **kw)
        assert_parity(got.data, ref.data, what=f"{method} {output} {nchan} ch")


def test_coherence_of_short_trials():
    data = _data(20, 5, 30, seed=2)
    got, ref = _both(spy.connectivityanalysis, data, ORACLE_CONN, method="coh", taper="hann")
    assert_parity(got.data, ref.data, what="coh of 20-sample trials")


def test_granger_two_channels():
    data = spy.synthdata.ar2_network(AdjMat=np.array([[0, 0.3], [0, 0]]), nSamples=800, nTrials=30, seed=4)
    got, ref = _both(spy.connectivityanalysis, data, ORACLE_CONN, method="granger", tapsmofrq=3)
    # the reference's own tolerance for Granger estimates is atol = 1e-2 (tests/test_connectivity.py:149).  The DC bin of this
    # detrended spectrum is rounding noise (S(0) ~ 0), it carries the largest relative error of psi psi^H and so decides at
    # WHICH iteration max_rel_err drops below rtol = 5e-6 (wilson_sf.py:99-103): the error sequence here runs ... 1e-3,
    # ~3e-6, ~1e-11, and whether the middle value lands below rtol (stop) or just above it (one more iteration) depends on the last bits
    # of the transforms - the oracle stops at 7.8e-7, float32 kernels at 5.7e-8 or 2.7e-6 (by kernel), float64 ones at
    # 8.7e-12.  One iteration more or less moves the estimates by up to ~1e-3 (2e-2 in the two bins next to DC), all of it
    # inside the reference's own tolerance; tests/test_gpu_golden.py::test_conn5_granger holds a well-conditioned case to 2e-3
    for prec in ("auto", "reference"):
        g = got if prec == "auto" else spy.connectivityanalysis(data, method="granger", tapsmofrq=3, precision=prec)
        assert g.info["converged"] and g.info["max rel. err"] < 5e-6
        # away from DC: well inside the reference's atol = 1e-2; only the two bins next to DC get the one-iteration bound
        np.testing.assert_allclose(g.data[:, 2:], ref.data[:, 2:], rtol=2e-3, atol=1.5e-3)
        np.testing.assert_allclose(g.data[:, :2], ref.data[:, :2], atol=2.5e-2)


@pytest.mark.parametrize("nsamp", [40, 300])
def test_wavelet_kernels_longer_than_the_trial(nsamp):
    """Morlet kernels of 10 s / dt taps on trials that are shorter than the kernel (transform.py:96-107 convolves in
    full and crops to the signal): 10 Hz -> 966 taps on 40- and 300-sample trials."""
    data = _data(nsamp, 3, 3, seed=nsamp)
    kw = dict(method="wavelet", wavelet="Morlet", width=6, foi=np.array([10.0, 40.0, 120.0]), toi="all", output="pow",
              keeptrials=True)
    got, ref = _both(spy.freqanalysis, data, ORACLE_FREQ, **kw)
    assert_parity(got.data, ref.data, what=f"wavelet on {nsamp}-sample trials")


def test_mtmconvol_window_as_long_as_the_trial():
    data = _data(256, 3, 4, seed=9)
    kw = dict(method="mtmconvol", taper="hann", t_ftimwin=0.256, toi="all", output="pow", keeptrials=True)
    got, ref = _both(spy.freqanalysis, data, ORACLE_FREQ, **kw)
    assert_parity(got.data, ref.data, what="mtmconvol, window = trial")


def test_mtmconvol_tiny_window():
    data = _data(300, 2, 3, seed=10)
    kw = dict(method="mtmconvol", taper="hann", t_ftimwin=0.016, toi=0.5, output="abs", keeptrials=True)
    got, ref = _both(spy.freqanalysis, data, ORACLE_FREQ, **kw)
    assert_parity(got.data, ref.data, what="mtmconvol, 16-sample windows")


@pytest.mark.parametrize("per_trial", [False, True])
def test_mtmconvol_first_window_cut_short_at_the_trial_edge(per_trial):
    """Irregular toi whose first window overhangs the trial: the reference matches the bins on that shortened window
    (compRoutines.py:402-404) and cannot write 117 of them into a block of 129 - NumPy's broadcast ValueError, before any
    later (still shorter) window could run off the frequency axis."""
    data = _data(300, 3, 3, seed=11)
    trl = data.trialdefinition.copy()
    trl[:, 2] = -75                                   # 75 samples before time zero: the first window ends at the trial's end
    data = spy.AnalogData(data.data, samplerate=1000.0, trialdefinition=trl)
    kw = dict(method="mtmconvol", taper="hann", t_ftimwin=0.256, toi=np.array([0.1214, 0.1323, 0.1556, 0.1889]),
              output="abs", keeptrials=True)
    with pytest.raises(ValueError, match="could not broadcast"):
        spy.freqanalysis(data, compute_method="sequential", routine_classes=ORACLE_FREQ, **kw)
    with pytest.raises(ValueError, match="could not broadcast"):
        spy.freqanalysis(data, **({"compute_method": "sequential"} if per_trial else {}), **kw)


def test_dropped_results_release_their_device_memory():
    """A result nobody reads holds its spectra in HBM behind lazy thunks (the trial-averaged CSD: F x C x C complex64).
    Those thunks must not tie the object into a reference cycle: dropping the result has to release the memory at once,
    not whenever Python's cyclic collector runs - a loop over recordings would otherwise grow by a CSD per call."""
    import gc
    import torch
    data = _data(512, 64, 12, seed=21)
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        used = []
        for _ in range(5):
            r = spy.connectivityanalysis(data, method="coh", taper="hann")
            torch.cuda.synchronize()
            del r
            used.append(torch.cuda.memory_allocated())
        assert used[-1] <= used[1], used                  # flat after the first calls (plans and buffers are cached once)
        small = _data(512, 6, 40, seed=22)
        for d, method, kw in ((data, "csd", {"taper": "hann"}), (small, "granger", {"tapsmofrq": 4})):
            u0 = None
            for _ in range(3):
                r = spy.connectivityanalysis(d, method=method, **kw)
                torch.cuda.synchronize()
                del r
                u = torch.cuda.memory_allocated()
                u0 = u if u0 is None else u0
            assert u <= u0, (method, u0, u)
        for kw in (dict(method="mtmfft", taper="hann", keeptrials=True), dict(method="mtmfft", taper="hann", keeptrials=False),
                   dict(method="mtmconvol", taper="hann", t_ftimwin=0.128, toi="all", keeptrials=True),
                   dict(method="wavelet", foi=np.array([20.0, 40.0]), toi="all", keeptrials=False)):
            u0 = None
            for _ in range(3):
                r = spy.freqanalysis(small, output="pow", **kw)
                torch.cuda.synchronize()
                del r
                u = torch.cuda.memory_allocated()
                u0 = u if u0 is None else u0
            assert u <= u0, (kw, u0, u)
    finally:
        if was_enabled:
            gc.enable()
