"""spy.mean (VERDICT r2 missing 4) against vectors written by the real reference
(tests/golden/mean_variants.npz, oracle/gen_golden.py): the oracle through the product's front end on the CPU, the
device kernels on the GPU."""
import os

import numpy as np
import pytest

import syncopy_amd as spy
from oracle import spy_oracle as O
from parity import assert_parity

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mean_variants.npz"))


def _analog():
    return spy.AnalogData(np.concatenate(list(Z["data"])), samplerate=float(Z["samplerate"]), trialdefinition=Z["trialdefinition"])


def _spectral(key):
    s = spy.SpectralData(Z[key], samplerate=float(Z["samplerate"]), trialdefinition=Z["spec_trldef"])
    s.channel = np.array(["channel%d" % (i + 1) for i in range(Z[key].shape[-1])])
    return s


CASES = [("analog", "analog_trials", dict(dim="trials")), ("analog", "analog_time", dict(dim="time")),
         ("analog", "analog_time_avg", dict(dim="time", keeptrials=False)), ("analog", "analog_channel", dict(dim="channel")),
         ("analog", "analog_trials_sel", dict(dim="trials", select={"trials": [0, 2, 3], "channel": [0, 3]})),
         ("spec", "spec_trials", dict(dim="trials")), ("spec", "spec_freq", dict(dim="freq")), ("spec", "spec_taper", dict(dim="taper")),
         ("spec", "spec_channel_avg", dict(dim="channel", keeptrials=False)),
         ("pow", "pow_trials", dict(dim="trials")), ("pow", "pow_freq_avg", dict(dim="freq", keeptrials=False))]


def _run(src, opts, **how):
    data = _analog() if src == "analog" else _spectral(src)
    return spy.mean(data, **opts, **how)


@pytest.mark.parametrize("src,key,opts", CASES)
def test_oracle_mean_matches_reference(src, key, opts):
    out = _run(src, opts, compute_method="sequential", routine_classes=O.STAT_OPS)
    assert out.data.shape == Z[key].shape and out.data.dtype == Z[key].dtype
    assert_parity(out.data, Z[key], what=key)
    if key + "_trldef" in Z.files:
        assert np.array_equal(np.asarray(out.trialdefinition, dtype=float), Z[key + "_trldef"].astype(float))


def test_mean_output_labels_follow_the_reference():
    """statistics/compRoutines.py:131-141: the averaged dimension carries ONE label, the operation's name; a numerical
    frequency axis is gone (None); other dimensions keep their (selected) labels; unsupported selection keys raise."""
    from syncopy_amd.shared.errors import SPYValueError
    how = dict(compute_method="sequential", routine_classes=O.STAT_OPS)
    out = spy.mean(_analog(), dim="channel", **how)
    assert list(out.channel) == ["mean"]
    spec = _spectral("spec")
    spec.freq = np.arange(spec.data.shape[2], dtype=float)
    out = spy.mean(spec, dim="freq", **how)
    assert out.freq is None and list(out.channel) == list(spec.channel)
    out = spy.mean(spec, dim="channel", keeptrials=False, **how)
    assert list(out.channel) == ["mean"] and np.array_equal(out.freq, spec.freq)
    with pytest.raises(SPYValueError):
        spy.mean(spec, dim="trials", select={"frequency": [1, 10]}, **how)


def test_mean_argument_checks():
    from syncopy_amd.shared.errors import SPYTypeError, SPYValueError
    with pytest.raises(SPYValueError):
        spy.mean(_analog(), dim="freq", compute_method="sequential", routine_classes=O.STAT_OPS)
    with pytest.raises(SPYTypeError):
        spy.mean(np.zeros((3, 3)), dim="time")
    uneq = spy.AnalogData(np.zeros((30, 2), np.float32), samplerate=10.0, trialdefinition=np.array([[0, 10, 0], [10, 30, 0]]))
    with pytest.raises(SPYValueError):
        spy.mean(uneq, dim="trials", compute_method="sequential", routine_classes=O.STAT_OPS)


@pytest.mark.gpu
@pytest.mark.parametrize("src,key,opts", CASES)
def test_device_mean_matches_reference(src, key, opts):
    out = _run(src, opts)
    assert out.data.shape == Z[key].shape and out.data.dtype == Z[key].dtype
    if opts["dim"] == "trials":
        assert np.array_equal(out.data, Z[key]), key            # the reference's rounding sequence, bit for bit
    else:
        assert_parity(out.data, Z[key], what=key)
        # NumPy's own summation order along the axis: equal to the last bit up to the final divide
        assert np.abs(out.data - Z[key]).max() <= 2.5e-7 * np.abs(Z[key]).max(), key


@pytest.mark.gpu
def test_device_mean_skips_nans_and_long_axes():
    import torch
    from syncopy_amd import backend
    rng = np.random.default_rng(0)
    for shape, axis in (((5000, 3), 0), ((4, 3000), 1), ((2, 7, 1000), 2), ((6, 130, 2), 1)):
        x = rng.normal(size=shape).astype(np.float32) + 10
        x[tuple(rng.integers(0, s, size=5) for s in shape)] = np.nan
        got = backend.axis_nanmean(torch.from_numpy(x).cuda(), axis).cpu().numpy()
        np.testing.assert_allclose(got, np.nanmean(x, axis=axis, keepdims=True), rtol=3e-7)
        clean = np.nan_to_num(x, nan=1.0)
        got = backend.axis_nanmean(torch.from_numpy(clean).cuda(), axis).cpu().numpy()
        np.testing.assert_allclose(got, np.mean(clean, axis=axis, keepdims=True), rtol=3e-7)
        z = (clean + 1j * clean[::-1].copy()).astype(np.complex64) if axis == 0 else (clean + 1j * clean).astype(np.complex64)
        got = backend.axis_nanmean(torch.from_numpy(z).cuda(), axis).cpu().numpy()
        np.testing.assert_allclose(got, np.mean(z, axis=axis, keepdims=True), rtol=3e-7)
