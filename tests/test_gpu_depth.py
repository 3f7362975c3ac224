"""Parity at the PRODUCTION ACCUMULATION DEPTH ("What's weak" 1 / "Next round" 4).

K4 is benchmarked on 1000 trials x 7 tapers = 7000 rows x 2049 frequencies x 256 channels; the other parity tests stop at
300 rows.  Here the same launch shape is checked against complex128 products of the same spectra, for uncorrelated rows
AND for strongly coherent channel pairs near zero lag - the case in which the 3-multiplication kernels' imaginary part
(im = P3 - P1 + P2, three independently rounded 7000-term fp32 sums) is far less accurate than the reference's directly
summed complex64 products (connectivity/csd.py:94-102,164-168).  Consequences pinned here:
  * complex values, moduli, real parts: default kernels within the standard criterion at full depth;
  * imaginary part / phase: only the phase-exact (4-multiplication) kernels are within it, and those are what
    `connectivityanalysis(output="imag" | "angle")` runs (connectivity_analysis._run_stages)."""
import numpy as np
import pytest

import syncopy_amd as spy
from oracle_routines import ORACLE_CONN
from parity import assert_parity, excess

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

FSEL = (0, 1, 2, 255, 256, 257, 1023, 1024, 1025, 1500, 2040, 2044, 2045, 2046, 2047, 2048)

# Imaginary part / phase of a coherency are projections of a complex number of modulus <= 1: the absolute floor of the
# criterion refers to THAT scale, not to the largest imaginary part (which is ~0.007 here - a floor of 1e-6 of it, 7e-9,
# is below the rounding noise of any float32 accumulation over 7000 rows, the reference's own complex64 sums included:
# half an ulp of the partial sums times sqrt(7000) is ~5e-5 on |S| ~ 7600, i.e. ~7e-9 of the normalisation).
IMAG_ATOL = 1e-7


def excess_abs(a, b, rtol=1e-5, atol=IMAG_ATOL):
    err = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    assert np.isfinite(err).all()
    return float((err / (rtol * np.abs(b) + atol)).max())


@pytest.fixture(scope="module")
def be():
    from syncopy_amd import backend
    backend.require_gpu()
    return backend


def _spectra(kind, R=7000, F=2049, C=256):
    """(R, F, C) complex64 on the device.  'noise': independent rows (what bench.py's synthetic AR(2) trials give);
    'coherent': one common source per (row, frequency) seen by every channel through a gain exp(i d_c), |d_c| <= 1e-3
    rad, + 30 % independent noise: every pair has coherence ~0.9 and a phase of at most 2e-3 rad."""
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((R, F, C, 2), generator=g, device="cuda", dtype=torch.float32)
    if kind == "coherent":
        x *= 0.3
        src = torch.randn((R, F, 1, 2), generator=g, device="cuda", dtype=torch.float32)
        d = (torch.rand((C,), generator=g, device="cuda", dtype=torch.float64) * 2 - 1) * 1e-3
        gain = torch.stack((torch.cos(d), torch.sin(d)), dim=-1).to(torch.float32)        # (C, 2)
        x[..., 0] += src[..., 0] * gain[:, 0] - src[..., 1] * gain[:, 1]
        x[..., 1] += src[..., 0] * gain[:, 1] + src[..., 1] * gain[:, 0]
        del src
    return torch.view_as_complex(x)


def _reference(spec, f):
    x = spec[:, f, :].to(torch.complex128)
    return (x.T @ x.conj()).cpu().numpy()          # sum over rows of X_i conj(X_j), complex128


@pytest.mark.parametrize("kind", ["noise", "coherent"])
def test_csd_accumulate_at_production_depth(be, kind):
    """One launch of the bench shape (7000 x 2049 x 256, `csdh_kernel` + row-split tail) against complex128
    products at 16 frequencies incl. 0, 2047 and 2048 (the tail)."""
    spec = _spectra(kind)
    R, F, C = spec.shape
    acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(spec, acc)
    il = np.tril_indices(C)
    worst = 0.0
    for f in FSEL:
        ref = _reference(spec, f)
        got = acc[f].cpu().numpy()
        e = excess(got[il], ref[il].astype(np.complex64))
        worst = max(worst, e)
        assert e <= 1.0, f"{kind}: csd at depth {R}, f={f}: max err/tol = {e:.3g}"
        # real part and modulus on their own (what abs / pow / real coherence are made of)
        assert_parity(got.real[il], ref.real[il].astype(np.float32), what=f"{kind} re f={f}")
        assert_parity(np.abs(got[il]), np.abs(ref[il]).astype(np.float32), what=f"{kind} |S| f={f}")
    print(f"[depth] {kind}: default kernels, complex criterion, worst err/tol = {worst:.3g}")
    # coherence outputs that do not isolate the imaginary part, straight from the raw accumulator
    coh = be.coh_from_accumulator(acc, 1.0 / R, "abs")
    for f in FSEL:
        ref = _reference(spec, f)
        d = np.sqrt(np.real(np.diag(ref)))
        assert_parity(coh[f].cpu().numpy(), (np.abs(ref) / np.outer(d, d)).astype(np.float32), what=f"{kind} coh f={f}")


@pytest.mark.parametrize("C", [256, 128])
def test_imaginary_part_needs_phase_exact_kernels(be, C):
    """Strongly coherent pairs near zero lag at full depth: the imaginary part of the float32 3-multiplication
    accumulator (every channel count but 256) misses the criterion by orders of magnitude, the phase-exact kernels meet
    it - which is why the front ends select them for output='imag' / 'angle'.  256 channels run the half-precision
    kernel (csdh_kernel.h), a 4-multiplication product whose imaginary part is summed directly: it meets the criterion
    in either mode.  If the 3M kernels ever pass here the routing can go."""
    spec = _spectra("coherent", C=C, F=2049 if C == 256 else 1025)
    R, F, C = spec.shape
    il = np.tril_indices(C, -1)
    res = {}
    for exact in (False, True):
        acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
        with be.csd_phase_exact(exact):
            be.csd_accumulate(spec, acc)
        worst = 0.0
        for f in (f for f in FSEL if f < F):
            ref = _reference(spec, f)
            d = np.sqrt(np.real(np.diag(ref)))
            cref = ref / np.outer(d, d)
            for output, r in (("imag", cref.imag), ("angle", np.angle(cref))):
                got = be.coh_from_accumulator(acc, 1.0 / R, output)[f].cpu().numpy()
                worst = max(worst, excess_abs(got[il], r[il]))
        res[exact] = worst
        del acc
    print(f"[depth] {C} channels, coherence imag / angle of coherent pairs (|d| <= 1e-5 |b| + {IMAG_ATOL:g}): default kernels "
          f"err/tol = {res[False]:.3g}, phase-exact kernels {res[True]:.3g}")
    assert res[True] <= 1.0, res
    if C == 256:
        assert res[False] <= 1.0, ("the half-precision kernel sums the imaginary part directly", res)
        assert be.csd_split_fallbacks() == 0, "no frequency of this data should need the float32 kernels"
    else:
        assert res[False] > 1.0, ("the 3-multiplication kernels now meet the criterion on imaginary parts: "
                                  "drop the phase-exact routing", res)


def test_coherence_imag_angle_256_channels_vs_oracle():
    """256 channels, every pair strongly coupled at a lag of a small fraction of a sample, through
    spy.connectivityanalysis(output='imag' | 'angle' | 'abs' | 'complex') against the oracle's complex64 arithmetic
    (csd.py:94-102 per trial, sequential complex64 trial sum, normalize_csd csd.py:118-172)."""
    rng = np.random.default_rng(5)
    T, N, C = 24, 128, 256
    a = rng.uniform(-0.02, 0.02, size=C)
    trials = []
    for _ in range(T):
        s = rng.normal(size=N + 1)
        x = s[1:, None] + a[None, :] * s[:-1, None] + 0.2 * rng.normal(size=(N, C))
        trials.append(x.astype(np.float32))
    k = np.arange(T, dtype=float)[:, None] * N
    data = spy.AnalogData(np.concatenate(trials), samplerate=1000.0, trialdefinition=np.hstack((k, k + N, 0 * k)))
    for output in ("imag", "angle", "abs", "complex"):
        got = spy.connectivityanalysis(data, method="coh", tapsmofrq=20, output=output)
        ref = spy.connectivityanalysis(data, method="coh", tapsmofrq=20, output=output, compute_method="sequential",
                                       routine_classes=ORACLE_CONN)
        assert got.data.shape == ref.data.shape == (1, N // 2 + 1, C, C)
        if output == "angle":
            # phases of pairs whose coherency is real up to rounding flip between 0 and +-pi on either side: compare
            # on the circle, and strictly where the reference's imaginary part is not rounding noise
            dphi = np.angle(np.exp(1j * (got.data.astype(np.float64) - ref.data)))
            assert np.abs(dphi).max() < 2e-5, np.abs(dphi).max()
        else:
            assert_parity(got.data, ref.data, what=f"256-channel coherence, output={output}")


# ---------------------------------------------------------------------------------------------------------------
# Beyond one launch: 17 920 rows (configs[4]'s share of one of eight GPUs at the 7 tapers of its taper setting ... the
# depth test_c5_granger_front_end_256x4096 pushes through K4h) and 56 000 rows (configs[4] on ONE GPU: 8000 trials x 7
# tapers), accumulated launch by launch like the product does (batches of 1000 trials = 7000 rows into ONE float32
# accumulator).  Three sums against complex128 at 16 frequencies:
#   K4h            fp16 (hi, lo) operands, float32 accumulation                                   (the default)
#   K4 float32     csd3m_kernel, float32 operands and accumulation                                (split=False)
#   reference      the oracle's arithmetic restated on the device: per trial the taper mean of complex64 outer products
#                  (csd.py:94-102), trials summed one after the other in complex64
#                  (computational_routine.py:1022-1032)
# so that the table says where the reference's OWN float32 error sits at that depth.  VERDICT r5 "weak" 2.
DEPTH_TABLE = {}


def _reference_c64_sum(spec, ntaper, fsel, running):
    """running[f] += per-trial taper means of complex64 outer products, trial after trial (float32 arithmetic)."""
    R, F, C = spec.shape
    x = spec[:, list(fsel), :].reshape(R // ntaper, ntaper, len(fsel), C)               # (trials, K, nf, C) complex64
    inv = 1.0 / ntaper
    for t in range(x.shape[0]):
        xt = x[t]                                                                        # (K, nf, C)
        outer = xt[:, :, :, None] * xt[:, :, None, :].conj()                             # (K, nf, C, C) complex64 products
        running += outer.sum(dim=0) * inv                                                # taper mean, then the trial sum
    return running


@pytest.mark.parametrize("kind", ["noise", "coherent"])
@pytest.mark.parametrize("rows", [17920, 56000])
def test_csd_accumulate_beyond_one_launch(be, rows, kind):
    F, C, K = 2049, 256, 7
    il = np.tril_indices(C)
    acc_h = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    acc_f = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    run_ref = torch.zeros((len(FSEL), C, C), dtype=torch.complex64, device="cuda")
    ref128 = torch.zeros((len(FSEL), C, C), dtype=torch.complex128, device="cuda")
    done, batch = 0, 0
    while done < rows:
        nb = min(7000, rows - done)
        g = torch.Generator(device="cuda").manual_seed(1000 + batch)
        x = torch.randn((nb, F, C, 2), generator=g, device="cuda", dtype=torch.float32)
        if kind == "coherent":
            x *= 0.3
            src = torch.randn((nb, F, 1, 2), generator=g, device="cuda", dtype=torch.float32)
            d = (torch.rand((C,), generator=torch.Generator(device="cuda").manual_seed(5), device="cuda", dtype=torch.float64) * 2 - 1) * 1e-3
            gain = torch.stack((torch.cos(d), torch.sin(d)), dim=-1).to(torch.float32)
            x[..., 0] += src[..., 0] * gain[:, 0] - src[..., 1] * gain[:, 1]
            x[..., 1] += src[..., 0] * gain[:, 1] + src[..., 1] * gain[:, 0]
            del src
        spec = torch.view_as_complex(x)
        be.csd_accumulate(spec, acc_h)
        assert be.csd_split_fallbacks() == 0
        be.csd_accumulate(spec, acc_f, split=False)
        _reference_c64_sum(spec, K, FSEL, run_ref)
        for i, f in enumerate(FSEL):
            xf = spec[:, f, :].to(torch.complex128)
            ref128[i] += xf.T @ xf.conj()
        del spec, x
        done += nb
        batch += 1
    worst = {"k4h": 0.0, "k4_f32": 0.0, "reference_c64": 0.0}
    for i, f in enumerate(FSEL):
        ref = ref128[i].cpu().numpy()
        refK = ref / K                                   # the reference sums taper MEANS
        worst["k4h"] = max(worst["k4h"], excess(acc_h[f].cpu().numpy()[il], ref[il].astype(np.complex64)))
        worst["k4_f32"] = max(worst["k4_f32"], excess(acc_f[f].cpu().numpy()[il], ref[il].astype(np.complex64)))
        worst["reference_c64"] = max(worst["reference_c64"], excess(run_ref[i].cpu().numpy()[il], refK[il].astype(np.complex64)))
    DEPTH_TABLE[(kind, rows)] = worst
    print(f"[depth] {kind}, {rows} rows x {F} x {C}: err/tol against complex128: K4h {worst['k4h']:.3g}, float32 K4 "
          f"{worst['k4_f32']:.3g}, the reference's complex64 sequential sum {worst['reference_c64']:.3g}")
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "depth_table.json"), "w") as fh:
        json.dump({"%s/%d" % k: v for k, v in DEPTH_TABLE.items()}, fh, indent=1)
    # the product must not be worse than the criterion - or, where the reference's own float32 sum already exceeds it,
    # not worse than the reference
    assert worst["k4h"] <= max(1.0, worst["reference_c64"]), worst
    assert worst["k4_f32"] <= max(1.0, worst["reference_c64"]), worst
