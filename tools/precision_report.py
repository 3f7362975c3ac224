"""Where the fp32 kernels stand against the reference's float64 taper + FFT: for the c1 / conn5 /
harmonic fixtures and an offset-laden channel set, the distribution of |kernel - reference| relative to the largest
bin of each channel, the fraction of bins outside a PURE rtol = 1e-5, and what the parity criterion
(rtol 1e-5 + 1e-6 max|b|) leaves - with the reference-order float32 mean and with the float64 block sums.

Runs on the CPU through the kernel emulator (tests/emu: the unchanged kernel headers compiled for the host), so the
numbers are those of the device arithmetic (fp32 fma / add order included) without needing a GPU.
    python tools/precision_report.py > profiles/r2_fft_precision.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    sys.path.insert(0, p)
import emu_driver as E  # noqa: E402
from oracle import spy_oracle as O  # noqa: E402


def report(name, x, fs, nfft, taper, topt):
    nsig, C = x.shape
    tap = O.taper_table(taper, nsig, nfft, topt)
    sc = O.spec_scale(nsig, nfft)
    ref, _ = O.mtmfft(O.detrend(x, 0), fs, nfft, taper, topt)
    ref = ref.astype(np.complex128)
    print(f"\n== {name}: {C} ch x {nsig} samples, nfft {nfft}, {tap.shape[0]} taper(s) '{taper}'")
    for label, kw in (("reference-order float32 mean", dict(reference_mean=True)), ("float64 block sums", {})):
        got = E.fft_exec(x, [0], [0], [nsig], nsig, nfft, tap, sc, detrend=0, output="fourier", keeptapers=True, **kw)[0]
        err = np.abs(got.astype(np.complex128) - ref)
        chmax = np.abs(ref).max(axis=(0, 1), keepdims=True)                 # largest bin of each channel
        rel_ch = err / chmax
        pure = err > 1e-5 * np.abs(ref)
        crit = err > 1e-5 * np.abs(ref) + 1e-6 * np.abs(ref).max()
        q = np.percentile(rel_ch, [50, 90, 99, 100])
        print(f"  {label:30s} err / max|channel|: median {q[0]:.1e}  p90 {q[1]:.1e}  p99 {q[2]:.1e}  max {q[3]:.1e} | "
              f"bins outside pure rtol 1e-5: {100 * pure.mean():.2f} %  | outside the criterion: {100 * crit.mean():.3f} %")
        # the bins that miss pure rtol are the small ones: their size relative to the channel's largest bin
        if pure.any():
            small = (np.abs(ref) / chmax)[pure]
            print(f"  {'':30s} bins missing pure rtol have |b| / max|channel| <= {small.max():.1e} (median {np.median(small):.1e})")


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "c1.npz"))
    report("c1 trial 0 (AR(2), BASELINE config 1)", z["trial0"], 1000.0, 2000, "dpss", {"NW": 4.0, "Kmax": 7})
    z5 = np.load(os.path.join(ROOT, "tests", "golden", "conn5.npz"))
    trials = O.ar2_network(z5["adj"], 1000, 1, seed=7)
    report("conn5 trial 0 (coupled AR(2), 200 Hz)", trials[0], 200.0, 1000, "dpss", {"NW": 15.0, "Kmax": 5})
    t = np.arange(4096) / 1000.0
    rng = np.random.default_rng(0)
    harm = np.stack([np.cos(2 * np.pi * f * t) for f in (40.0, 100.0, 7.3, 333.0)], axis=1)
    harm = (harm + 1e-3 * rng.normal(size=harm.shape)).astype(np.float32)
    report("harmonics + 60 dB noise floor (known-answer style, test_timefreq.py:351-404)", harm, 1000.0, 4096, "hann", {})
    off = rng.normal(size=(4096, 4)).astype(np.float32)
    off += np.array([1000.0, -313.7, 25.0, 0.0], dtype=np.float32)
    off[:, 2] += (5.0 * np.sin(2 * np.pi * 50.0 * t)).astype(np.float32)
    report("channels riding on offsets 1000 / -313.7 / 25 (+ 50 Hz line) / 0", off, 1000.0, 4096, "dpss", {"NW": 4.096, "Kmax": 7})
    print("""
Reading: with the reference-order mean the fp32 kernels sit at ~1e-7 of a channel's largest bin everywhere (fp32 has
6e-8 of relative precision; the radix-16 passes add log-many roundings).  A pure rtol of 1e-5 is missed only by bins
that are >= 40 dB below their channel's peak - there an absolute error of 1e-7 max|b| IS more than 1e-5 |b| - which no
fp32 transform can avoid and which the reference's own complex64 storage does not resolve either (its spectra are
rounded to 6e-8 RELATIVE to each bin, but its float32 detrending leaves ~1e-6 x offset of absolute error in the bins
next to DC).  The floor term 1e-6 max|b| of the criterion covers exactly that; with the float64 block sums the
offset channels break the criterion next to DC (the reference's own mean is the less exact one).  A compensated first
stage (float64 taper multiply / first radix pass) would not move any of these numbers: the taper product contributes
3e-8 |x w| of white rounding noise, i.e. 1e-9 of the peak after the transform.""")


if __name__ == "__main__":
    main()
