"""The HIP kernel sources, compiled unchanged for the host and run workgroup-by-workgroup on OS
threads (tests/emu/hip_emu.h), against the oracle.  This checks index arithmetic, LDS layouts,
the packed-real-FFT separation and the MFMA lane maps without a GPU; the `-m gpu` tests are the
parity tests proper."""
import numpy as np
import pytest

import emu_driver as E
from oracle import spy_oracle as O
from parity import assert_parity, excess


def _fft_case(nsig, nfft, nchan, K, output, keeptapers, detrend, demean_taper=False, G=None, generic=False,
              freq_idx=None, chan_idx=None, nseg=2, seed=1, long=False, no_mixed=False, nostage=False, dec=None):
    rng = np.random.default_rng(seed)
    data = rng.normal(size=(nsig * nseg + 9, nchan)).astype("f4")
    ss = np.array([4 + i * nsig for i in range(nseg)])
    taper, topt = ("dpss", {"NW": (K + 1) / 2, "Kmax": K}) if K > 1 else ("hann", {})
    tapers = O.taper_table(taper, nsig, nfft, topt)
    out = E.fft_exec(data, ss, ss, ss + nsig, nsig, nfft, tapers, O.spec_scale(nsig, nfft), detrend, demean_taper,
                     freq_idx, output, keeptapers, chan_idx=chan_idx, G=G, force_generic=generic, force_long=long,
                     no_mixed=no_mixed, mixed_nostage=nostage, dec=dec)
    freqs = np.fft.rfftfreq(nfft, 1e-3)
    foi = freqs if freq_idx is None else freqs[freq_idx]
    for b in range(nseg):
        x = data[ss[b]:ss[b] + nsig]
        if chan_idx is not None:
            x = x[:, chan_idx]
        ref, _ = O.mtmfft_cF(np.array(x), foi=foi, keeptapers=keeptapers, polyremoval=None if detrend < 0 else detrend,
                             output=output, method_kwargs=dict(samplerate=1000.0, taper=taper, taper_opt=topt,
                                                               nSamples=nfft, demean_taper=demean_taper))
        assert_parity(out[b], ref[0], what=f"segment {b}")


@pytest.mark.parametrize("log2n,G", [(8, 16), (9, 8), (10, 4), (11, 2), (12, 1), (13, 1), (14, 1)])
def test_pow2_kernel_every_length(log2n, G):
    n = 1 << log2n
    nchan = 4 * G + 1 if n <= 2048 else (5 if n == 4096 else 3)   # ragged channel count: padded quads / pairs
    _fft_case(n, n, nchan, 2, "pow", False, 0, G=G, nseg=1)


def test_pow2_kernel_full_quads_fast_epilogue():
    # channel counts that are multiples of 4 with every bin kept take the straight-line store path
    _fft_case(1024, 1024, 8, 2, "pow", True, 0)
    _fft_case(512, 512, 4, 2, "fourier", True, 0)
    _fft_case(256, 256, 8, 3, "pow", False, 1)
    _fft_case(512, 512, 8, 2, "abs", True, -1)
    _fft_case(300, 512, 6, 2, "fourier", True, 0)       # even, not a multiple of 4


@pytest.mark.parametrize("C,N,B,K", [(5, 256, 2, 1), (8, 512, 2, 2), (37, 256, 2, 1)])
def test_blocked_handover_layout(C, N, B, K):
    """Channel-blocked FFT -> CSD hand-over: a pure re-ordering of the spectra, identical accumulator."""
    rng = np.random.default_rng(C)
    data = rng.normal(size=(B * N, C)).astype("f4")
    ss = np.arange(B) * N
    tapers = O.taper_table("dpss", N, N, {"NW": 2.0, "Kmax": K}) if K > 1 else O.taper_table("hann", N, N)
    F = N // 2 + 1
    std = E.fft_exec(data, ss, ss, ss + N, N, N, tapers, np.sqrt(2) / N, 0, False, None, "fourier", True)
    blk = E.fft_exec(data, ss, ss, ss + N, N, N, tapers, np.sqrt(2) / N, 0, False, None, "fourier", True, blocked=True)
    nq = (C + 3) // 4
    got = blk.transpose(0, 2, 1, 3).reshape(B * K, F, 4 * nq)      # back to (rows, F, padded channels)
    np.testing.assert_array_equal(got[:, :, :C], std.reshape(B * K, F, C))
    assert np.abs(got[:, :, C:]).max(initial=0.0) < 1e-5            # padding channels: rounding residue only
    a_std = np.zeros((F, C, C), np.complex64)
    a_blk = np.zeros((F, C, C), np.complex64)
    # the blocked hand-over is served by the 4-multiplication kernels only: compare like with like (the standard
    # layout would otherwise take the 3-multiplication kernel, whose rounding differs in the last bit)
    E.csd_accumulate(std.reshape(B * K, F, C), a_std, force_4m=True)
    E.csd_accumulate(np.ascontiguousarray(blk), a_blk, blocked=True)
    np.testing.assert_array_equal(a_std, a_blk)


def test_pow2_kernel_modes():
    _fft_case(1000, 1024, 5, 3, "fourier", True, 1, demean_taper=True)
    _fft_case(1024, 1024, 5, 3, "fourier", False, 0)
    _fft_case(700, 1024, 4, 1, "abs", True, -1)
    _fft_case(512, 512, 6, 2, "real", False, 0, freq_idx=np.array([5, 0, 256, 100]), chan_idx=[5, 5, 0, 3, 1])


@pytest.mark.parametrize("nsig,nfft", [(2000, 2000), (500, 1000), (360, 360), (77, 154), (101, 202), (64, 64)])
def test_generic_kernel_lengths(nsig, nfft):
    # lengths that are not powers of two: Bluestein on the packed engine (M = 256 ... 4096) ...
    _fft_case(nsig, nfft, 3, 2, "pow", False, 0, no_mixed=True)
    _fft_case(nsig, nfft, 2, 1, "fourier", True, 1, demean_taper=True, nseg=1, no_mixed=True)
    # ... and the mixed-radix / Bluestein LDS kernel that still serves nfft > 4096
    _fft_case(nsig, nfft, 3, 2, "pow", False, 0, generic=True, nseg=1)


def test_long_transform_path():
    """Bluestein with four-step transforms through HBM (the path of nfft > 4096 non-powers-of-two and of
    nfft > 16384), forced onto small lengths: M = 4096 = 64 x 64 and M = 8192 = 128 x 64."""
    _fft_case(1500, 1500, 5, 2, "pow", False, 1, demean_taper=True, long=True, nseg=1)
    _fft_case(700, 900, 4, 1, "fourier", True, 0, long=True, nseg=2)
    _fft_case(2500, 2500, 3, 1, "abs", True, -1, long=True, nseg=1, freq_idx=np.array([7, 0, 1250, 333]))


def test_bluestein_kernel_modes():
    _fft_case(500, 500, 8, 1, "pow", True, 0, no_mixed=True)       # full quads: float4 stores
    _fft_case(500, 500, 5, 3, "abs", False, 1, no_mixed=True)
    _fft_case(300, 360, 6, 2, "fourier", False, -1, no_mixed=True)
    _fft_case(77, 77, 4, 2, "real", True, 0, freq_idx=np.array([3, 0, 38, 20]), chan_idx=[3, 3, 0, 1])
    for n in (3, 5, 7, 13):                                        # trials of a handful of samples
        _fft_case(n, n, 3, 1, "fourier", True, 0, nseg=2)
    _fft_case(3, 8, 2, 1, "pow", False, -1, nseg=1)


@pytest.mark.parametrize("nsig,nfft,sched", [
    (2000, 2000, (200, 10, 2)),      # BASELINE config 1: 10 x 10 x 10 x 2
    (500, 1000, (100, 10, 10)),      # padded: 10 x 10 x 10
    (360, 360, (45, 5, 8)),          # 5 x 3 x 3 x 8
    (3000, 3000, (334, 3, 10)),      # 3 x 10 x 10 x 10: ragged butterfly counts per thread
    (5000, 5000, (500, 5, 10)),      # 5 x 10 x 10 x 10, 512 threads, 160 KB of LDS
    (64, 64, (8, 8, 8)), (16, 16, (2, 8, 2)), (25, 30, (4, 3, 10)), (1215, 1215, (135, 5, 3)),
])
def test_mixed_radix_kernel_lengths(nsig, nfft, sched):
    """K1m, the packed engine for 5-smooth lengths, with the schedule mix_schedule (the code plan_create runs) picks."""
    _fft_case(nsig, nfft, 5, 2, "pow", False, 0, nseg=2)
    assert E.LAST_MIXED["nfft"] == nfft
    assert (E.LAST_MIXED["th"], E.LAST_MIXED["radix0"], E.LAST_MIXED["radix_last"]) == sched
    _fft_case(nsig, nfft, 2, 1, "fourier", True, 1, demean_taper=True, nseg=1)


def test_mixed_radix_kernel_modes():
    _fft_case(500, 500, 8, 1, "pow", True, 0)                      # full quads: float4 stores
    _fft_case(500, 500, 5, 3, "abs", False, 1)
    _fft_case(300, 360, 6, 2, "fourier", False, -1)
    _fft_case(300, 360, 9, 2, "fourier", True, 0, demean_taper=True, nostage=True)     # the re-read path on a short length
    _fft_case(250, 250, 17, 2, "pow", False, 1, nostage=True, nseg=3)                 # several quad groups, ragged last quad
    _fft_case(80, 80, 4, 2, "real", True, 0, freq_idx=np.array([3, 0, 40, 20]), chan_idx=[3, 3, 0, 1])
    _fft_case(150, 150, 4, 3, "angle", False, -1)
    _fft_case(36, 36, 4, 1, "fourier", True, 0)                     # 3 x 3 x 4
    assert E.LAST_MIXED["nfft"] == 36 and E.LAST_MIXED["radix_last"] == 4


def test_generic_kernel_matches_pow2_kernel():
    _fft_case(1024, 1024, 4, 2, "pow", True, 0, generic=True)


def test_zero_extended_segments():
    rng = np.random.default_rng(2)
    nsig = nfft = 256
    data = rng.normal(size=(1000, 4)).astype("f4")
    starts = np.array([-128, 0, 872, 300])
    lo = np.array([0, 0, 100, 350])
    hi = np.array([1000, 1000, 1000, 420])
    tapers = O.taper_table("hann", nsig, nfft)
    out = E.fft_exec(data, starts, lo, hi, nsig, nfft, tapers, np.sqrt(2) / nsig, 0, False, None, "fourier", True)
    for b in range(4):
        seg = np.zeros((nsig, 4), "f4")
        for n in range(nsig):
            r = starts[b] + n
            if lo[b] <= r < hi[b]:
                seg[n] = data[r]
        seg -= seg.mean(axis=0)
        ref = (np.fft.rfft(tapers[0][:, None] * seg, axis=0) * (np.sqrt(2) / nsig)).astype(np.complex64)
        assert_parity(out[b, 0], ref, what=f"segment {b}")


def test_fp32_fft_error_level():
    """Documents the accuracy the float32 kernel delivers: ~1e-7 of the rms bin amplitude."""
    rng = np.random.default_rng(1)
    n = 4096
    data = rng.normal(size=(n, 4)).astype("f4")
    z = np.array([0])
    out = E.fft_exec(data, z, z, z + n, n, n, np.ones((1, n)), 1.0, -1, False, None, "fourier", True)[0, 0]
    ref = np.fft.rfft(data.astype("f8"), axis=0)
    rms = np.sqrt((np.abs(ref) ** 2).mean())
    assert np.abs(out - ref).max() / rms < 1e-6
    assert np.sqrt((np.abs(out - ref) ** 2).mean()) / rms < 3e-7


@pytest.mark.parametrize("C,F,R,tpw", [(5, 9, 6, 0), (40, 5, 7, 0), (70, 3, 10, 5), (130, 2, 4, 0), (33, 4, 5, 1),
                                       (100, 3, 5, 0), (256, 2, 4, 0), (256, 1, 37, 0), (240, 1, 6, 0),
                                       # the instruction-lean path away from 256 channels: 2 / 4 / 8 frequencies per
                                       # LDS row (incl. a last, partial group), tiles crossing into the next frequency
                                       (128, 5, 6, 0), (64, 7, 5, 0), (32, 11, 4, 0), (6, 20, 3, 0),
                                       (100, 3, 5, 3), (40, 5, 7, 1),
                                       # the wide variant (512-element rows, several workgroups per frequency)
                                       (320, 2, 5, 0), (384, 1, 9, 0), (257, 1, 3, 0),
                                       # odd channel counts on the lean path: 8-byte staging loads, odd row lengths
                                       (5, 9, 6, 5), (33, 4, 5, 3), (63, 5, 6, 0), (255, 2, 4, 0)])
def test_csd_mfma_kernel(C, F, R, tpw):
    rng = np.random.default_rng(C)
    spec = (rng.normal(size=(R, F, C)) + 1j * rng.normal(size=(R, F, C))).astype(np.complex64)
    acc = np.zeros((F, C, C), np.complex64)
    E.csd_accumulate(spec[:R // 2], acc, tpw)
    E.csd_accumulate(spec[R // 2:], acc, tpw)
    fused = {o: E.coh_from_accumulator(acc, 1.0 / R, o) for o in ("abs", "complex")}
    E.csd_finalize(acc, 1.0 / R)
    for o, got in fused.items():                       # fused K5 = finalize + normalize, bit for bit
        np.testing.assert_array_equal(got, E.coh_normalize(acc, o))
    ref = (np.einsum("rfi,rfj->fij", spec.astype(np.complex128), spec.conj().astype(np.complex128)) / R)
    assert_parity(acc, ref.astype(np.complex64), what="csd")
    assert np.array_equal(acc, acc.conj().transpose(0, 2, 1)) and np.all(acc.imag[:, range(C), range(C)] == 0)
    for output in ("abs", "pow", "complex", "imag"):
        assert_parity(E.coh_normalize(acc, output), O.normalize_csd(acc, output), what=output)


@pytest.mark.parametrize("C,F,R", [(256, 2, 7), (128, 5, 9), (64, 6, 9), (192, 2, 6), (320, 1, 6),
                                   # rows narrower than the kernel's LDS image: odd counts (last row through the
                                   # 4-multiplication kernels, 8-byte aligned copies), even non-multiples of 16
                                   (37, 9, 7), (63, 5, 9), (90, 3, 6), (255, 2, 7), (301, 1, 6), (14, 20, 5)])
def test_csd_3m_kernel(C, F, R):
    """csd3m_kernel (3-multiplication complex product, 16 x 16 sub-tiles, one workgroup per frequency, rows global ->
    LDS by DMA): all 136 sub-tiles land where they belong, ragged last chunks are zero-filled, the accumulation
    continues across launches, and the result agrees with the 4-multiplication kernel to rounding.  The spectra
    carry a 40 dB spread over channels and a strong common component (coherent channels, small imaginary parts) -
    the dynamic range the 3M form is sensitive to.  128 channels: two frequencies per workgroup (72 sub-tiles), the
    odd last frequency alone in its packed row."""
    code = {256: 8}.get(C, 9)
    rng = np.random.default_rng(R)
    gain = 10 ** rng.uniform(-1, 1, size=C)
    common = rng.normal(size=(R, F, 1)) + 1j * rng.normal(size=(R, F, 1))
    spec = ((rng.normal(size=(R, F, C)) + 1j * rng.normal(size=(R, F, C)) + 2.0 * common) * gain).astype(np.complex64)
    acc = np.zeros((F, C, C), np.complex64)
    assert E.csd_accumulate(spec[:R // 2], acc) == code and E.csd_accumulate(spec[R // 2:], acc) == code
    acc4 = np.zeros((F, C, C), np.complex64)
    assert E.csd_accumulate(spec, acc4, force_4m=True) != code
    ref = np.einsum("rfi,rfj->fij", spec.astype(np.complex128), spec.conj().astype(np.complex128))
    ii, jj = np.tril_indices(C)
    # element-wise against the exact product: the error scale of an entry is sqrt(S_ii S_jj), not the global maximum
    scale = np.sqrt(np.abs(ref[:, ii, ii]) * np.abs(ref[:, jj, jj]))
    for got in (acc, acc4):
        err = np.abs(got[:, ii, jj] - ref[:, ii, jj]) / scale
        assert err.max() < 3e-6, err.max()
    assert_parity(acc[:, ii, jj], ref[:, ii, jj].astype(np.complex64), what="3M csd, lower triangle")
    E.csd_finalize(acc, 1.0 / R)
    assert np.array_equal(acc, acc.conj().transpose(0, 2, 1)) and np.all(acc.imag[:, range(C), range(C)] == 0)
    assert_parity(acc, (ref / R).astype(np.complex64), what="3M csd finalised")


@pytest.mark.parametrize("C,F,T,K", [(5, 4, 6, 3), (40, 2, 7, 1), (70, 1, 5, 7)])
def test_ppc_kernel(C, F, T, K):
    """K7's kernel source on the CPU: phasor sums + closed form = the oracle's walk over all trial pairs."""
    rng = np.random.default_rng(C)
    spec = (rng.normal(size=(T, K, F, C)) + 1j * rng.normal(size=(T, K, F, C))).astype(np.complex64)
    spec += (2.0 * rng.normal(size=(1, K, F, C))).astype(np.complex64)
    spec[..., C - 1] = 0                                # np.angle(0) = 0: consistent with everything
    st = O.spectral_dyadic_product(spec)
    ref = O.ppc(st)[0]
    U = np.zeros((F, C, C), np.complex64)
    E.ppc_accumulate(spec[:2].reshape(-1, F, C), K, U)
    E.ppc_accumulate(spec[2:].reshape(-1, F, C), K, U)
    got = E.ppc_finalize(U, T, True)
    assert_parity(got, ref, what="ppc", rtol=1e-4, atol_rel=2e-5)
    assert np.array_equal(got, got.transpose(0, 2, 1)) and np.allclose(got[:, C - 1], 1, atol=1e-6)
    U2 = np.zeros((F, C, C), np.complex64)
    E.ppc_accumulate_csd(st, U2)
    assert_parity(E.ppc_finalize(U2, T, False), ref, what="ppc from csd", rtol=1e-4, atol_rel=2e-5)


@pytest.mark.parametrize("C,F,T,K,output", [(5, 3, 6, 3, "abs"), (40, 2, 5, 2, "complex"), (33, 1, 4, 1, "imag")])
def test_jackknife_coherence_kernel(C, F, T, K, output):
    """K9's kernel source on the CPU: sums of (leave-one-out coherence - direct) and of its squared modulus against a
    NumPy walk over the replicates (statistics/jackknifing.py:14-108 + csd.py:118-172)."""
    rng = np.random.default_rng(C)
    spec = (rng.normal(size=(T, K, F, C)) + 1j * rng.normal(size=(T, K, F, C))).astype(np.complex64)
    spec += (1.5 * rng.normal(size=(1, K, F, C))).astype(np.complex64)
    st = O.spectral_dyadic_product(spec).astype(np.complex128)                  # (T, F, C, C)
    S = st.mean(axis=0)
    direct = O.normalize_csd(S.astype(np.complex64)[None], output)[0]
    sum_d = np.zeros((F, C, C), np.complex128 if output == "complex" else np.float64)
    sum_d2 = np.zeros((F, C, C), np.float64)
    E.jack_coh_accumulate(spec[:2].reshape(-1, F, C), K, S, direct, output, T, sum_d, sum_d2)
    E.jack_coh_accumulate(spec[2:].reshape(-1, F, C), K, S, direct, output, T, sum_d, sum_d2)
    rd, rd2 = np.zeros_like(sum_d), np.zeros_like(sum_d2)
    for t in range(T):
        loo = ((T * S - st[t]) / (T - 1)).astype(np.complex64)
        d = O.normalize_csd(loo[None], output)[0].astype(sum_d.dtype) - direct
        rd += d
        rd2 += np.abs(d) ** 2
    np.testing.assert_allclose(sum_d, rd, rtol=1e-3, atol=2e-6 * T)
    np.testing.assert_allclose(sum_d2, rd2, rtol=2e-3, atol=1e-9)


@pytest.mark.parametrize("C,N,norm", [(5, 600, 0), (6, 301, 1), (3, 1400, 2)])
def test_ccov_kernel(C, N, norm):
    """K8's kernel source on the CPU: lags from accumulated cross spectra of zero-padded trials = the oracle's
    per-pair fftconvolve walk (even N: the upper triangle one lag late, as in the reference)."""
    rng = np.random.default_rng(N)
    T = 1 if norm == 2 else 3
    x = rng.normal(size=(T, N, C))
    x[:, 2:, 1] += 0.7 * x[:, :-2, 0]
    x = (x - x.mean(axis=1, keepdims=True)).astype(np.float32)
    nlag = N // 2 + (N & 1)
    L = 1024
    while L < N + nlag:
        L *= 2
    X = np.fft.rfft(x.astype(np.float64), n=L, axis=1).astype(np.complex64)          # (T, F, C)
    acc = np.einsum("tfa,tfb->fab", X, X.conj()).astype(np.complex64)
    got = E.ccov_from_accumulator(acc, N, 1.0 / T, norm)
    ref = np.mean([O.cross_covariance(t, 1.0, None, norm == 2)[0] for t in x], axis=0)
    if norm == 1:
        ref = O.normalize_ccov(ref)
    assert_parity(got, ref[:, 0].astype(np.float32), what=f"ccov norm={norm}", atol_rel=1e-5)


@pytest.mark.parametrize("nsig,scales,detrend,output", [
    (700, [0.05, 0.02, 0.004], 0, "pow"),            # kernels of 500/200/40 taps, one block
    (3000, [0.03, 0.006], 1, "fourier"),             # several overlap-save blocks
    (300, [0.2, 0.01], -1, "abs"),                   # kernel (2000 taps) much longer than the signal: trimmed
])
def test_cwt_kernel(nsig, scales, detrend, output):
    rng = np.random.default_rng(nsig)
    nchan = 3
    data = rng.normal(size=(nsig + 40, nchan)).astype("f4") + 0.5
    pre0 = 7                                          # pre-selection starts inside the trial
    ss, lo, hi = np.array([pre0 + 5]), np.array([5]), np.array([5 + nsig + 20])
    scales = np.asarray(scales)
    out = E.cwt_exec(data, ss, lo, hi, nsig, scales, 1e-3, 6.0, detrend, output)[0]
    trial = O.detrend(np.array(data[5:5 + nsig + 20]), None if detrend < 0 else detrend)
    ref = O.convert_output(O.cwt(trial[pre0:pre0 + nsig], 1000.0, scales).transpose(1, 0, 2), output)
    assert_parity(out, ref, what="cwt")


@pytest.mark.parametrize("output", ["pow", "fourier"])
@pytest.mark.parametrize("nbmin,nchan", [(1024, 3), (2048, 4)])
def test_cwt_direct_kernels(output, nbmin, nchan):
    """cwt2d_kernel (round 6): the transform kernel writes the (time, scale, channel) layout itself - 1024- and 2048-point
    blocks, an odd channel count (padded pair, scalar stores), plain and accumulating (out[b] += segment b) calls, a
    post-selection of samples, the channel-major input copy in front of it."""
    rng = np.random.default_rng(nbmin + nchan)
    nsig, T = 1400, 2
    data = rng.normal(size=(T * nsig + 30, nchan)).astype("f4") + 1.5
    ss = np.array([10, nsig + 20])
    lo, hi = ss.copy(), ss + nsig
    scales = np.array([0.012, 0.004])
    ref = np.stack([O.convert_output(O.cwt(O.detrend(np.array(data[a:a + nsig]), 0), 1000.0, scales).transpose(1, 0, 2), output)
                    for a in ss])
    for mode in (2, 2 | 4):
        out = E.cwt_exec(data, ss, lo, hi, nsig, scales, 1e-3, 6.0, 0, output, mode=mode, nbmin=nbmin)
        assert_parity(out, ref, what=f"direct cwt, mode {mode}")
    keep = np.r_[0:3, 5:1300:9, 1023, 1024, 1399]
    tpos = np.full(nsig, -1, dtype=np.int32)
    tpos[keep] = np.arange(keep.size)
    sel = E.cwt_exec(data, ss, lo, hi, nsig, scales, 1e-3, 6.0, 0, output, tpos=tpos, ntime_out=keep.size, mode=2 | 4, nbmin=nbmin)
    assert_parity(sel, ref[:, keep], what="direct cwt, selected samples")


@pytest.mark.parametrize("output", ["pow", "fourier"])
@pytest.mark.parametrize("T", [4, 5])
def test_cwt_trial_sums_on_pairs_of_trials(output, T):
    """cwt2_kernel<..., PAIRT> (round 6): one channel of TWO consecutive trials in the packed halves, their sum staged;
    an odd trial count leaves the last pair half empty; with and without the channel-major input copy; 2048-point
    blocks as the trial-sum policy of cwt.hip picks longer ones."""
    rng = np.random.default_rng(T)
    nsig, nchan = 900, 3
    data = rng.normal(size=(T * nsig, nchan)).astype("f4") - 0.7
    ss = np.arange(T) * nsig
    scales = np.array([0.015, 0.005])
    ref = sum(O.convert_output(O.cwt(O.detrend(np.array(data[a:a + nsig]), 0), 1000.0, scales).transpose(1, 0, 2).astype(np.complex128), output)
              for a in ss)[None]
    for mode, nbmin in ((1, 1024), (1 | 4, 1024), (1 | 4, 2048)):
        out = E.cwt_exec(data, ss, ss, ss + nsig, nsig, scales, 1e-3, 6.0, 0, output, accumulate=2, mode=mode, nbmin=nbmin)
        assert_parity(out, ref.astype(out.dtype), what=f"pair sums, mode {mode}, blocks >= {nbmin}")


@pytest.mark.parametrize("n", [32, 37, 70])
def test_blocked_inverse(n):
    """Block Gauss-Jordan inverse (16 x 16 diagonal blocks): ragged sizes, identity padding, tiny-pivot flag."""
    rng = np.random.default_rng(n)
    B = 2
    A = rng.normal(size=(B, n, n)) + 1j * rng.normal(size=(B, n, n)) + 3 * np.sqrt(n) * np.eye(n)
    inv, info = E.w_inv(A, blocked=True)
    assert not info.any()
    np.testing.assert_allclose(inv @ A, np.tile(np.eye(n), (B, 1, 1)), atol=1e-10)
    P = np.zeros((1, n, n), complex)
    P[0] = np.eye(n)[::-1]                     # anti-diagonal permutation: every leading block is singular
    _, info = E.w_inv(P, blocked=True)
    assert info[0] == 2


@pytest.mark.parametrize("n", [64, 37, 70])
def test_blocked_inverse_on_f64_matrix_cores(n):
    """zinv_mfma_kernel (32 x 32 blocks, R = D A_k* and the trailing update as v_mfma_f64_16x16x4_f64 tiles): ragged
    sizes (identity padding, partial tiles), asymmetric complex matrices (a row/column swap of a fragment layout would
    show), the tiny-pivot flag."""
    rng = np.random.default_rng(n)
    B = 2
    A = rng.normal(size=(B, n, n)) + 1j * rng.normal(size=(B, n, n)) + 3 * np.sqrt(n) * np.eye(n)
    inv, info = E.w_inv(A, blocked="mfma")
    assert not info.any()
    np.testing.assert_allclose(inv, np.linalg.inv(A), rtol=1e-9, atol=1e-11)
    P = np.zeros((1, n, n), complex)
    P[0] = np.eye(n)[::-1]
    _, info = E.w_inv(P, blocked="mfma")
    assert info[0] == 2


@pytest.mark.parametrize("n", [128, 100, 192, 64])
def test_blocked_inverse_with_64_row_blocks(n):
    """zinv64_mfma_kernel (64 x 64 diagonal blocks, the matrix walked in column quarters through a 64 x 64 R panel in
    LDS, the quarter of the block itself last): ragged sizes, asymmetric complex matrices, out of place, the tiny-pivot
    flag."""
    rng = np.random.default_rng(n)
    B = 1 if n > 128 else 2            # (one OS thread per GPU thread: 192 x 192 costs 20 s per matrix)
    A = rng.normal(size=(B, n, n)) + 1j * rng.normal(size=(B, n, n)) + 3 * np.sqrt(n) * np.eye(n)
    inv, info = E.w_inv(A, blocked="mfma64")
    assert not info.any()
    np.testing.assert_allclose(inv, np.linalg.inv(A), rtol=1e-9, atol=1e-11)
    P = np.zeros((1, n, n), complex)
    P[0] = np.eye(n)[::-1]
    _, info = E.w_inv(P, blocked="mfma64")
    assert info[0] == 2


@pytest.mark.parametrize("F", [9, 17, 11, 8])       # L = 16 (4*4), 32 (4*4*2), 20 (4*5), 14 (2*7)
def test_plus_operator(F):
    rng = np.random.default_rng(F)
    n, L = 3, 2 * (F - 1)
    g = rng.normal(size=(F, n, n)) + 1j * rng.normal(size=(F, n, n))
    full = np.zeros((L, n, n), complex)
    full[:F] = g
    full[F:] = np.conj(g[1:F - 1][::-1])
    ref, ref0 = O.plus_operator(full)
    gp, g0 = E.w_plus(g)
    np.testing.assert_allclose(gp, ref[:F], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g0, ref0, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("F,n", [(129, 3), (257, 2), (513, 5), (1025, 2), (2049, 2)])
def test_plus_operator_two_entries_per_transform(F, n):
    """plus4_kernel: two real lag sequences per complex radix-16 transform, four adjacent entries per workgroup
    (n = 3, 5: ragged last workgroup / odd pair), complex DC and Nyquist bins (their imaginary parts must be
    ignored as real(ifft(.)) ignores them) - against the oracle's fft/ifft formulation and the radix-4 kernel."""
    rng = np.random.default_rng(F + n)
    L = 2 * (F - 1)
    g = rng.normal(size=(F, n, n)) + 1j * rng.normal(size=(F, n, n))
    full = np.zeros((L, n, n), complex)
    full[:F] = g
    full[F:] = np.conj(g[1:F - 1][::-1])
    ref, ref0 = O.plus_operator(full)
    gp, g0 = E.w_plus(g, fast=True)
    np.testing.assert_allclose(gp, ref[:F], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g0, ref0, rtol=1e-12, atol=1e-12)
    old, old0 = E.w_plus(g)
    np.testing.assert_allclose(gp, old, rtol=1e-12, atol=1e-12)


def test_zgemm_on_f64_matrix_cores():
    """n >= 48 takes the v_mfma_f64_16x16x4_f64 tiles (asymmetric operands catch row/column swaps; n = 50 leaves
    ragged tiles)."""
    rng = np.random.default_rng(50)
    n = 50
    A = rng.normal(size=(1, n, n)) + 1j * rng.normal(size=(1, n, n))
    Bm = rng.normal(size=(1, n, n)) + 1j * rng.normal(size=(1, n, n))
    np.testing.assert_allclose(E.w_gemm(A, Bm), A @ Bm, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(E.w_gemm(A, Bm, opB=1, addI=1), A @ Bm.conj().transpose(0, 2, 1) + np.eye(n), rtol=1e-12,
                               atol=1e-12)
    # addI & 2: B declared lower triangular (the Cholesky factor): the zero rows above a column tile are skipped
    n = 130
    A = rng.normal(size=(1, n, n)) + 1j * rng.normal(size=(1, n, n))
    L = np.tril(rng.normal(size=(1, n, n)) + 1j * rng.normal(size=(1, n, n)))
    np.testing.assert_allclose(E.w_gemm(A, L, addI=2), A @ L, rtol=1e-12, atol=1e-12)


def test_zgemm_fused_skew_and_error_check():
    """The two fused forms of the matrix-core gemm used inside the Wilson iteration: psi (g+ + S) with
    S = triu(g0) - triu(g0)^H joined to the B operand, and max_rel_err(CSD, psi psi^H) without storing the product
    (wilson_sf.py:97-101,190-194); n = 70 leaves partial 64 x 64 tiles."""
    rng = np.random.default_rng(12)
    n, B = 70, 2
    psi = rng.normal(size=(B, n, n)) + 1j * rng.normal(size=(B, n, n))
    gp = rng.normal(size=(B, n, n)) + 1j * rng.normal(size=(B, n, n))
    g0 = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    S, g0S = E.w_skew(g0)
    Sref = np.triu(g0) - np.triu(g0).conj().T
    np.testing.assert_allclose(S, Sref, rtol=0, atol=0)
    np.testing.assert_allclose(g0S, g0 + Sref, rtol=1e-15)
    np.testing.assert_allclose(E.w_gemm_fused(psi, gp, badd=S), psi @ (gp + Sref), rtol=1e-12, atol=1e-12)
    # X X^H + I through the Hermitian instance (lower tiles computed, mirrored above the diagonal)
    np.testing.assert_allclose(E.w_gemm(psi, psi, opB=1, addI=1), psi @ psi.conj().transpose(0, 2, 1) + np.eye(n),
                               rtol=1e-12, atol=1e-12)
    herm = rng.normal(size=(B, n, n)) * 1e-3
    ref = psi @ psi.conj().transpose(0, 2, 1) * (1 + herm + herm.transpose(0, 2, 1))        # Hermitian reference
    err = E.w_gemm_fused(psi, psi, opB=1, ref=ref)
    np.testing.assert_allclose(err, O.max_rel_err(ref, psi @ psi.conj().transpose(0, 2, 1)), rtol=1e-10)


def test_wilson_building_blocks():
    rng = np.random.default_rng(3)
    n, B = 37, 3
    A = rng.normal(size=(B, n, n)) + 1j * rng.normal(size=(B, n, n))
    Bm = rng.normal(size=(B, n, n)) + 1j * rng.normal(size=(B, n, n))
    np.testing.assert_allclose(E.w_gemm(A, Bm), A @ Bm, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(E.w_gemm(A, Bm, opB=1, addI=1), A @ Bm.conj().transpose(0, 2, 1) + np.eye(n), rtol=1e-12,
                               atol=1e-12)
    np.testing.assert_allclose(E.w_gemm(A, Bm[0]), A @ Bm[0], rtol=1e-12, atol=1e-12)        # broadcast B
    inv, info = E.w_inv(A)
    assert not info.any()
    np.testing.assert_allclose(inv @ A, np.tile(np.eye(n), (B, 1, 1)), atol=1e-9)
    P = A @ A.conj().transpose(0, 2, 1) + n * np.eye(n)
    Lc, info = E.w_chol(P)
    assert not info.any()
    np.testing.assert_allclose(Lc, np.linalg.cholesky(P), rtol=1e-10, atol=1e-10)
    assert abs(E.w_cond(P[:1, :9, :9].copy(), iters=40) / np.linalg.cond(P[0, :9, :9]) - 1) < 5e-2


@pytest.mark.parametrize("n", [64, 70])
def test_panel_cholesky(n):
    """zchol_panel_kernel (n >= 64: left-looking panels of 32 columns; 70: a ragged last panel of 6): the factor, the
    zeroed strict upper triangle, and the flag for a matrix that is not positive definite."""
    rng = np.random.default_rng(n)
    X = rng.normal(size=(3, n, n)) + 1j * rng.normal(size=(3, n, n))
    P = X @ X.conj().transpose(0, 2, 1) + 0.5 * n * np.eye(n)
    P[2, 40, 40] = -1.0                                        # third matrix: indefinite
    Lc, info = E.w_chol(P)
    assert list(info) == [0, 0, 1]
    np.testing.assert_allclose(Lc[:2], np.linalg.cholesky(P[:2]), rtol=1e-10, atol=1e-10)
    assert np.all(np.triu(Lc[:2], 1) == 0)


@pytest.mark.skipif(not __import__("os").environ.get("SPY_EMU_SLOW"),
                    reason="~5 min of thread emulation; set SPY_EMU_SLOW=1 (the GPU suite covers the same chain)")
def test_wilson_granger_small_vs_oracle():
    """Whole Wilson/Granger chain on emulated kernels for a small problem (the golden-vector comparison of
    the real C++ driver runs on the GPU: tests/test_gpu_golden.py)."""
    adj = np.zeros((3, 3))
    adj[0, 1] = 0.3
    trials = O.ar2_network(adj, 32, 40, seed=3)
    acc = np.zeros((17, 3, 3), np.complex64)
    for x in trials:
        cs, _ = O.csd(O.detrend(x, 0), 200.0, 32, "dpss", {"NW": 2, "Kmax": 3}, demean_taper=True)
        acc += cs
    acc /= len(trials)
    G, H, Sigma, info = E.granger(acc, cond_iters=25)
    Gr, meta = O.granger_cF(acc[None])
    assert bool(info[0]) == bool(meta["converged--bool"]) and info[2] == float(meta["reg. factor--float"])
    np.testing.assert_allclose(info[3], float(meta["initial cond. num--float"]), rtol=0.2)   # 25 power iterations only
    assert info[1] < 5e-6
    rec = H @ Sigma @ H.conj().transpose(0, 2, 1)
    assert O.max_rel_err(acc.astype(np.complex128), rec) < 1e-5
    np.testing.assert_allclose(G[2:-1], Gr[0, 2:-1], rtol=5e-3, atol=5e-4)
    np.testing.assert_allclose(G, Gr[0], atol=1e-2)


def test_reference_order_mean_is_numpy_bit_for_bit():
    """seq_mean_kernel = np.mean(float32 (time x channel), axis=0) bit for bit: NumPy reduces the slow axis with one
    float32 accumulator per channel in time order (what scipy.signal.detrend(type='constant') subtracts,
    specest/compRoutines.py:169-170).  Offsets make the rounding sequence matter; zero-extended segments, channel
    selections and channel counts around the 64-lane edge are covered."""
    rng = np.random.default_rng(7)
    x = (rng.normal(size=(5000, 70)) * rng.uniform(0.1, 30, size=70) + rng.uniform(-500, 500, size=70)).astype(np.float32)
    got = E.seq_mean(x, [0, 17, 1000], [0, 17, 1000], [4096, 17 + 3001, 1000 + 33], 4096)
    assert np.array_equal(got[0], np.mean(x[:4096], axis=0))
    assert got[0].dtype == np.float32
    # rows outside [lo, hi) count as zeros: sum over the valid rows, divided by nsig
    assert np.array_equal(got[1], np.add.reduce(x[17:17 + 3001], axis=0) / np.float32(4096))
    assert np.array_equal(got[2], np.add.reduce(x[1000:1033], axis=0) / np.float32(4096))
    ci = [69, 0, 5, 5, 64, 63]
    got = E.seq_mean(x, [3], [3], [3 + 2000], 2000, chan_idx=ci)
    assert np.array_equal(got[0], np.mean(np.ascontiguousarray(x[3:2003][:, ci]), axis=0))    # C order, as a trial is
    # the float64 block sums differ from this by up to ~1e-6 of the offset - that is the whole point
    exact = x[:4096].astype(np.float64).mean(axis=0)
    assert np.abs(np.mean(x[:4096], axis=0) - exact).max() > 1e-5
    # ONE channel: the trial is an (nSamples, 1) array, contiguous along the reduced axis, which NumPy sums pairwise
    # (pairwise_sum_FLOAT) - a different rounding sequence from the row-by-row one above
    for n in (5, 8, 100, 128, 129, 1000, 4096, 4100, 4999):
        for src in (np.ascontiguousarray(x[:n, 3:4]), x):                      # a one-channel recording / one selected column
            ci = None if src.shape[1] == 1 else [3]
            got = E.seq_mean(src, [0], [0], [n], n, chan_idx=ci)
            assert np.array_equal(got[0], np.mean(np.ascontiguousarray(x[:n, 3:4]), axis=0)), n
    seq = np.float32(0)
    for v in x[:4096, 3]:
        seq = np.float32(seq + v)
    assert seq / np.float32(4096) != np.mean(np.ascontiguousarray(x[:4096, 3:4]), axis=0)[0]    # (the orders do differ)


@pytest.mark.parametrize("nsig,nfft,kw", [(4096, 4096, {}), (1000, 1024, {}), (2000, 2000, {}),
                                          (600, 600, {"force_generic": True}), (5003, 5003, {"force_long": True}),
                                          (16384, 16384, {})])
def test_offset_channels_match_the_reference_next_to_dc(nsig, nfft, kw):
    """Every kernel family with `reference_mean`: a channel riding on an offset 1000x its fluctuations reproduces the
    oracle (float32 sequential mean, then float64 taper and FFT) within the unwidened criterion; with the float64
    block sums the bins next to DC miss it (which is why the golden tests used to carry a wider floor)."""
    rng = np.random.default_rng(nsig)
    C = 6
    x = rng.normal(size=(nsig, C)).astype(np.float32)
    x[:, 1] += 1000.0
    x[:, 4] -= 313.7
    topt = {"NW": 3, "Kmax": 4}
    tap = O.taper_table("dpss", nsig, nfft, topt)
    sc = O.spec_scale(nsig, nfft)
    ref, _ = O.mtmfft(O.detrend(x, 0), 1000.0, nfft, "dpss", topt)
    got = E.fft_exec(x, [0], [0], [nsig], nsig, nfft, tap, sc, detrend=0, output="fourier", keeptapers=True,
                     reference_mean=True, **kw)[0]
    for c in range(C):                       # per channel: the criterion relative to THAT channel's largest bin
        assert_parity(got[:, :, c], ref[:, :, c], what=f"channel {c} with reference-order mean")
    plain = E.fft_exec(x, [0], [0], [nsig], nsig, nfft, tap, sc, detrend=0, output="fourier", keeptapers=True, **kw)[0]
    if nsig >= 1000:                         # (short sums happen to round well)
        assert excess(plain[:, :, 1], ref[:, :, 1]) > 1.0


# ---------------------------------------------------------------------------------------------------------------
# K1d: compile-time radix schedules (mtmfft_dec_kernel.h)
@pytest.mark.parametrize("dec,nfft,nchan,K,output,keeptapers,detrend,demean", [
    (100, 100, 68, 2, "fourier", True, 0, False),      # 10 x 10, sixteen quads per workgroup (+ one padded group)
    (400, 400, 35, 2, "pow", False, 1, False),         # 10 x 10 x 2 x 2, eight quads per workgroup
    (3200, 3200, 4, 1, "fourier", True, 0, False),     # 20 x 20 x 4 x 2
    (1000, 1000, 8, 2, "fourier", True, 0, False),     # 10 x 10 x 10, two quads per workgroup, fast stores
    (1000, 1000, 5, 3, "pow", False, 1, True),         # ragged channels, taper mean, linear detrend, demean_taper
    (2000, 2000, 4, 2, "pow", True, 0, False),         # 10 x 10 x 10 x 2: BASELINE config 1's length
    (2000, 2000, 3, 2, "fourier", False, -1, False),
    (2001, 2000, 8, 2, "abs", True, 0, False),         # 20 x 10 x 10
    (5000, 5000, 4, 1, "fourier", True, 0, False),     # 10 x 10 x 10 x 5
    (512, 512, 16, 2, "pow", True, 0, False),          # V = 8: 8 x 8 x 8
    (4096, 4096, 4, 2, "fourier", True, 0, False),     # 8 x 8 x 8 x 8
])
def test_dec_kernel_vs_oracle(dec, nfft, nchan, K, output, keeptapers, detrend, demean):
    _fft_case(nfft, nfft, nchan, K, output, keeptapers, detrend, demean_taper=demean, nseg=2 if nfft <= 1000 else 1, dec=dec)


@pytest.mark.parametrize("nfft,nchan,K,output,keeptapers,detrend,demean", [
    (600, 16, 2, "fourier", True, 0, False),           # 3 x 200, four quads per workgroup, fast stores
    (1500, 5, 3, "pow", False, 1, True),               # 3 x 500: ragged channels, taper mean, linear detrend, demean_taper
    (3000, 4, 2, "fourier", True, 0, False),           # 3 x 1000
    (3000, 3, 2, "abs", False, -1, False),
    (6000, 4, 1, "pow", True, 0, False),               # 3 x 2000
    (300, 33, 2, "fourier", True, 0, False),           # 3 x 100, eight quads per workgroup
    (2400, 4, 2, "abs", True, 0, False),               # 3 x 800, 20 values per thread
    (768, 9, 2, "fourier", True, 0, False),            # 3 x 256, 16 values per thread
    (3072, 4, 2, "pow", False, 0, False),              # 3 x 1024
])
def test_dec_kernel_radix3_decimation(nfft, nchan, K, output, keeptapers, detrend, demean):
    # N = 3 M: three scheduled sub-transforms side by side and one radix-3 combine through LDS (CfgD::P)
    _fft_case(nfft, nfft, nchan, K, output, keeptapers, detrend, demean_taper=demean, nseg=2 if nfft <= 1500 else 1, dec=nfft)


@pytest.mark.parametrize("dec,nfft,nchan,K,output,keeptapers,detrend,demean", [
    (1001, 1000, 8, 2, "fourier", True, 0, False),     # the 1000-point schedule with split exchanges
    (1001, 1000, 5, 3, "pow", False, 1, True),
    (10000, 10000, 4, 1, "fourier", True, 0, False),   # 20 x 20 x 5 x 5, 500 threads
    (10000, 10000, 3, 2, "pow", False, 0, False),
])
def test_dec_kernel_split_exchanges(dec, nfft, nchan, K, output, keeptapers, detrend, demean):
    _fft_case(nfft, nfft, nchan, K, output, keeptapers, detrend, demean_taper=demean, nseg=2 if nfft <= 1000 else 1, dec=dec)


def test_dec_kernel_radix3_padding_and_selection():
    _fft_case(2700, 3000, 6, 2, "pow", True, 0, dec=3000, nseg=1, freq_idx=np.array([0, 1, 1499, 1500, 37, 1000, 1001]),
              chan_idx=np.array([5, 0, 2, 2]))
    _fft_case(1400, 1500, 4, 2, "fourier", True, 1, dec=1500, nseg=2)


@pytest.mark.parametrize("P,M,nsig,nchan,K,output,keeptapers,detrend,demean,refmean", [
    (6, 2000, 12000, 5, 2, "fourier", True, 0, False, True),       # ragged quads, reference-order mean
    (6, 2000, 11000, 4, 2, "pow", False, 1, True, False),          # zero padding, line fit and demean_taper from the statistics
    (3, 4096, 12288, 3, 1, "abs", True, -1, False, False),         # 16 values per thread in the sub-transform
    (4, 5000, 20000, 2, 1, "fourier", False, 0, False, True),      # radix 4: the factors +-1, +-i
])
def test_long_trials_through_scratch_memory(P, M, nsig, nchan, K, output, keeptapers, detrend, demean, refmean):
    """K1L2 (mtmfft_declong.h): N = P M beyond one workgroup's LDS - statistics, scheduled sub-transforms of the samples
    P m + r, radix-P step + separation + conversion, against the oracle."""
    nfft = P * M
    rng = np.random.default_rng(P * M + nchan)
    data = (rng.normal(size=(nsig + 9, nchan)) + (5.0 if detrend == 0 else 0.0)).astype("f4")
    ss = np.array([4])
    taper, topt = ("dpss", {"NW": (K + 1) / 2, "Kmax": K}) if K > 1 else ("hann", {})
    tapers = O.taper_table(taper, nsig, nfft, topt)
    fi = np.array([0, 1, M - 1, M, M + 1, nfft // 2 - 1, nfft // 2, 37]) if output == "abs" else None
    out = E.fft_exec_declong(data, ss, ss, ss + nsig, nsig, P, M, tapers, O.spec_scale(nsig, nfft), detrend, demean, fi, output,
                             keeptapers, reference_mean=refmean)
    freqs = np.fft.rfftfreq(nfft, 1e-3)
    ref, _ = O.mtmfft_cF(np.array(data[4:4 + nsig]), foi=freqs if fi is None else freqs[fi], keeptapers=keeptapers,
                         polyremoval=None if detrend < 0 else detrend, output=output,
                         method_kwargs=dict(samplerate=1000.0, taper=taper, taper_opt=topt, nSamples=nfft, demean_taper=demean))
    assert_parity(out[0], ref[0], what=f"{P} x {M}")


@pytest.mark.parametrize("nsig,nfft,nchan,K,output,keeptapers,detrend,demean", [
    (2000, 2000, 8, 2, "fourier", True, 0, False),     # pairs through 10 x 10 x 10, two pairs per workgroup, fast stores
    (2000, 2000, 5, 3, "pow", False, 1, True),         # ragged channels, taper mean, line fit over even + odd samples, demean_taper
    (1999, 2000, 3, 2, "abs", True, 0, False),         # odd sample count: the last pair is half padding
    (1501, 2000, 4, 2, "fourier", False, -1, False),   # zero padding, complex taper mean
    (2000, 2002, 6, 2, "fourier", True, 0, False),     # ... with split exchanges (id 2002 = the 1000-point schedule, SPLIT)
    (1200, 1200, 9, 2, "fourier", True, 0, True),      # 3 x 200: radix-3 decimation in front, four pairs per workgroup
    (1100, 1200, 4, 3, "pow", False, 1, False),
    (1024, 1024, 6, 2, "pow", True, 0, False),         # 16 x 16 x 2
    (5000, 5000, 4, 1, "pow", True, 0, False),         # 10 x 10 x 5 x 5 (the product's choice for nfft = 5000)
    (12000, 12000, 2, 1, "fourier", True, 0, False),   # 3 x (10 x 10 x 10 x 2): trials beyond a quad's LDS
])
def test_dec_kernel_half_form(nsig, nfft, nchan, K, output, keeptapers, detrend, demean):
    """CfgD::HALF: z[m] = x[2 m] + i x[2 m + 1] through the length-nfft/2 schedule, bins f and nfft/2 - f from Z[f], Z[nfft/2 - f]."""
    real_nfft = 2000 if nfft == 2002 else nfft
    _fft_case(nsig, real_nfft, nchan, K, output, keeptapers, detrend, demean_taper=demean, nseg=2 if nfft <= 2002 else 1, dec=-nfft)


def test_dec_kernel_half_form_selection():
    _fft_case(1700, 2000, 6, 2, "pow", True, 0, dec=-2000, nseg=1, freq_idx=np.array([0, 1, 999, 1000, 37, 500, 501, 499]),
              chan_idx=np.array([5, 0, 2, 2, 1]))
    _fft_case(1200, 1200, 3, 2, "fourier", False, 0, dec=-1200, nseg=1, freq_idx=np.array([600, 0, 300, 299, 301, 200, 400]))


def test_dec_kernel_padding_and_selection():
    _fft_case(1700, 2000, 6, 2, "pow", True, 0, dec=2000, nseg=1, freq_idx=np.array([0, 1, 999, 1000, 37]),
              chan_idx=np.array([5, 0, 2, 2]))
    _fft_case(900, 1000, 4, 2, "fourier", True, 1, dec=1000, nseg=2)


def test_csd_3m_more_than_512_channels():
    """More than 512 channels (csd.hip): Hermitian 3M products of the 256-channel blocks + the rectangle kernel
    (csd3m_kernel<512, 8, false, true>: two channel ranges side by side in one LDS image) for every pair of blocks; an odd
    count sends its last row through the rank-1 update.  520 channels = blocks of 256, 256, 8."""
    C, F, R = 521, 1, 6
    rng = np.random.default_rng(3)
    spec = (rng.normal(size=(R, F, C)) + 1j * rng.normal(size=(R, F, C))).astype(np.complex64)
    acc = np.zeros((F, C, C), np.complex64)
    assert E.csd_accumulate(spec[:2], acc) == 10 and E.csd_accumulate(spec[2:], acc) == 10
    ref = np.einsum("rfi,rfj->fij", spec.astype(np.complex128), spec.conj().astype(np.complex128))
    ii, jj = np.tril_indices(C)
    assert_parity(acc[:, ii, jj], ref[:, ii, jj].astype(np.complex64), what="wide csd, lower triangle")
    iu, ju = np.triu_indices(C, 1)
    off = iu // 16 != ju // 16                           # (a diagonal 16 x 16 tile is stored whole, as for <= 512)
    assert not acc[:, iu[off], ju[off]].any()            # nothing else lands above the diagonal
    # phase-exact accumulation: the same walk with the 4-multiplication instances (csd3m_kernel<..., M4 = true>)
    acc4 = np.zeros((F, C, C), np.complex64)
    assert E.csd_accumulate(spec, acc4, force_4m=True) == 11
    assert_parity(acc4[:, ii, jj], ref[:, ii, jj].astype(np.complex64), what="wide csd, 4 multiplications")
    # its imaginary part is a sum of products, not a difference of the three 3M products: error ~ eps * |Im| terms only
    coherent = np.repeat(spec[:, :, :1], C, axis=2) * (1 + 1e-3 * rng.normal(size=(1, 1, C))).astype(np.float32)
    acc4[:] = 0
    E.csd_accumulate(coherent.astype(np.complex64), acc4, force_4m=True)
    scale = np.abs(acc4[0, 0, 0])
    assert np.abs(acc4[:, ii, jj].imag).max() <= 1e-6 * scale       # real multiples of one signal: Im = 0 up to rounding


# K1r second generation: reference-precision transforms (mtmfft_dec64_kernel.h, mtmfft_f64_kernel.h incl. Bluestein)
def _f64_case(nsig, nfft, nchan, K, output, keeptapers, detrend, demean_taper=False, freq_idx=None, chan_idx=None, nseg=1,
              seed=3, dec=True, bluestein=False, pure_frac=0.99, emu_id=None):
    """Data with 60 dB of dynamic range (a 40 Hz line 1000 x the noise, an offset): the criterion everywhere, and for
    complex output PURE rtol 1e-5 bin by bin on >= 99 % of the bins - which only a float64 transform delivers."""
    rng = np.random.default_rng(seed)
    t = np.arange(nsig * nseg + 9) / 1000.0
    data = (rng.normal(size=(nsig * nseg + 9, nchan)) + 1000.0 * np.sin(2 * np.pi * 40.0 * t)[:, None] + 50.0).astype("f4")
    ss = np.array([4 + i * nsig for i in range(nseg)])
    taper, topt = ("dpss", {"NW": (K + 1) / 2, "Kmax": K}) if K > 1 else ("hann", {})
    tapers = O.taper_table(taper, nsig, nfft, topt)
    out = E.fft_exec_f64(data, ss, ss, ss + nsig, nsig, nfft, tapers, O.spec_scale(nsig, nfft), detrend, demean_taper,
                         freq_idx, output, keeptapers, chan_idx=chan_idx, reference_mean=True, dec=dec, bluestein=bluestein,
                         emu_id=emu_id)
    freqs = np.fft.rfftfreq(nfft, 1e-3)
    foi = freqs if freq_idx is None else freqs[freq_idx]
    for b in range(nseg):
        x = data[ss[b]:ss[b] + nsig]
        if chan_idx is not None:
            x = x[:, chan_idx]
        ref, _ = O.mtmfft_cF(np.array(x, order="C"), foi=foi, keeptapers=keeptapers, polyremoval=None if detrend < 0 else detrend,
                             output=output, method_kwargs=dict(samplerate=1000.0, taper=taper, taper_opt=topt,
                                                               nSamples=nfft, demean_taper=demean_taper))
        assert_parity(out[b], ref[0], what=f"segment {b}")
        if output == "fourier" and detrend != 1:
            err = np.abs(out[b].astype(np.complex128) - ref[0])
            frac = float((err <= 1e-5 * np.abs(ref[0])).mean())
            assert frac >= pure_frac, (nfft, frac)


@pytest.mark.parametrize("nfft,nchan,K", [(256, 33, 2), (1024, 9, 2), (4096, 3, 2),
                                          (200, 17, 2), (1000, 5, 2), (2000, 3, 2), (2500, 2, 1)])
def test_dec64_kernel_vs_oracle(nfft, nchan, K):
    _f64_case(nfft, nfft, nchan, K, "fourier", True, 0)


@pytest.mark.parametrize("nfft", [8192, 16384, 4000, 5000, 10000])
def test_dec64_kernel_long_schedules(nfft):
    # 8192: four passes; 16384: 1024 threads, split exchange, samples re-read per taper; 10000: split exchange, V = 20
    _f64_case(nfft, nfft, 2, 1, "fourier", True, 0)


@pytest.mark.parametrize("nfft,nchan,K", [(100, 35, 2), (300, 17, 2), (400, 9, 2), (2400, 3, 1), (3200, 2, 1), (600, 9, 2), (768, 5, 2), (1500, 5, 2), (3000, 3, 2), (3072, 2, 1), (6000, 2, 1),
                                          (7500, 2, 1)])
def test_dec64_kernel_radix3_decimation(nfft, nchan, K):
    # N = 3 M: three scheduled sub-transforms side by side and one radix-3 combine through LDS (CfgD64::P)
    _f64_case(nfft, nfft, nchan, K, "fourier", True, 0)


def test_dec64_kernel_radix3_options():
    _f64_case(2800, 3000, 3, 2, "pow", False, 1)                                  # padding, linear trend, taper mean
    _f64_case(1400, 1500, 5, 2, "abs", True, 0, freq_idx=np.array([0, 1, 749, 750, 37, 500, 501]), chan_idx=[4, 0, 3])
    _f64_case(600, 600, 7, 3, "fourier", False, -1, demean_taper=True, nseg=2)    # complex taper mean, demean_taper
    _f64_case(3000, 3000, 1, 2, "pow", True, 0)                                   # a single channel: half a pair


def test_dec64_kernel_options():
    _f64_case(900, 1000, 4, 2, "pow", False, 1)                                   # padding, linear trend, taper mean
    _f64_case(1700, 2000, 6, 2, "abs", True, 0, freq_idx=np.array([0, 1, 999, 1000, 37]), chan_idx=[5, 0, 3])
    _f64_case(256, 256, 7, 3, "fourier", False, -1, demean_taper=True, nseg=2)    # complex taper mean, demean_taper
    _f64_case(900, 1024, 2, 1, "imag", True, 0, nseg=2)
    _f64_case(4096, 4096, 1, 2, "pow", True, 0)                                   # a single channel: half a pair


@pytest.mark.parametrize("nsig,nfft,nchan,K,output,keeptapers,detrend,demean,emu_id", [
    (2000, 2000, 5, 2, "fourier", True, 0, False, None),     # single channels through 10 x 10 x 10, two per workgroup
    (2000, 2000, 3, 3, "pow", False, 1, True, None),         # taper mean, line fit over even + odd samples, demean_taper
    (1999, 2000, 2, 2, "abs", True, 0, False, None),         # odd sample count: the last pair is half padding
    (1501, 2000, 3, 2, "fourier", False, -1, False, None),   # zero padding, complex taper mean
    (2000, 2000, 4, 2, "fourier", True, 0, False, 2002),     # split exchanges
    (1200, 1200, 5, 2, "fourier", True, 0, True, None),      # 3 x 200: radix-3 decimation in front
    (1100, 1200, 3, 2, "pow", False, 1, False, None),
    (1024, 1024, 3, 2, "pow", True, 0, False, None),         # powers from the table (no hoisted base twiddles)
    (12000, 12000, 1, 1, "fourier", True, 0, False, None),   # the product's schedule for nfft = 12000
])
def test_dec64_kernel_half_form(nsig, nfft, nchan, K, output, keeptapers, detrend, demean, emu_id):
    """CfgD64::HALF: one channel = one complex128 transform of (even, odd) samples through the schedule of nfft / 2."""
    _f64_case(nsig, nfft, nchan, K, output, keeptapers, detrend, demean_taper=demean, nseg=2 if nfft <= 2000 else 1,
              dec="half", emu_id=emu_id)


def test_dec64_kernel_half_form_selection():
    _f64_case(1700, 2000, 5, 2, "pow", True, 0, dec="half", freq_idx=np.array([0, 1, 999, 1000, 37, 500, 501, 499]),
              chan_idx=np.array([4, 0, 2, 2]))
    _f64_case(1200, 1200, 3, 2, "fourier", False, 0, dec="half", freq_idx=np.array([600, 0, 300, 299, 301, 200, 400]))


@pytest.mark.parametrize("nfft,nchan,K,bluestein", [(360, 3, 2, False), (1009, 2, 1, True), (134, 3, 2, True),
                                                     (3001, 1, 1, True), (268, 2, 1, False)])
def test_f64_any_kernel_and_bluestein_vs_oracle(nfft, nchan, K, bluestein):
    # 1009 and 3001 are primes, 134 = 2 x 67, 268 = 4 x 67 through the O(R^2) pass AND through the chirp-z form
    _f64_case(nfft, nfft, nchan, K, "fourier", True, 0, dec=False, bluestein=bluestein)
    if nfft == 268:
        _f64_case(nfft, nfft, nchan, K, "fourier", True, 0, dec=False, bluestein=True)
