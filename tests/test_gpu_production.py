"""Parity at the BASELINE.json production shapes ("Next round" item 1): the kernels that serve
c2/c3 (256 ch x 4096, K1 <12,...> + K4 <5,4,1>), c4 (128 ch x 16384: every CWT block group, the 512-sample sliding
window) and c5 (Wilson / Granger with the 16x16 blocked inverse, the fp64-MFMA zgemm tiles and the condition-number
iteration at C >= 16 ... 256) are compared with the oracle at the sizes they are benchmarked on - through the front
ends and the C ABI, a few trials each so the CPU oracle finishes in seconds."""
import numpy as np
import pytest

import syncopy_amd as spy
from oracle import spy_oracle as O
from oracle_routines import ORACLE_FREQ
from parity import assert_parity, excess

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from syncopy_amd import backend
    backend.require_gpu()


# ------------------------------------------------------------------------------------------------ c2 / c3
@pytest.fixture(scope="module")
def c3_data():
    """4 trials of 256 ch x 4096 samples, AR(2) with a handful of couplings (so that coherence is not flat) and one
    channel with an offset + a 50 Hz line (dynamic range the AR(2) spectra do not have; the offset makes the bins next
    to DC depend on the ORDER of the float32 mean - the reference sums the rows sequentially)."""
    adj = np.zeros((256, 256))
    for i, j in ((0, 1), (10, 200), (255, 3), (128, 129)):
        adj[i, j] = 0.25
    data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=4096, nTrials=4, seed=21)
    t = np.arange(4096 * 4) / 1000.0
    data.data[:, 7] += (3.0 + 5.0 * np.sin(2 * np.pi * 50.0 * t)).astype(np.float32)
    return data


def _trials(data):
    td = np.asarray(data.trialdefinition)
    return [np.array(data.data[int(a):int(b)]) for a, b in td[:, :2]]


def test_c2_mtmfft_pow_256x4096(c3_data):
    """BASELINE configs[1] shape through spy.freqanalysis: K1 <12,1,0,true> at 256 channels vs the oracle's float64
    taper + rfft (mtmfft.py:96-127, compRoutines.py:169-189)."""
    out = spy.freqanalysis(c3_data, method="mtmfft", tapsmofrq=1)
    assert out.data.shape == (4, 1, 2049, 256) and out.data.dtype == np.float32
    mk = dict(samplerate=1000.0, taper="dpss", taper_opt={"NW": 4.096, "Kmax": 7}, nSamples=4096, demean_taper=False)
    for t, trl in enumerate(_trials(c3_data)):
        ref, _ = O.mtmfft_cF(trl, foi=np.fft.rfftfreq(4096, 1e-3), keeptapers=False, polyremoval=0, output="pow",
                             method_kwargs=mk)
        assert_parity(out.data[t], ref[0], what=f"c2 pow trial {t}")


def test_c2_fourier_keeptapers_256x4096(c3_data):
    """The coherence front half (complex spectra of every taper, K1 <12,2,2,false>) at 256 channels."""
    out = spy.freqanalysis(c3_data, method="mtmfft", tapsmofrq=1, output="fourier", keeptapers=True,
                           select={"trials": [0, 3]})
    assert out.data.shape == (2, 7, 2049, 256) and out.data.dtype == np.complex64
    mk = dict(samplerate=1000.0, taper="dpss", taper_opt={"NW": 4.096, "Kmax": 7}, nSamples=4096, demean_taper=False)
    trials = _trials(c3_data)
    for k, t in enumerate((0, 3)):
        ref, _ = O.mtmfft_cF(trials[t], foi=np.fft.rfftfreq(4096, 1e-3), keeptapers=True, polyremoval=0,
                             output="fourier", method_kwargs=mk)
        assert_parity(out.data[k], ref[0], what=f"c2 fourier trial {t}")


def test_c3_coherence_256x4096(c3_data):
    """BASELINE configs[2] shape through spy.connectivityanalysis: K1 + K4 <5,4,1> (+ tail) + fused K5 vs the
    oracle's per-trial csd (einsum form of csd.py:94-102, complex64 products), the sequential complex64 trial sum
    (computational_routine.py:1022-1032) and normalize_csd (csd.py:118-172)."""
    coh = spy.connectivityanalysis(c3_data, method="coh", tapsmofrq=1)
    csd = spy.connectivityanalysis(c3_data, method="csd", tapsmofrq=1)
    assert coh.data.shape == (1, 2049, 256, 256) and coh.data.dtype == np.float32
    acc = np.zeros((2049, 256, 256), dtype=np.complex64)
    for trl in _trials(c3_data):
        specs, _ = O.mtmfft(O.detrend(trl, 0), 1000.0, 4096, "dpss", {"NW": 4.096, "Kmax": 7})
        for f0 in range(0, 2049, 64):            # O.csd(faithful=False), 64 frequencies at a time (cache-sized)
            s = specs[:, f0:f0 + 64]
            acc[f0:f0 + 64] += (np.einsum("kfi,kfj->fij", s, s.conj()) / 7).astype(np.complex64)
    acc /= 4
    assert_parity(csd.data[0], acc, what="c3 csd")
    sub = np.r_[0:2049:8, 2047, 2048]            # NumPy's complex64 sqrt takes ~10 s per 1000 bins: every 8th bin
    assert_parity(coh.data[0][sub], O.normalize_csd(acc[sub], "abs"), what="c3 coherence")
    assert np.array_equal(coh.data[0], coh.data[0].transpose(0, 2, 1))
    assert np.all(np.abs(coh.data[0][:, np.arange(256), np.arange(256)] - 1) < 1e-6)


@pytest.mark.parametrize("N,T,kernel", [(12000, 30, "HALF of N = 12000"), (24000, 26, "declong<6 x 4000")])
def test_long_trials_256_channels(N, T, kernel):
    """Trials beyond a channel quad's LDS at 256 channels x 7 tapers.  12000: channel pairs through the 6000-point schedule
    (CfgD::HALF).  24000 = 6 x 4000 through HBM (K1L2, mtmfft_declong.h): 172 MB of scratch per trial, so 26 trials take
    three chunks of the 2-GiB scratch.  Against a float64 rfft of the same tapered trials formed on the device
    (mtmfft.py:96-127; no detrending: the reference-order mean is tests/test_gpu_kernels.py's subject), every 5th trial."""
    torch = pytest.importorskip("torch")
    from scipy.signal import windows
    from syncopy_amd import backend as be
    C, K = 256, 7
    g = torch.Generator(device="cuda").manual_seed(12)
    data = torch.randn((T * N, C), device="cuda", dtype=torch.float32, generator=g)
    tap = windows.dpss(N, 4.0, K) * np.sqrt(N)
    starts = torch.arange(T, device="cuda", dtype=torch.int64) * N
    for output, keeptapers in (("pow", False), ("fourier", True)):
        plan = be.FFTPlan(N, N, C, tap, np.sqrt(2) / N, None, False, None, output, keeptapers)
        assert kernel in plan.kernel_name
        got = plan.execute(data, starts)
        w = torch.from_numpy(tap).cuda()                                   # (K, N) float64
        for t in range(0, T, 5):
            x = data[t * N:(t + 1) * N].to(torch.float64)
            spec = torch.fft.rfft(w[:, :, None] * x[None], dim=1) * (np.sqrt(2) / N)          # (K, F, C)
            if output == "pow":
                ref = (spec.real ** 2 + spec.imag ** 2).mean(dim=0).to(torch.float32).cpu().numpy()
                assert_parity(got[t, 0].cpu().numpy(), ref, what=f"pow trial {t}")
            else:
                assert_parity(got[t].cpu().numpy(), spec.to(torch.complex64).cpu().numpy(), what=f"fourier trial {t}")
            del spec, x
        del got, plan


# ------------------------------------------------------------------------------------------------ c4
@pytest.fixture(scope="module")
def c4_data():
    return spy.synthdata.ar2_network(AdjMat=np.zeros((128, 128)), nSamples=16384, nTrials=1, seed=4)


def test_c4_mtmconvol_128x16384(c4_data):
    """BASELINE configs[3] (i): 512-sample Hann windows, 50 % overlap (K2 at N = 512, 128 channels)."""
    kw = dict(method="mtmconvol", taper="hann", t_ftimwin=0.512, toi=0.5, output="pow")
    got = spy.freqanalysis(c4_data, **kw)
    ref = spy.freqanalysis(c4_data, compute_method="sequential", routine_classes=ORACLE_FREQ, **kw)
    assert got.data.shape == ref.data.shape == (64, 1, 257, 128)
    assert_parity(got.data, ref.data, what="c4 mtmconvol")


def test_c4_wavelet_128x16384(c4_data):
    """BASELINE configs[3] (ii): Morlet, 25 scales 4 ... 100 Hz, toi='all' - hits every CWT block group
    (1024 / 2048 / 4096 / 8192-point blocks) at 128 channels (transform.py:88-108)."""
    kw = dict(method="wavelet", wavelet="Morlet", width=6, foi=np.arange(4, 104, 4), toi="all", output="pow")
    got = spy.freqanalysis(c4_data, **kw)
    ref = spy.freqanalysis(c4_data, compute_method="sequential", routine_classes=ORACLE_FREQ, **kw)
    assert got.data.shape == ref.data.shape == (16384, 1, 25, 128)
    assert_parity(got.data, ref.data, what="c4 wavelet")


# ------------------------------------------------------------------------------------------------ c5
def _var_csd(C, F, seed, floor=0.05):
    """Spectral matrix of a random stable VAR(2) process on F rfft bins: S(f) = H(f) Sigma H(f)^H + floor, complex64
    (what the ST stage hands to the AV stage, AV_compRoutines.py:395)."""
    rng = np.random.default_rng(seed)
    A1 = 0.5 * np.eye(C) + rng.normal(size=(C, C)) * (0.25 / np.sqrt(C))
    A2 = -0.6 * np.eye(C) + rng.normal(size=(C, C)) * (0.15 / np.sqrt(C))
    L = np.eye(C) + 0.1 * np.tril(rng.normal(size=(C, C)), -1)
    Sigma = L @ L.T
    w = np.pi * np.arange(F) / (F - 1)
    if C >= 128:
        # test fixture only (2049 inversions of 256 x 256 take a minute in NumPy): the spectrum of a moving-average
        # process instead, S = B Sigma B^H + floor with B(f) = I + B1 e^{-iw} + B2 e^{-2iw}, products on the device
        t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.complex128)).cuda()      # noqa: E731
        B = t(np.eye(C))[None] + t(A1 - 0.2 * np.eye(C))[None] * t(np.exp(-1j * w))[:, None, None] \
            + t(A2 + 0.75 * np.eye(C))[None] * t(np.exp(-2j * w))[:, None, None]
        S = B @ t(Sigma)[None] @ B.conj().transpose(1, 2) + floor * t(np.eye(C))[None]
        S = 0.5 * (S + S.conj().transpose(1, 2))
        return S.to(torch.complex64).cpu().numpy()
    A = np.eye(C)[None] - A1[None] * np.exp(-1j * w)[:, None, None] - A2[None] * np.exp(-2j * w)[:, None, None]
    H = np.linalg.inv(A)
    S = H @ Sigma[None] @ H.conj().transpose(0, 2, 1) + floor * np.eye(C)[None]
    S = 0.5 * (S + S.conj().transpose(0, 2, 1))
    return S.astype(np.complex64)


@pytest.mark.parametrize("C", [16, 33, 64])
def test_wilson_granger_vs_oracle(C):
    """K6 at channel counts that use the 16x16 blocked inverse and the 32x32 zgemm tiles (16: one block; 33: ragged
    edge; 64: several tiles) against O.wilson_sf / O.granger / O.regularize_csd on the same complex64 CSD."""
    from syncopy_amd import backend
    F = 65
    csd = _var_csd(C, F, seed=C)
    G, meta, H, Sigma = backend.granger(torch.from_numpy(csd).cuda(), want_factors=True)
    G, H, Sigma = G.cpu().numpy(), H.cpu().numpy(), Sigma.cpu().numpy()
    reg, factor, cn0 = O.regularize_csd(csd, cond_max=1e4, eps_max=1e-1)
    Ho, So, conv, err = O.wilson_sf(reg.astype(np.complex128), nIter=100, rtol=5e-6)
    assert conv and meta["converged"] and meta["max rel. err"] < 5e-6
    assert meta["reg. factor"] == factor
    np.testing.assert_allclose(meta["initial cond. num"], cn0, rtol=1e-4)
    # the reference's own acceptance test (tests/backend/test_conn.py:197-202)
    assert O.max_rel_err(reg.astype(np.complex128), H @ Sigma @ H.conj().transpose(0, 2, 1)) < 1e-5
    np.testing.assert_allclose(H, Ho, rtol=2e-4, atol=2e-5 * np.abs(Ho).max())
    np.testing.assert_allclose(Sigma, So, rtol=2e-4, atol=2e-5 * np.abs(So).max())
    Go = O.granger(reg.astype(np.complex128), Ho, So)
    np.testing.assert_allclose(G, Go, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("C,F", [(8, 4097), (6, 8193), (5, 2501), (4, 3001)])
def test_wilson_long_lag_domain_vs_oracle(C, F):
    """Lag-domain lengths 2(F-1) whose working arrays do not fit LDS (8192, 16384: trials of 5000 samples padded to
    the next power of two, trials of 16384 samples) and lengths above 4096 that are not powers of two (5000, 6000: the
    generic LDS kernel) - wilson_sf.py:154-184 has no length limit.  The plus operator then works in global scratch."""
    from syncopy_amd import backend
    csd = _var_csd(C, F, seed=F)
    G, meta, H, Sigma = backend.granger(torch.from_numpy(csd).cuda(), want_factors=True)
    G, H, Sigma = G.cpu().numpy(), H.cpu().numpy(), Sigma.cpu().numpy()
    reg, factor, cn0 = O.regularize_csd(csd, cond_max=1e4, eps_max=1e-1)
    Ho, So, conv, err = O.wilson_sf(reg.astype(np.complex128), nIter=100, rtol=5e-6)
    assert conv and meta["converged"] and meta["max rel. err"] < 5e-6 and meta["reg. factor"] == factor
    assert O.max_rel_err(reg.astype(np.complex128), H @ Sigma @ H.conj().transpose(0, 2, 1)) < 1e-5
    np.testing.assert_allclose(H, Ho, rtol=2e-4, atol=2e-5 * np.abs(Ho).max())
    np.testing.assert_allclose(Sigma, So, rtol=2e-4, atol=2e-5 * np.abs(So).max())
    np.testing.assert_allclose(G, O.granger(reg.astype(np.complex128), Ho, So), rtol=2e-3, atol=2e-4)


def test_granger_long_trials_through_the_front_end():
    """spy.connectivityanalysis(method='granger') on trials of 16384 samples (F = 8193) and on 5000-sample trials with
    pad='nextpow2' (F = 4097) against the oracle-bound front end (missing 1)."""
    from oracle_routines import ORACLE_CONN
    adj = np.zeros((8, 8))
    adj[0, 1] = adj[3, 5] = 0.3
    for nSamples, pad in ((16384, "maxperlen"), (5000, "nextpow2")):
        data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=nSamples, nTrials=12, seed=3)
        kw = dict(method="granger", tapsmofrq=3, pad=pad)
        got = spy.connectivityanalysis(data, **kw)
        ref = spy.connectivityanalysis(data, compute_method="sequential", routine_classes=ORACLE_CONN, **kw)
        assert got.data.shape == ref.data.shape
        # bin 0 is excluded: method='granger' removes the mean of every tapered segment (demean_taper, mtmfft.py:115-116),
        # so X(0) is the rounding noise of that subtraction - in float64 for the reference, in float32 here - and the
        # Granger ratio at DC is a ratio of two noises on either side
        np.testing.assert_allclose(got.data[:, 1:], ref.data[:, 1:], rtol=1e-2, atol=1e-2)   # Granger atol 1e-2
        # (the oracle's own convergence flag can hinge on that noise bin: its relative error |A - psi psi^H| / |A| is
        # taken over all bins, bin 0 included, where |A| ~ 1e-26 of the spectrum in float64)
        assert bool(got.info["converged"])
        if bool(ref.info["converged"]):
            assert abs(got.info["max rel. err"]) < 5e-6


@pytest.mark.parametrize("C,F", [(16, 65), (33, 129), (64, 257), (256, 2049)])
def test_wilson_steps_equal_monolithic(C, F):
    """The stepped K6 entry points (spyhip_wilson_*, the frequency-shard ABI) driven by wilson_sharded.granger_sharded
    with one rank run the same kernels in the same order as spyhip_granger: same iteration count, same values."""
    from syncopy_amd import backend
    from syncopy_amd.connectivity.wilson_sharded import HipPrims, granger_sharded
    csd = torch.from_numpy(_var_csd(C, F, seed=C + 1)).cuda()
    G0, meta0, H0, S0 = backend.granger(csd, want_factors=True)
    it0 = backend.granger_stats()["iterations"]
    G1, meta1, H1, S1 = granger_sharded(csd, 0, F, HipPrims(csd.device))
    assert meta1["converged"] and meta0["converged"] and meta1["iterations"] == it0
    assert meta1["reg. factor"] == meta0["reg. factor"]
    np.testing.assert_allclose(meta1["initial cond. num"], meta0["initial cond. num"], rtol=1e-9)
    # compared on the device: at 256 x 2049 the factors are 2 GB each
    assert float((H1 - H0).abs().max()) <= 1e-9 * float(H0.abs().max())
    assert float((S1 - S0).abs().max()) <= 1e-9 * float(S0.abs().max())
    assert bool(torch.isfinite(G1).all()) and float((G1 - G0).abs().max()) <= 1e-5 * float(G0.abs().max()) + 1e-7
    if C == 256:
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        granger_sharded(csd, 0, F, HipPrims(csd.device))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        backend.granger(csd)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"\nWilson 2049 x 256 x 256: stepped ABI (one shard) {t1 - t0:.3f} s, spyhip_granger {t2 - t1:.3f} s")
        assert t1 - t0 < 3.0 * (t2 - t1) + 1.0               # the steps add allocations and copies, not another algorithm


def test_wilson_shards_of_one_device():
    """Two and three frequency shards emulated on ONE device: per-shard steps on slices, the exchanges replaced by
    slicing/concatenation - checks the f_lo/nftot and nent arguments of the step ABI against the unsharded call."""
    from syncopy_amd import backend
    from syncopy_amd import parallel
    from syncopy_amd.connectivity.wilson_sharded import HipPrims
    C, F = 33, 129
    csd = torch.from_numpy(_var_csd(C, F, seed=9)).cuda()
    G0, meta0, H0, S0 = backend.granger(csd, want_factors=True)
    it0 = backend.granger_stats()["iterations"]
    P = HipPrims(csd.device)
    for R in (2, 3):
        fb, eb = parallel.shard_bounds(F, R), parallel.shard_bounds(C * C, R)
        A, U, gam = [], [], 0
        for lo, hi in fb:
            a, _ = P.cond(csd[lo:hi].contiguous(), 0.0)
            u, gp = P.init(a, lo, F)
            A.append(a); U.append(u); gam = gam + gp
        psi0s, psis = zip(*[P.psi0(gam.clone(), hi - lo) for lo, hi in fb])
        for it in range(it0):
            g = torch.cat([P.g(psis[r], U[r], False)[0].reshape(-1, C * C) for r in range(R)], dim=0)
            parts = [P.plus(g[:, lo:hi].contiguous()) for lo, hi in eb]
            gp = torch.cat([p[0] for p in parts], dim=1)
            g0 = torch.cat([p[1] for p in parts]).reshape(C, C)
            err = max(P.update(psis[r], gp[lo:hi].reshape(-1, C, C).contiguous(), g0, psi0s[r], A[r])
                      for r, (lo, hi) in enumerate(fb))
        assert err < 5e-6
        outs = [P.finish(A[r], psis[r], psi0s[r]) for r in range(R)]
        G = torch.cat([o[0] for o in outs], dim=0)
        H = torch.cat([o[1] for o in outs], dim=0)
        np.testing.assert_allclose(H.cpu().numpy(), H0.cpu().numpy(), rtol=1e-8, atol=1e-10 * float(H0.abs().max()))
        np.testing.assert_allclose(G.cpu().numpy(), G0.cpu().numpy(), rtol=1e-5, atol=1e-7)
        for o in outs:
            np.testing.assert_allclose(o[2].cpu().numpy(), S0.cpu().numpy(), rtol=1e-9, atol=1e-11 * float(S0.abs().max()))


def test_wilson_reconstruction_256x2049():
    """BASELINE configs[4] AV stage at full size: the CSD of 120 trials x 7 tapers of 256 ch x 4096 (K1 + K4 on the
    device), factorised by K6; acceptance = the reference's max_rel_err(CSD, H Sigma H^H) (test_conn.py:197-202),
    evaluated with complex128 batched products on the device (checker only)."""
    from syncopy_amd import backend
    C, N, T = 256, 4096, 120
    x = spy.synthdata.ar2_uncoupled_fast(C, N, T, seed=5)
    x[:, 1:] += 0.3 * x[:, :-1]                                   # instantaneous mixing: off-diagonal CSD entries
    trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
    data = spy.AnalogData(x.cpu().numpy(), samplerate=1000.0, trialdefinition=trl)
    del x
    csd = spy.connectivityanalysis(data, method="csd", tapsmofrq=1)
    S = torch.from_numpy(np.ascontiguousarray(csd.data[0])).cuda()
    G, meta, H, Sigma = backend.granger(S, want_factors=True)
    assert meta["converged"] and meta["max rel. err"] < 5e-6 and meta["reg. factor"] == 0
    rec = H @ Sigma.unsqueeze(0) @ H.conj().transpose(1, 2)
    S128 = S.to(torch.complex128)
    err = float(((S128 - rec).abs() / S128.abs()).max())
    assert err < 1e-5, err
    assert bool(torch.isfinite(G).all())
    # Granger of (nearly) uncoupled channels is ~0 everywhere off the coupled neighbours; never negative beyond noise
    assert float(G.min()) > -1e-3


def c5_dataset(T=2560, C=256, N=4096, seed=6):
    """BASELINE configs[4] at one GPU's share: AR(2) channels with an instantaneous mixing of neighbours (off-diagonal
    CSD entries), 10 trials per channel (the reference's rule, connectivity_analysis.py:808).  Generated on the device
    (not the reference's random stream), handed over as host-resident AnalogData like any user's recording."""
    x = spy.synthdata.ar2_uncoupled_fast(C, N, T, seed=seed)
    x[:, 1:] += 0.3 * x[:, :-1]
    trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
    data = spy.AnalogData(x.cpu().numpy(), samplerate=1000.0, trialdefinition=trl)
    del x
    torch.cuda.empty_cache()
    return data


def test_c5_granger_front_end_256x4096():
    """BASELINE configs[4] as ONE front-end call (VERDICT r3 missing 2): spy.connectivityanalysis(method="granger",
    tapsmofrq=1) on 256 ch x 4096 samples x 2560 trials - the ST stage with demean_taper=True
    (connectivity_analysis.py:864), the trial average, regularize_csd + wilson_sf + granger (AV_compRoutines.py:293-412).
    Acceptance: converged below the reference's rtol, H Sigma H^H reconstructs the CSD (test_conn.py:197-202), and the
    call equals its two stages run by hand (CrossSpectra with demean_taper -> backend.granger)."""
    import time
    from syncopy_amd import backend
    from syncopy_amd.connectivity.ST_compRoutines import CrossSpectra
    from syncopy_amd.datatype import CrossSpectralData
    from syncopy_amd.shared.input_processors import process_taper
    C, N, T = 256, 4096, 2560
    data = c5_dataset(T, C, N)
    t0 = time.perf_counter()
    out = spy.connectivityanalysis(data, method="granger", tapsmofrq=1)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = spy.connectivityanalysis(data, method="granger", tapsmofrq=1)
    t_warm = time.perf_counter() - t0
    print(f"c5 on one GPU, {T} trials: first call {t_first:.2f} s (upload of {data.data.nbytes / 1e9:.1f} GB included), "
          f"warm call {t_warm:.2f} s, iterations {backend.granger_stats()['iterations']}")
    assert out.data.shape == (1, N // 2 + 1, C, C) and out.data.dtype == np.float32
    assert out.info["converged"] and out.info["max rel. err"] < 5e-6 and out.info["reg. factor"] == 0
    # the two stages by hand
    freqs = np.fft.rfftfreq(N, 1e-3)
    taper, taper_opt = process_taper("hann", None, 1, None, keeptapers=False, foimax=freqs.max(), samplerate=1000.0,
                                     nSamples=N, output="pow")
    st = CrossSpectra(samplerate=1000.0, nSamples=N, taper=taper, taper_opt=taper_opt, demean_taper=True, polyremoval=0,
                      timeAxis=0, foi=freqs)
    st_out = CrossSpectralData(dimord=CrossSpectra.dimord)
    st.initialize(data, st_out._stackingDim, chan_per_worker=None, keeptrials=False)
    st.compute(data, st_out, parallel=False, log_dict={}, method="hip")
    S = st_out._dev[0].contiguous()
    G, meta, H, Sigma = backend.granger(S, want_factors=True)
    assert meta["converged"] and meta["max rel. err"] == out.info["max rel. err"]
    assert np.array_equal(G.cpu().numpy(), out.data[0])
    rec = H @ Sigma.unsqueeze(0) @ H.conj().transpose(1, 2)
    S128 = S.to(torch.complex128)
    err = float(((S128 - rec).abs() / S128.abs()).max())
    assert err < 1e-5, err
    assert float(G.min()) > -1e-3 and bool(torch.isfinite(G).all())
    spy.release_device_buffers()


@pytest.mark.parametrize("C", [8, 48])
def test_regularize_near_threshold(C):
    """G1: the decision kappa >= cond_max (wilson_sf.py:239-248) with kappa within 0.5 % of the threshold, on both
    sides, must follow np.linalg.cond (LAPACK SVD) - the device estimate comes from an eigenvalue iteration."""
    from syncopy_amd import backend
    rng = np.random.default_rng(C)
    F = 17
    S = np.empty((F, C, C), dtype=np.complex128)
    for f in range(F):
        q, _ = np.linalg.qr(rng.normal(size=(C, C)) + 1j * rng.normal(size=(C, C)))
        lam = np.exp(rng.uniform(np.log(1.0), np.log(2000.0 + 400.0 * f), size=C))
        lam[0], lam[-1] = 1.0, 2000.0 + 400.0 * f          # condition number grows with f; two bins nearly tie
        S[f] = (q * lam) @ q.conj().T
    S = (0.5 * (S + S.conj().transpose(0, 2, 1))).astype(np.complex64)
    kappa = np.linalg.cond(S).max()
    dev = torch.from_numpy(S).cuda()
    for cmax in (kappa * 1.005, kappa * 0.995, kappa * 0.5):
        _, factor, cn0 = O.regularize_csd(S, cond_max=cmax, eps_max=1e-1)
        _, meta = backend.granger(dev, niter=2, cond_max=cmax)
        np.testing.assert_allclose(meta["initial cond. num"], cn0, rtol=1e-4)
        assert meta["reg. factor"] == pytest.approx(factor, rel=1e-9), (cmax, kappa)


def test_ppc_of_time_resolved_spectra():
    """method='ppc' on SpectralData from mtmconvol (VERDICT r1 missing 5): the reference's pair loop works sample by
    sample on equally long trials (connectivity_analysis.py:624-663); K7 runs once per time sample."""
    from oracle_routines import ORACLE_CONN, ORACLE_FREQ
    adj = np.zeros((5, 5))
    adj[0, 1] = adj[3, 2] = 0.35
    data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=600, nTrials=9, seed=11, samplerate=300)
    kw = dict(method="mtmconvol", t_ftimwin=0.4, toi=np.linspace(-0.5, 0.5, 6), foilim=[20, 80], tapsmofrq=6,
              output="fourier", keeptapers=True)
    spec = spy.freqanalysis(data, **kw)
    assert spec.data.shape[0] == 9 * 6
    res = spy.connectivityanalysis(spec, method="ppc")
    seq = spy.connectivityanalysis(spec, method="ppc", compute_method="sequential", routine_classes=ORACLE_CONN)
    T, L = 9, 6
    csd = np.einsum("tkfi,tkfj->tfij", np.asarray(spec.data), np.conj(np.asarray(spec.data))) / spec.data.shape[1]
    csd = csd.reshape(T, L, *csd.shape[1:]).astype(np.complex64)
    expect = np.concatenate([O.ppc(csd[:, ti]) for ti in range(L)], axis=0)
    assert res.data.shape == expect.shape == (L,) + csd.shape[2:]
    assert_parity(seq.data, expect, atol_rel=5e-6, what="sequential")
    assert_parity(res.data, expect, atol_rel=5e-6, what="hip")
    assert np.array_equal(res.trialdefinition, seq.trialdefinition) and res.trialdefinition.shape == (1, 3)
    # ragged trials are refused the way the reference's accumulator refuses them
    rag = spy.freqanalysis(data, select={"trials": [0, 1]}, **kw)
    rag.trialdefinition = np.array([[0, 5, 0], [5, 12, 0]])
    with pytest.raises((spy.shared.errors.SPYValueError, NotImplementedError)):
        spy.connectivityanalysis(rag, method="ppc")                 # (the reference: "Averaging trials of unequal
                                                                    # lengths in output currently not supported!")


@pytest.mark.parametrize("nsamples", [6000, 7001, 12000])
def test_corr_of_long_trials(nsamples):
    """method='corr' beyond 5461 samples per trial (VERDICT r1 missing 5): the lags come from the FORWARD real
    transform of the even / odd parts of the accumulated cross spectra (16384 points in LDS; 32768 on the four-step
    path for 12000 samples) - against the product's own per-trial path bound to the oracle (fftconvolve per channel
    pair and trial, ST_compRoutines.py:466-584).  Same floor as the short-trial test."""
    from oracle_routines import ORACLE_CONN
    adj = np.zeros((4, 4))
    adj[0, 1] = 0.3
    data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=nsamples, nTrials=3, seed=21, samplerate=1000)
    data.data[:, 2] += 2.0                                          # an offset channel: polyremoval matters
    data.invalidate()
    for kw in ({}, {"polyremoval": 1}, {"keeptrials": True}):
        ref = spy.connectivityanalysis(data, method="corr", compute_method="sequential", routine_classes=ORACLE_CONN, **kw)
        got = spy.connectivityanalysis(data, method="corr", **kw)
        assert got.data.shape == ref.data.shape and got.data.shape[0] == (3 if kw.get("keeptrials") else 1) * ((nsamples + 1) // 2)
        assert_parity(got.data, ref.data, atol_rel=1e-5, what=f"corr {nsamples} {kw}")
