"""ctypes front end of the CPU emulation of the HIP kernels (TEST INFRASTRUCTURE ONLY).

Mirrors the plan construction of syncopy_amd/csrc/mtmfft.hip in NumPy so that a
test can run one kernel configuration on host arrays and compare it with the
oracle.  Never imported by the package.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_emu  # noqa: E402

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_emu.build())
    return _lib


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(C.POINTER(typ))


def twiddles(n):
    m = np.arange(n, dtype=np.float64)
    ang = -2.0 * np.pi * m / n
    return np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32).copy()


def factorize(n):
    rad = []
    for c in (16, 8, 4, 2, 3, 5, 7, 11, 13):
        while n % c == 0 and n > 1:
            rad.append(c)
            n //= c
    return rad, n == 1


OUT_KINDS = {"pow": 0, "abs": 1, "fourier": 2, "complex": 2, "real": 3, "imag": 4, "angle": 5,
             "absreal": 6, "absimag": 7}


def fft_exec(data, seg_start, seg_lo, seg_hi, nsig, nfft, tapers, scale, detrend=-1, demean_taper=False,
             freq_idx=None, output="pow", keeptapers=True, chan_idx=None, G=None, force_generic=False, blocked=False,
             force_long=False, reference_mean=False, no_mixed=False, mixed_nostage=False, dec=None):
    """`reference_mean`: constant detrending with the means of seq_mean_kernel (spyhip_fft_plan_set_reference_mean).
    `dec`: id of a compile-time schedule of mtmfft_dec_kernel (emu_kernels.cpp)."""
    if reference_mean and detrend == 0:
        d32 = np.ascontiguousarray(data, dtype=np.float32)
        nch = d32.shape[1] if chan_idx is None else len(chan_idx)
        means = seq_mean(d32, seg_start, seg_lo, seg_hi, nsig, chan_idx)
        assert means.shape == (len(seg_start), nch)
        lib().emu_set_means(_p(means, C.c_float))
        try:
            return fft_exec(d32, seg_start, seg_lo, seg_hi, nsig, nfft, tapers, scale, detrend, demean_taper, freq_idx,
                            output, keeptapers, chan_idx, G, force_generic, blocked, force_long, False, no_mixed,
                            mixed_nostage, dec)
        finally:
            lib().emu_set_means(None)
    return _fft_exec(data, seg_start, seg_lo, seg_hi, nsig, nfft, tapers, scale, detrend, demean_taper, freq_idx,
                     output, keeptapers, chan_idx, G, force_generic, blocked, force_long, no_mixed, mixed_nostage, dec)


def seq_mean(data, seg_start, seg_lo, seg_hi, nsig, chan_idx=None):
    """Emulated seq_mean_kernel: (nseg, nchan) float32 means in the reference's summation order."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    ci = None if chan_idx is None else np.ascontiguousarray(chan_idx, dtype=np.int32)
    nchan = data.shape[1] if ci is None else len(ci)
    ss = np.ascontiguousarray(seg_start, dtype=np.int64)
    sl = np.ascontiguousarray(seg_lo, dtype=np.int64)
    sh = np.ascontiguousarray(seg_hi, dtype=np.int64)
    means = np.full((len(ss), nchan), np.nan, dtype=np.float32)
    lib().emu_seq_mean(_p(data, C.c_float), C.c_longlong(data.shape[1]), _p(ci, C.c_int), _p(ss, C.c_longlong),
                       _p(sl, C.c_longlong), _p(sh, C.c_longlong), C.c_int(len(ss)), C.c_int(nsig), C.c_int(nchan),
                       _p(means, C.c_float))
    return means


def twiddles64(n):
    ang = -2.0 * np.pi * np.arange(n, dtype=np.float64) / n
    return np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=1))


def fft_exec_f64(data, seg_start, seg_lo, seg_hi, nsig, nfft, tapers, scale, detrend=-1, demean_taper=False,
                 freq_idx=None, output="pow", keeptapers=True, chan_idx=None, reference_mean=False, seg_f64=False,
                 dec=True, bluestein=False, emu_id=None):
    """Emulated spyhip_fft_exec of a plan with spyhip_fft_plan_set_precision(plan, 1): `dec` - the compile-time
    schedule of mtmfft_dec64_kernel for this nfft; otherwise the any-length kernel, `bluestein` in its chirp-z form
    (tables as spyhip_fft_plan_set_precision builds them).  dec="half": the HALF form (single channels through the
    schedule of nfft / 2); `emu_id` picks a schedule variant of the emulator that shares its nfft with another (2002)."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    ld = data.shape[1]
    nchan = ld if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else np.ascontiguousarray(chan_idx, dtype=np.int32)
    ss = np.ascontiguousarray(seg_start, dtype=np.int64)
    sl = np.ascontiguousarray(seg_lo, dtype=np.int64)
    sh = np.ascontiguousarray(seg_hi, dtype=np.int64)
    nseg, K = len(ss), tapers.shape[0]
    tp = np.ascontiguousarray(tapers, dtype=np.float64)
    nf = nfft // 2 + 1
    if freq_idx is None:
        fpos, nfsel = None, nf
    else:
        fi = np.asarray(freq_idx, dtype=np.int64)
        nfsel = len(fi)
        fpos = np.full(nf, -1, dtype=np.int32)
        fpos[fi] = np.arange(nfsel, dtype=np.int32)
    kind = OUT_KINDS[output]
    out = np.full((nseg, K if keeptapers else 1, nfsel, nchan), np.nan, dtype=np.complex64 if kind == 2 else np.float32)
    M, chirp, bhat = 0, None, None
    if bluestein:
        M = 16
        while M < 2 * nfft - 1:
            M *= 2
        n = np.arange(nfft, dtype=np.int64)
        ang = np.pi * ((n * n) % (2 * nfft)).astype(np.float64) / nfft
        chirp = np.ascontiguousarray(np.stack([np.cos(ang), -np.sin(ang)], axis=1))
        b = np.zeros(M, dtype=np.complex128)
        b[:nfft] = np.cos(ang) + 1j * np.sin(ang)
        b[M - n[1:]] = b[n[1:]]
        B = np.fft.fft(b) / M
        bhat = np.ascontiguousarray(np.stack([B.real, B.imag], axis=1))
    means = None
    if reference_mean and detrend == 0:
        means = seq_mean(data, ss, sl, sh, nsig, chan_idx)
        lib().emu_set_means(_p(means, C.c_float))
    try:
        rc = lib().emu_mtmfft_f64(
            C.c_int(nfft if emu_id is None else emu_id), C.c_int(M), _p(data, C.c_float), C.c_longlong(ld), _p(ci, C.c_int), _p(ss, C.c_longlong),
            _p(sl, C.c_longlong), _p(sh, C.c_longlong), C.c_int(nseg), C.c_int(nsig), C.c_int(nchan), C.c_int(K),
            _p(tp, C.c_double), _p(twiddles64(M if bluestein else nfft), C.c_double), _p(chirp, C.c_double),
            _p(bhat, C.c_double), C.c_float(scale), C.c_int(detrend), C.c_int(int(demean_taper)), C.c_int(int(seg_f64)),
            _p(fpos, C.c_int), C.c_int(nfsel), C.c_int(kind), C.c_int(int(keeptapers)),
            C.c_int(2 if (dec == "half") else int(bool(dec) and not bluestein)),
            out.ctypes.data_as(C.c_void_p))
    finally:
        lib().emu_set_means(None)
    assert rc == 0, f"no emulated reference-precision kernel for nfft={nfft} (rc={rc})"
    return out


LAST_MIXED = {}


def _fft_exec(data, seg_start, seg_lo, seg_hi, nsig, nfft, tapers, scale, detrend=-1, demean_taper=False,
              freq_idx=None, output="pow", keeptapers=True, chan_idx=None, G=None, force_generic=False, blocked=False,
              force_long=False, no_mixed=False, mixed_nostage=False, dec=None):
    """Emulated spyhip_fft_exec.  data: (rows, ld) float32; tapers: (K, nsig) float64.
    blocked: channel-blocked hand-over layout (nseg*K, ceil(nchan/4), nfsel, 4) (fourier, keeptapers)."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    ld = data.shape[1]
    nchan = ld if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else np.ascontiguousarray(chan_idx, dtype=np.int32)
    ss = np.ascontiguousarray(seg_start, dtype=np.int64)
    sl = np.ascontiguousarray(seg_lo, dtype=np.int64)
    sh = np.ascontiguousarray(seg_hi, dtype=np.int64)
    nseg = len(ss)
    K = tapers.shape[0]
    tp = np.ascontiguousarray(tapers, dtype=np.float32)
    nf = nfft // 2 + 1
    if freq_idx is None:
        fpos, nfsel = None, nf
    else:
        fi = np.asarray(freq_idx, dtype=np.int64)
        nfsel = len(fi)
        fpos = np.full(nf, -1, dtype=np.int32)
        fpos[fi] = np.arange(nfsel, dtype=np.int32)
    kind = OUT_KINDS[output]
    kout = K if keeptapers else 1
    out = np.full((nseg, kout, nfsel, nchan), np.nan, dtype=np.complex64 if kind == 2 else np.float32)
    if blocked:
        assert kind == 2 and keeptapers
        out = np.full((nseg * kout, (nchan + 3) // 4, nfsel, 4), np.nan, dtype=np.complex64)
    if dec is not None:
        # mtmfft_dec_kernel: the compile-time schedule `dec` (emu_kernels.cpp: emu_mtmfft_dec); dec < 0: the HALF form of
        # nfft = -dec (channel pairs through the schedule of nfft / 2: twiddles of nfft / 2 + the half-step table)
        half = int(dec) < 0
        twh = np.ascontiguousarray(twiddles(nfft)[: nfft // 4 + 1]) if half else None
        lib().emu_set_twh(_p(twh, C.c_float))
        rc = lib().emu_mtmfft_dec(
            C.c_int(int(dec)), _p(data, C.c_float), C.c_longlong(ld), _p(ci, C.c_int),
            _p(ss, C.c_longlong), _p(sl, C.c_longlong), _p(sh, C.c_longlong), C.c_int(nseg),
            C.c_int(nsig), C.c_int(nchan), C.c_int(K), _p(tp, C.c_float), _p(twiddles(nfft // 2 if half else nfft), C.c_float),
            C.c_float(scale), C.c_int(detrend), C.c_int(int(demean_taper)), _p(fpos, C.c_int),
            C.c_int(nfsel), C.c_int(kind), C.c_int(int(keeptapers)), out.ctypes.data_as(C.c_void_p))
        lib().emu_set_twh(None)
        assert rc == 0, f"no emulated decimal kernel {dec}"
        return out
    pow2 = (nfft & (nfft - 1)) == 0 and 256 <= nfft <= 16384 and not force_generic
    if pow2:
        log2n = int(np.log2(nfft))
        if G is None:
            G = {8: 16, 9: 8, 10: 4, 11: 2, 12: 1, 13: 1, 14: 1}[log2n]
            if log2n == 12 and kind == 2 and keeptapers:
                G = 2               # as spyhip_fft_plan_create: store-bound complex spectra take two quads per workgroup
        # 2^14 (as spyhip_fft_plan_create): channel pairs through the 8192-point engine, HALF form of mtmfft_quad_kernel
        tw = twiddles(nfft // 2 if log2n == 14 else nfft)
        twh = np.ascontiguousarray(twiddles(nfft)[: nfft // 4 + 1]) if log2n == 14 else None
        lib().emu_set_twh(_p(twh, C.c_float))
        set_blocked(blocked)
        rc = lib().emu_mtmfft_pow2(
            C.c_int(log2n), C.c_int(G), _p(data, C.c_float), C.c_longlong(ld), _p(ci, C.c_int),
            _p(ss, C.c_longlong), _p(sl, C.c_longlong), _p(sh, C.c_longlong), C.c_int(nseg),
            C.c_int(nsig), C.c_int(nchan), C.c_int(K), _p(tp, C.c_float), _p(tw, C.c_float),
            C.c_float(scale), C.c_int(detrend), C.c_int(int(demean_taper)), _p(fpos, C.c_int),
            C.c_int(nfsel), C.c_int(kind), C.c_int(int(keeptapers)), out.ctypes.data_as(C.c_void_p))
        set_blocked(False)
        lib().emu_set_twh(None)
        assert rc == 0, f"no emulated kernel for log2n={log2n} G={G}"
        return out
    if not force_generic and not force_long and not no_mixed:
        # 5-smooth lengths: the packed mixed-radix engine with the schedule of spyhip_fft_plan_create
        info = np.zeros(7, dtype=np.int32)
        rc = lib().emu_mtmfft_mixed(
            C.c_int(nfft), C.c_int(int(mixed_nostage)), _p(info, C.c_int), _p(data, C.c_float), C.c_longlong(ld),
            _p(ci, C.c_int), _p(ss, C.c_longlong), _p(sl, C.c_longlong), _p(sh, C.c_longlong), C.c_int(nseg),
            C.c_int(nsig), C.c_int(nchan), C.c_int(K), _p(tp, C.c_float), _p(twiddles(nfft), C.c_float),
            C.c_float(scale), C.c_int(detrend), C.c_int(int(demean_taper)), _p(fpos, C.c_int), C.c_int(nfsel),
            C.c_int(kind), C.c_int(int(keeptapers)), out.ctypes.data_as(C.c_void_p))
        if rc == 0:
            LAST_MIXED.update(dict(zip(("th", "G", "npass", "stage", "threads", "radix0", "radix_last"),
                                       (int(v) for v in info))), nfft=nfft)
            return out
    if force_long or (not pow2 and nfft > 4096 and not force_generic):
        # Bluestein with four-step length-M transforms through (emulated) HBM - mirror of spyhip_fft_plan_create
        m = 12
        while (1 << m) < 2 * nfft - 1:
            m += 1
        M = 1 << m
        l1, l2 = (m + 1) // 2, m // 2
        M1, M2 = 1 << l1, 1 << l2
        k = np.arange(nfft, dtype=np.int64)
        ang = np.pi * ((k * k) % (2 * nfft)) / nfft
        chirp = np.stack([np.cos(ang), -np.sin(ang)], axis=1).astype(np.float32).copy()
        bq = np.zeros(M, dtype=np.complex128)
        bq[:nfft] = np.exp(1j * ang)
        bq[M - nfft + 1:] = bq[1:nfft][::-1]
        bh = (np.fft.fft(bq) / M).reshape(M2, M1).T                 # [k1][k2] with k = k1 + M1*k2
        bhat = np.stack([bh.real, bh.imag], axis=-1).astype(np.float32).copy()
        tp64 = tp.astype(np.float64)
        mid = 0.5 * (nsig - 1)
        wsum = np.stack([tp64.sum(axis=1), (tp64 * (np.arange(nsig) - mid)).sum(axis=1)], axis=1).copy()
        rc = lib().emu_mtmfft_long(
            C.c_int(l1), C.c_int(l2), C.c_int(nfft), _p(chirp, C.c_float), _p(bhat, C.c_float),
            _p(twiddles(M1), C.c_float), _p(twiddles(M2), C.c_float), _p(twiddles(M), C.c_float),
            wsum.ctypes.data_as(C.POINTER(C.c_double)), _p(data, C.c_float), C.c_longlong(ld), _p(ci, C.c_int),
            _p(ss, C.c_longlong), _p(sl, C.c_longlong), _p(sh, C.c_longlong), C.c_int(nseg), C.c_int(nsig),
            C.c_int(nchan), C.c_int(K), _p(tp, C.c_float), C.c_float(scale), C.c_int(detrend),
            C.c_int(int(demean_taper)), _p(fpos, C.c_int), C.c_int(nfsel), C.c_int(kind), C.c_int(int(keeptapers)),
            out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return out
    if 2 * nfft - 1 <= 8192 and nfft >= 2 and not force_generic:
        # Bluestein on the packed power-of-two engine (mirror of spyhip_fft_plan_create)
        M = 256
        while M < 2 * nfft - 1:
            M *= 2
        k = np.arange(nfft, dtype=np.int64)
        ang = np.pi * ((k * k) % (2 * nfft)) / nfft
        chirp = np.stack([np.cos(ang), -np.sin(ang)], axis=1).astype(np.float32).copy()
        bq = np.zeros(M, dtype=np.complex128)
        bq[:nfft] = np.exp(1j * ang)
        bq[M - nfft + 1:] = bq[1:nfft][::-1]
        bh = np.fft.fft(bq) / M
        bhat = np.stack([bh.real, bh.imag], axis=1).astype(np.float32).copy()
        log2m = int(np.log2(M))
        Gb = {8: 16, 9: 8, 10: 4, 11: 2, 12: 1, 13: 1}[log2m]
        tw = twiddles(M)
        rc = lib().emu_mtmfft_blue(
            C.c_int(log2m), C.c_int(Gb), C.c_int(nfft), _p(chirp, C.c_float), _p(bhat, C.c_float),
            _p(data, C.c_float), C.c_longlong(ld), _p(ci, C.c_int), _p(ss, C.c_longlong), _p(sl, C.c_longlong),
            _p(sh, C.c_longlong), C.c_int(nseg), C.c_int(nsig), C.c_int(nchan), C.c_int(K), _p(tp, C.c_float),
            _p(tw, C.c_float), C.c_float(scale), C.c_int(detrend), C.c_int(int(demean_taper)), _p(fpos, C.c_int),
            C.c_int(nfsel), C.c_int(kind), C.c_int(int(keeptapers)), out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return out
    rad, ok = factorize(nfft)
    n, blu, chirp, bhat = nfft, 0, None, None
    if not ok:
        M = 16
        while M < 2 * nfft - 1:
            M *= 2
        k = np.arange(nfft, dtype=np.int64)
        ang = np.pi * ((k * k) % (2 * nfft)) / nfft
        chirp = np.stack([np.cos(ang), -np.sin(ang)], axis=1).astype(np.float32).copy()
        b = np.zeros(M, dtype=np.complex128)
        b[:nfft] = np.exp(1j * ang)
        b[M - nfft + 1:] = b[1:nfft][::-1]
        bh = np.fft.fft(b) / M
        bhat = np.stack([bh.real, bh.imag], axis=1).astype(np.float32).copy()
        n, blu = M, 1
        rad, _ = factorize(M)
    tw = twiddles(n)
    radix = np.asarray(rad, dtype=np.int32)
    stage_x = 1 if (2 * n + nsig) * 8 <= 160 * 1024 else 0
    lib().emu_mtmfft_generic(
        C.c_int(n), C.c_int(len(rad)), _p(radix, C.c_int), C.c_int(nfft), C.c_int(blu), _p(chirp, C.c_float),
        _p(bhat, C.c_float), C.c_int(stage_x), _p(data, C.c_float), C.c_longlong(ld), _p(ci, C.c_int),
        _p(ss, C.c_longlong), _p(sl, C.c_longlong), _p(sh, C.c_longlong), C.c_int(nseg), C.c_int(nsig),
        C.c_int(nchan), C.c_int(K), _p(tp, C.c_float), _p(tw, C.c_float), C.c_float(scale), C.c_int(detrend),
        C.c_int(int(demean_taper)), _p(fpos, C.c_int), C.c_int(nfsel), C.c_int(kind), C.c_int(int(keeptapers)),
        out.ctypes.data_as(C.c_void_p))
    return out


def fft_exec_declong(data, seg_start, seg_lo, seg_hi, nsig, P, M, tapers, scale, detrend=-1, demean_taper=False, freq_idx=None,
                     output="pow", keeptapers=True, chan_idx=None, reference_mean=False):
    """Emulated spyhip_fft_exec of a K1L2 plan (mtmfft_declong.h): nfft = P M, tables as spyhip_fft_plan_create builds them."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    nfft = P * M
    ld = data.shape[1]
    nchan = ld if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else np.ascontiguousarray(chan_idx, dtype=np.int32)
    ss = np.ascontiguousarray(seg_start, dtype=np.int64)
    sl = np.ascontiguousarray(seg_lo, dtype=np.int64)
    sh = np.ascontiguousarray(seg_hi, dtype=np.int64)
    nseg, K = len(ss), tapers.shape[0]
    tp = np.ascontiguousarray(tapers, dtype=np.float32)
    nf = nfft // 2 + 1
    if freq_idx is None:
        fpos, nfsel = None, nf
    else:
        fi = np.asarray(freq_idx, dtype=np.int64)
        nfsel = len(fi)
        fpos = np.full(nf, -1, dtype=np.int32)
        fpos[fi] = np.arange(nfsel, dtype=np.int32)
    kind = OUT_KINDS[output]
    out = np.full((nseg, K if keeptapers else 1, nfsel, nchan), np.nan, dtype=np.complex64 if kind == 2 else np.float32)
    mid = 0.5 * (nsig - 1)
    t64 = np.asarray(tapers, dtype=np.float64)
    wsum = np.ascontiguousarray(np.stack([t64.sum(axis=1), (t64 * (np.arange(nsig) - mid)).sum(axis=1)], axis=1))
    means = None
    if reference_mean and detrend == 0:
        means = seq_mean(data, ss, sl, sh, nsig, chan_idx)
    lib().emu_set_means(_p(means, C.c_float) if means is not None else None)
    tws, twn, twp = twiddles(M), twiddles(nfft), twiddles(P)
    rc = lib().emu_mtmfft_declong(
        C.c_int(P), C.c_int(M), _p(tws, C.c_float), _p(twn, C.c_float), _p(twp, C.c_float), _p(wsum, C.c_double),
        _p(data, C.c_float), C.c_longlong(ld), _p(ci, C.c_int), _p(ss, C.c_longlong), _p(sl, C.c_longlong),
        _p(sh, C.c_longlong), C.c_int(nseg), C.c_int(nsig), C.c_int(nchan), C.c_int(K), _p(tp, C.c_float),
        C.c_float(scale), C.c_int(detrend), C.c_int(int(demean_taper)), _p(fpos, C.c_int), C.c_int(nfsel), C.c_int(kind),
        C.c_int(int(keeptapers)), out.ctypes.data_as(C.c_void_p))
    lib().emu_set_means(None)
    if rc != 0:
        raise RuntimeError(f"emu_mtmfft_declong({P}, {M}) -> {rc}")
    return out


def set_blocked(on):
    """Channel-blocked hand-over layout for the following fft_exec / csd_accumulate calls."""
    lib().emu_set_blocked(C.c_int(int(bool(on))))


def csd_accumulate(spec, acc, force_tpw=0, blocked=False, force_4m=False):
    """Emulated spyhip_csd_accumulate: spec (R, F, C) complex64 - or (R, ceil(C/4), F, 4) with blocked=True -,
    acc (F, C, C) complex64 (in place).  `force_4m`: 256 channels on the 4-multiplication kernel (SPYHIP_CSD_4M).
    Returns a code for the kernel that ran (8 = the 3-multiplication kernel)."""
    if force_4m:
        lib().emu_set_force_4m(1)
        try:
            return csd_accumulate(spec, acc, force_tpw, blocked)
        finally:
            lib().emu_set_force_4m(0)
    spec = np.ascontiguousarray(spec, dtype=np.complex64)
    assert acc.dtype == np.complex64 and acc.flags.c_contiguous
    if blocked:
        R, F, Cn = spec.shape[0], acc.shape[0], acc.shape[1]
        assert spec.shape == (R, (Cn + 3) // 4, F, 4)
        set_blocked(True)
        try:
            return lib().emu_csd_accumulate(spec.ctypes.data_as(C.c_void_p), C.c_longlong(R), C.c_int(F), C.c_int(Cn),
                                            acc.ctypes.data_as(C.c_void_p), C.c_int(force_tpw))
        finally:
            set_blocked(False)
    R, F, Cn = spec.shape
    return lib().emu_csd_accumulate(spec.ctypes.data_as(C.c_void_p), C.c_longlong(R), C.c_int(F), C.c_int(Cn),
                                    acc.ctypes.data_as(C.c_void_p), C.c_int(force_tpw))


def csd_finalize(acc, scale):
    F, Cn, _ = acc.shape
    lib().emu_csd_finalize(acc.ctypes.data_as(C.c_void_p), C.c_int(F), C.c_int(Cn), C.c_float(scale))


def jack_coh_accumulate(spec, ntaper, S, direct, output, ntrials_total, sum_d, sum_d2):
    """Emulated spyhip_jack_coh_accumulate (sum_d / sum_d2 float64 in place; sum_d complex128 for 'complex')."""
    spec = np.ascontiguousarray(spec, dtype=np.complex64)
    R, F, Cn = spec.shape
    S = np.ascontiguousarray(S, dtype=np.complex64)
    direct = np.ascontiguousarray(direct)
    lib().emu_jack_coh(spec.ctypes.data_as(C.c_void_p), C.c_int(R // ntaper), C.c_int(ntaper), C.c_int(F), C.c_int(Cn),
                       S.ctypes.data_as(C.c_void_p), direct.ctypes.data_as(C.c_void_p), C.c_int(OUT_KINDS[output]),
                       C.c_longlong(ntrials_total), sum_d.ctypes.data_as(C.c_void_p), sum_d2.ctypes.data_as(C.c_void_p))


def ppc_accumulate(spec, ntaper, acc):
    """Emulated spyhip_ppc_accumulate: spec (T * ntaper, F, C) complex64, acc (F, C, C) complex64 in place."""
    spec = np.ascontiguousarray(spec, dtype=np.complex64)
    R, F, Cn = spec.shape
    lib().emu_ppc_accumulate(spec.ctypes.data_as(C.c_void_p), C.c_int(R // ntaper), C.c_int(ntaper), C.c_int(F),
                             C.c_int(Cn), acc.ctypes.data_as(C.c_void_p))


def ppc_accumulate_csd(csd, acc):
    csd = np.ascontiguousarray(csd, dtype=np.complex64)
    lib().emu_ppc_accumulate_csd(csd.ctypes.data_as(C.c_void_p), C.c_int(csd.shape[0]), C.c_longlong(acc.size),
                                 acc.ctypes.data_as(C.c_void_p))


def ppc_finalize(acc, ntrials, lower_only):
    F, ni, nj = acc.shape
    out = np.empty((F, ni, nj), dtype=np.float32)
    lib().emu_ppc_finalize(acc.ctypes.data_as(C.c_void_p), C.c_int(F), C.c_int(ni), C.c_int(nj),
                           C.c_int(int(lower_only)), C.c_longlong(ntrials), out.ctypes.data_as(C.c_void_p))
    return out


def ccov_from_accumulator(acc, nsamples, scale, norm=0):
    """Emulated spyhip_ccov_from_accumulator: acc (nfft/2+1, C, C) complex64 -> (nlag, C, C) float32."""
    acc = np.ascontiguousarray(acc, dtype=np.complex64)
    F, Cn, _ = acc.shape
    L = 2 * (F - 1)
    tw = np.exp(-2j * np.pi * np.arange(L) / L).astype(np.complex64)
    nlag = nsamples // 2 + (nsamples & 1)
    out = np.zeros((nlag, Cn, Cn), dtype=np.float32)
    rc = lib().emu_ccov(acc.ctypes.data_as(C.c_void_p), tw.ctypes.data_as(C.c_void_p), C.c_int(L), C.c_int(Cn),
                        C.c_int(nsamples), C.c_double(scale), C.c_int(norm), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def coh_from_accumulator(acc, scale, output="abs"):
    """Emulated spyhip_coh_from_accumulator (raw lower-triangle accumulator -> coherence)."""
    F, Cn, _ = acc.shape
    kind = OUT_KINDS[output]
    out = np.zeros((F, Cn, Cn), dtype=np.complex64 if kind == 2 else np.float32)
    lib().emu_coh_from_accumulator(acc.ctypes.data_as(C.c_void_p), C.c_int(F), C.c_int(Cn), C.c_float(scale),
                                   C.c_int(kind), out.ctypes.data_as(C.c_void_p))
    return out


def coh_normalize(csd, output="abs"):
    F, Cn, _ = csd.shape
    kind = OUT_KINDS[output]
    out = np.empty((F, Cn, Cn), dtype=np.complex64 if kind == 2 else np.float32)
    lib().emu_coh_normalize(csd.ctypes.data_as(C.c_void_p), C.c_int(F), C.c_int(Cn), C.c_int(kind),
                            out.ctypes.data_as(C.c_void_p))
    return out


def cwt_plan_tables(nsig, scales, dt, w0, nbmin=1024):
    """NumPy mirror of the plan construction in syncopy_amd/csrc/cwt.hip."""
    kers, halo, right = [], 0, 0
    for sc in scales:
        M = 10.0 * sc / dt
        t0, t1 = (-M + 1.0) / 2.0, (M + 1.0) / 2.0
        L = max(int(np.ceil(t1 - t0)), 1)
        c = (L - 1) // 2
        m0, m1 = max(0, c - (nsig - 1)), min(L, c + nsig)
        m = np.arange(m0, m1, dtype=np.float64)
        x = (t0 + m) * dt / sc
        norm = np.sqrt(dt) / (sc * 8.0 * np.pi) * np.pi ** (-0.25)
        h = norm * np.exp(-0.5 * x * x) * (np.exp(1j * w0 * x) - np.exp(-0.5 * w0 * w0))
        kers.append((h, c - m0))
        halo = max(halo, h.size - 1 - (c - m0))
        right = max(right, c - m0)
    NB = nbmin
    while NB < 2 * (halo + right + 1) and NB < 16384:
        NB *= 2
    V = NB - halo - right
    assert V >= 1
    hs = np.zeros((len(scales), NB), dtype=np.complex128)
    cshift = np.zeros(len(scales), dtype=np.int32)
    for s, (h, c) in enumerate(kers):
        hs[s, :h.size] = h
        cshift[s] = halo + c
    hs = np.fft.fft(hs, axis=1) / NB
    hspec = np.stack([hs.real, hs.imag], axis=-1).astype(np.float32).copy()
    return NB, V, halo, cshift, hspec


def cwt_exec(data, seg_start, trial_lo, trial_hi, nsig, scales, dt, w0=6.0, detrend=-1, output="pow", tpos=None,
             ntime_out=None, chan_idx=None, accumulate=0, mode=0, nbmin=1024):
    """`mode` as cwt.hip combines the kernel variants: bit 0 trial sums on pairs of segments (accumulate=2), bit 1 the direct
    kernels (1024- / 2048-point blocks, accumulate 0 / 1), bit 2 the channel-major input copy; `nbmin`: the shortest
    block (build_groups)."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    ld = data.shape[1]
    nchan = ld if chan_idx is None else len(chan_idx)
    ci = None if chan_idx is None else np.ascontiguousarray(chan_idx, dtype=np.int32)
    ss, tl, th = (np.ascontiguousarray(a, dtype=np.int64) for a in (seg_start, trial_lo, trial_hi))
    NB, V, halo, cshift, hspec = cwt_plan_tables(nsig, scales, dt, w0, nbmin)
    log2n = int(np.log2(NB))
    G = {10: 4, 11: 2}.get(log2n, 1)      # channel pairs per workgroup (packed kernel), 2^14: channels
    if mode & 2:
        G = {10: 8, 11: 4}[log2n]         # the direct kernels' workgroups
    tw = twiddles(NB)
    kind = OUT_KINDS[output]
    tp = None if tpos is None else np.ascontiguousarray(tpos, dtype=np.int32)
    nto = nsig if tpos is None else int(ntime_out)
    out = np.zeros((1 if accumulate == 2 else len(ss), nto, len(scales), nchan), dtype=np.complex64 if kind == 2 else np.float32)
    rc = lib().emu_cwt(C.c_int(log2n), C.c_int(G), _p(data, C.c_float), C.c_longlong(ld), _p(ci, C.c_int),
                       _p(ss, C.c_longlong), _p(tl, C.c_longlong), _p(th, C.c_longlong), C.c_int(len(ss)),
                       C.c_int(nsig), C.c_int(nchan), C.c_int(len(scales)), _p(tw, C.c_float),
                       hspec.ctypes.data_as(C.POINTER(C.c_float)), _p(cshift, C.c_int), C.c_int(V), C.c_int(halo),
                       C.c_int((nsig + V - 1) // V), C.c_int(detrend), C.c_int(kind), _p(tp, C.c_int), C.c_int(nto),
                       out.ctypes.data_as(C.c_void_p), C.c_int(accumulate), C.c_int(mode))
    assert rc == 0
    return out


# ---------------------------------------------------------------- Wilson / Granger (mirror of the host loop in granger.hip)
def _dp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def w_gemm(A, B, opB=0, addI=0):
    A = np.ascontiguousarray(A, dtype=np.complex128)
    B = np.ascontiguousarray(B, dtype=np.complex128)
    batch, n = (A.shape[0], A.shape[1]) if A.ndim == 3 else (1, A.shape[0])
    sB = 0 if B.ndim == 2 and A.ndim == 3 else n * n
    out = np.zeros_like(A)
    lib().emu_w_gemm(_dp(A), _dp(B), _dp(out), C.c_int(n), C.c_int(batch), C.c_longlong(n * n), C.c_longlong(sB),
                     C.c_longlong(n * n), C.c_int(opB), C.c_int(addI))
    return out


def w_gemm_fused(A, B, opB=0, badd=None, ref=None):
    """zgemm_mfma_kernel's fused forms (n >= 48): A op(B + ...) with `badd` (n, n) joined to op(B); with `ref` the
    return value is max |ref - A op(B)| / |ref| and no product is stored."""
    A = np.ascontiguousarray(A, dtype=np.complex128)
    B = np.ascontiguousarray(B, dtype=np.complex128)
    batch, n = A.shape[0], A.shape[1]
    out = np.zeros_like(A)
    ba = None if badd is None else np.ascontiguousarray(badd, dtype=np.complex128)
    rf = None if ref is None else np.ascontiguousarray(ref, dtype=np.complex128)
    lib().emu_w_gemm_fused.restype = C.c_double
    err = lib().emu_w_gemm_fused(_dp(A), _dp(B), _dp(out), C.c_int(n), C.c_int(batch), C.c_longlong(n * n), C.c_int(opB),
                                 _dp(ba), _dp(rf))
    return err if ref is not None else out


def w_skew(g0):
    g0 = np.ascontiguousarray(g0, dtype=np.complex128)
    n = g0.shape[0]
    S, g0S = np.zeros_like(g0), np.zeros_like(g0)
    lib().emu_w_skew(_dp(g0), _dp(S), _dp(g0S), C.c_int(n))
    return S, g0S


def w_inv(M, blocked=False):
    M = np.array(M, dtype=np.complex128, order="C")
    batch, n = (M.shape[0], M.shape[1]) if M.ndim == 3 else (1, M.shape[0])
    info = np.zeros(batch, dtype=np.int32)
    fn = {False: lib().emu_w_inv, True: lib().emu_w_inv_blocked, "mfma": lib().emu_w_inv_mfma,
          "mfma64": lib().emu_w_inv_mfma64}[blocked]
    fn(_dp(M), C.c_int(n), C.c_int(batch), _dp(info))
    return M, info


def w_plus(g, fast=False):
    """Emulated plus operator on a half spectrum g (F, n, n) complex128 -> (gp (F, n, n), g0 (n, n)).
    `fast`: plus4_kernel (power-of-two lag-domain lengths; two entries per complex transform)."""
    g = np.ascontiguousarray(g, dtype=np.complex128)
    F, n, _ = g.shape
    L = 2 * (F - 1)
    tw = np.ascontiguousarray(np.exp(-2j * np.pi * np.arange(L) / L))
    gp = np.zeros_like(g)
    g0 = np.zeros((n, n), dtype=np.complex128)
    if fast:
        rc = lib().emu_w_plus4(_dp(g), C.c_int(F), C.c_int(n), _dp(tw), _dp(gp), _dp(g0))
        assert rc == 0, f"no plus4 kernel for F = {F}"
        return gp, g0
    lib().emu_w_plus(_dp(g), C.c_int(F), C.c_int(n), _dp(tw), _dp(gp), _dp(g0))     # returns the number of passes
    return gp, g0


def w_chol(M):
    M = np.array(M, dtype=np.complex128, order="C")
    batch, n = (M.shape[0], M.shape[1]) if M.ndim == 3 else (1, M.shape[0])
    info = np.zeros(batch, dtype=np.int32)
    lib().emu_w_chol(_dp(M), C.c_int(n), C.c_int(batch), _dp(info))
    return M, info


def w_cond(A, iters=400):
    F, n, _ = A.shape
    Ai, info = w_inv(A)
    l1, l2 = np.zeros(F), np.zeros(F)
    lib().emu_w_power(_dp(np.ascontiguousarray(A)), C.c_int(n), C.c_int(F), C.c_int(iters), _dp(l1))
    lib().emu_w_power(_dp(Ai), C.c_int(n), C.c_int(F), C.c_int(iters), _dp(l2))
    c = l1 * l2
    c[info != 0] = np.inf
    return float(np.nanmax(np.where(np.isnan(c), np.inf, c)))


def granger(csd64, rtol=5e-6, niter=100, cond_max=1e4, eps_max=1e-1, cond_iters=400):
    """Emulated spyhip_granger: csd64 (F, n, n) complex64 -> (granger float32, H, Sigma, info[4])."""
    csd64 = np.ascontiguousarray(csd64, dtype=np.complex64)
    F, n, _ = csd64.shape
    L = 2 * (F - 1)

    def widen(eps):
        out = np.zeros((F, n, n), dtype=np.complex128)
        lib().emu_w_widen(_dp(csd64), _dp(out), C.c_int(n), C.c_longlong(F * n * n), C.c_double(eps))
        return out
    A = widen(0.0)
    cond0 = w_cond(A, cond_iters)
    factor = 0.0
    if not cond0 < cond_max:
        factor = -1.0
        for s in range(15):
            eps = 10.0 ** (-10.0 + (np.log10(eps_max) + 10.0) * s / 14)
            A = widen(eps)
            if w_cond(A, cond_iters) < cond_max:
                factor = eps
                break
    U, info = w_chol(A)
    assert not info.any()
    g0m = np.zeros((n, n), dtype=np.complex128)
    lib().emu_w_gamma0(_dp(A), C.c_int(F), C.c_int(n), _dp(g0m))
    Lc, info = w_chol(g0m)
    psi0 = np.ascontiguousarray(Lc.T)
    psi = np.ascontiguousarray(np.tile(psi0, (F, 1, 1)))
    m = np.arange(L)
    tw = np.ascontiguousarray(np.exp(-2j * np.pi * m / L))
    converged, err = False, np.inf
    for _ in range(niter):
        pinv, _i = w_inv(psi)
        T2 = w_gemm(pinv, U)
        g = w_gemm(T2, T2, opB=1, addI=1)
        gp = np.zeros_like(g)
        g0 = np.zeros((n, n), dtype=np.complex128)
        lib().emu_w_plus(_dp(g), C.c_int(F), C.c_int(n), _dp(tw), _dp(gp), _dp(g0))
        g0S = np.zeros((n, n), dtype=np.complex128)
        lib().emu_w_addS(_dp(gp), _dp(g0), _dp(g0S), C.c_int(F), C.c_int(n))
        psi = w_gemm(psi, gp)
        psi0 = w_gemm(psi0, g0S)
        rec = w_gemm(psi, psi, opB=1)
        lib().emu_w_relerr.restype = C.c_double
        err = lib().emu_w_relerr(_dp(A), _dp(rec), C.c_longlong(F * n * n))
        if err < rtol:
            converged = True
            break
    Sigma = w_gemm(psi0, psi0, opB=1)
    p0i, _i = w_inv(psi0)
    H = w_gemm(psi, p0i)
    G = np.zeros((F, n, n), dtype=np.float32)
    lib().emu_w_granger(_dp(A), _dp(H), _dp(np.ascontiguousarray(Sigma)), C.c_int(F), C.c_int(n), _dp(G))
    return G, H, Sigma, np.array([float(converged), err, factor, cond0])
