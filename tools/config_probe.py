"""Timing of the BASELINE.json configurations 2, 4, 5 on one GPU (development aid -> gpurun_out/configs.json)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.signal import windows  # noqa: E402

from syncopy_amd import backend as be  # noqa: E402
from syncopy_amd import synthdata  # noqa: E402


def sync_time(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


res = {}
which = sys.argv[1:] or ["c2", "conv", "conv500", "wav", "frontend", "jack", "h2d", "granger", "ppc", "corr", "slt"]

if "c2" in which:
    C, N, T, K = 256, 4096, 1000, 7
    data = synthdata.ar2_uncoupled_fast(C, N, T, seed=1)
    starts = torch.arange(T, device="cuda", dtype=torch.int64) * N
    plan = be.FFTPlan(N, N, C, windows.dpss(N, 4.096, K) * np.sqrt(N), np.sqrt(2) / N, 0, False, None, "pow", False)
    out = torch.empty(plan.out_shape(T), dtype=torch.float32, device="cuda")
    dt = sync_time(lambda: plan.execute(data, starts, out=out))
    byt = T * (N * C * 4 + (N // 2 + 1) * C * 4)
    res["c2_mtmfft_pow"] = {"trials_per_s": T / dt, "us_per_trial": 1e6 * dt / T, "GBps": byt / dt / 1e9,
                            "kernel": plan.kernel_name}
    print("c2", res["c2_mtmfft_pow"], flush=True)
    del data, out

if "ppc" in which:
    # K7 alone on the headline shape: spectra of 100 trials x 7 tapers resident, phasor sums over them
    C, F, T, K = 256, 2049, 100, 7
    g = torch.Generator(device="cuda").manual_seed(5)
    spec = torch.view_as_complex(torch.randn((T * K, F, C, 2), device="cuda", generator=g))
    U = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    dt = sync_time(lambda: be.ppc_accumulate(spec, K, U), n=3)
    pairs = F * (C // 32) * (C // 32 + 1) // 2 * 1024
    res["ppc_accumulate"] = {"trials_per_s": T / dt, "us_per_trial": 1e6 * dt / T,
                             "Gpairs_per_s": pairs * T / dt / 1e9,
                             "GFLOPs": pairs * T * (8 * K + 12) / dt / 1e9}
    print("ppc", res["ppc_accumulate"], flush=True)
    del spec, U

if "corr" in which:
    # K8 on the headline shape: cross-correlation of 256 channels x 4096 samples, 200 trials, 2048 lags
    import syncopy_amd.connectivity.ST_compRoutines as ST
    C, N, T = 256, 4096, 200
    data = synthdata.ar2_uncoupled_fast(C, N, T, seed=6)
    rows = [(t * N, (t + 1) * N) for t in range(T)]
    state = {}

    def accumulate():
        state["acc"], _ = ST._ccov_trials(data, rows, None, 0, 1.0, False)
    dt_acc = sync_time(accumulate, n=2)
    dt_lag = sync_time(lambda: be.ccov_from_accumulator(state["acc"], N, 1.0 / T, 1), n=3)
    res["corr"] = {"accumulate_us_per_trial": 1e6 * dt_acc / T, "lags_ms": 1e3 * dt_lag,
                   "out_GB": 2048 * C * C * 4 / 1e9, "trials_per_s_incl_lags": T / (dt_acc + dt_lag)}
    print("corr", res["corr"], flush=True)
    del data, state

if "slt" in which:
    # superlets on the c4 shape: 128 ch x 16384 samples, 25 frequencies 4..100 Hz, orders 1..5 (c_1 = 3), pow
    from syncopy_amd.specest import compRoutines as CR
    C, N, T = 128, 16384, 8
    data = synthdata.ar2_uncoupled_fast(C, N, T, seed=7)
    foi = np.arange(4, 104, 4, dtype=float)
    mk = {"samplerate": 1000.0, "scales": (1 / foi) / (2 * np.pi), "order_max": 5, "order_min": 1, "c_1": 3,
          "adaptive": False}
    rows = [(t * N, (t + 1) * N) for t in range(T)]
    sl = [slice(None)] * T

    def run():
        return CR._superlet_device(data, rows, sl, sl, None, 0, "pow", mk)
    dt = sync_time(run, n=2)
    res["c4_superlet"] = {"trials_per_s": T / dt, "ms_per_trial": 1e3 * dt / T, "orders": 5}
    print("slt", res["c4_superlet"], flush=True)
    del data

if "conv" in which:
    C, N, T = 128, 16384, 100
    data = synthdata.ar2_uncoupled_fast(C, N, T, seed=2)
    nperseg, step = 512, 256
    w = windows.hann(nperseg)
    w = w * np.sqrt(4 / 3) * np.sqrt(nperseg / w.sum())
    plan = be.FFTPlan(nperseg, nperseg, C, w[None], np.sqrt(2) / nperseg, 0, False, None, "pow", False)
    nT = int(np.ceil(N / step))
    fr = torch.arange(nT, device="cuda", dtype=torch.int64) * step - nperseg // 2
    tr = torch.arange(T, device="cuda", dtype=torch.int64) * N
    starts = (tr[:, None] + fr[None, :]).reshape(-1).contiguous()
    lo = tr[:, None].expand(T, nT).reshape(-1).contiguous()
    hi = (lo + N).contiguous()
    out = torch.empty(plan.out_shape(T * nT), dtype=torch.float32, device="cuda")
    dt = sync_time(lambda: plan.execute(data, starts, lo, hi, out=out))
    byt = T * (N * C * 4 + nT * 257 * C * 4)
    res["c4_mtmconvol"] = {"trials_per_s": T / dt, "us_per_trial": 1e6 * dt / T, "GBps": byt / dt / 1e9,
                           "kernel": plan.kernel_name}
    print("conv", res["c4_mtmconvol"], flush=True)
    del data, out

if "conv500" in which:
    # the same sliding-window analysis with a 500-sample window (0.5 s at 1 kHz): mixed-radix kernel (4*5*5*5)
    C, N, T = 128, 16384, 100
    data = synthdata.ar2_uncoupled_fast(C, N, T, seed=2)
    nperseg, step = 500, 250
    w = windows.hann(nperseg)
    w = w * np.sqrt(4 / 3) * np.sqrt(nperseg / w.sum())
    plan = be.FFTPlan(nperseg, nperseg, C, w[None], np.sqrt(2) / nperseg, 0, False, None, "pow", False)
    nT = int(np.ceil(N / step))
    fr = torch.arange(nT, device="cuda", dtype=torch.int64) * step - nperseg // 2
    tr = torch.arange(T, device="cuda", dtype=torch.int64) * N
    starts = (tr[:, None] + fr[None, :]).reshape(-1).contiguous()
    lo = tr[:, None].expand(T, nT).reshape(-1).contiguous()
    hi = (lo + N).contiguous()
    out = torch.empty(plan.out_shape(T * nT), dtype=torch.float32, device="cuda")
    dt = sync_time(lambda: plan.execute(data, starts, lo, hi, out=out))
    byt = T * (N * C * 4 + nT * 251 * C * 4)
    res["c4_mtmconvol_500"] = {"trials_per_s": T / dt, "us_per_trial": 1e6 * dt / T, "GBps": byt / dt / 1e9,
                               "kernel": plan.kernel_name}
    print("conv500", res["c4_mtmconvol_500"], flush=True)
    del data, out

if "wav" in which:
    C, N, T = 128, 16384, 20
    data = synthdata.ar2_uncoupled_fast(C, N, T, seed=3)
    foi = np.arange(4, 104, 4, dtype=float)
    scales = (1 / foi) * (6 + np.sqrt(38)) / (4 * np.pi)
    plan = be.CWTPlan(N, C, scales, 1e-3, 6.0, 0, "pow")
    tr = torch.arange(T, device="cuda", dtype=torch.int64) * N
    out = torch.zeros(plan.out_shape(1), dtype=torch.float32, device="cuda")

    def run():          # trial average accumulated on the fly, as WaveletTransform.compute_hip does for keeptrials=False
        plan.execute(data, tr, tr, tr + N, out=out, accumulate=2)
    dt = sync_time(run, n=2)
    byt = T * (N * C * 4 + N * 25 * C * 4)
    res["c4_wavelet"] = {"trials_per_s": T / dt, "ms_per_trial": 1e3 * dt / T, "GBps_alg": byt / dt / 1e9}
    print("wav", res["c4_wavelet"], flush=True)
    del data, out

if "frontend" in which:
    # the whole front end on the headline shape: spy.connectivityanalysis(method="coh") on host-resident AnalogData
    # (first call: upload + plan creation + DPSS tapers; second call: trial queue already in HBM)
    import syncopy_amd as spy
    C, N, T = 256, 4096, 1000
    dev = synthdata.ar2_uncoupled_fast(C, N, T, seed=5)
    host = dev.cpu().numpy()
    del dev
    trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
    data = spy.AnalogData(host, samplerate=1000.0, trialdefinition=trl)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        coh = spy.connectivityanalysis(data, method="coh", tapsmofrq=1, polyremoval=0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res["frontend_coh"] = {"first_call_s": ts[0], "warm_call_s": min(ts[1:]), "trials_per_s_warm": T / min(ts[1:]),
                           "shape": list(coh.data.shape)}
    print("frontend", res["frontend_coh"], flush=True)
    del coh
    ts = []
    for kt in (True, True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pw = spy.freqanalysis(data, method="mtmfft", tapsmofrq=1, polyremoval=0, keeptrials=kt)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res["frontend_mtmfft"] = {"keeptrials_first_s": ts[0], "keeptrials_warm_s": ts[1], "trial_average_s": ts[2],
                              "shape": list(pw.data.shape)}
    print("frontend_mtmfft", res["frontend_mtmfft"], flush=True)
    del data, host, pw

if "jack" in which:
    # jackknife=True coherence on the headline channel count: streaming leave-one-out replicates on the device
    import syncopy_amd as spy
    C, N, T = 256, 4096, 200
    host = synthdata.ar2_uncoupled_fast(C, N, T, seed=6).cpu().numpy()
    trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
    data = spy.AnalogData(host, samplerate=1000.0, trialdefinition=trl)
    spy.connectivityanalysis(data, method="coh", tapsmofrq=1, polyremoval=0)        # upload + plans
    torch.cuda.synchronize()
    dt = 1e9
    for _ in range(3):                                   # best of three: the first call allocates pinned result buffers
        t0 = time.perf_counter()
        coh = spy.connectivityanalysis(data, method="coh", tapsmofrq=1, polyremoval=0, jackknife=True)
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    # K9 alone: leave-one-out replicates of 100 resident trials
    F, K = N // 2 + 1, 7
    g = torch.Generator(device="cuda").manual_seed(9)
    spec = torch.view_as_complex(torch.randn((100 * K, F, C, 2), device="cuda", generator=g))
    S = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    be.csd_accumulate(spec, S)
    be.csd_finalize(S, 1.0 / (100 * K))
    direct = be.coh_normalize(S, "abs")
    sd = torch.zeros((F, C, C), dtype=torch.float64, device="cuda")
    sd2 = torch.zeros_like(sd)
    dk = sync_time(lambda: be.jack_coh_accumulate(spec, K, S, direct, "abs", 100, sd, sd2), n=3)
    res["jackknife_coh"] = {"trials": T, "seconds": dt, "ms_per_replicate": 1e3 * dt / T,
                            "kernel_us_per_replicate": 1e6 * dk / 100,
                            "var_max": float(coh.jack_var.max()), "finite": bool(np.isfinite(coh.jack_var).all())}
    del spec, S, direct, sd, sd2
    print("jack", res["jackknife_coh"], flush=True)
    del data, host, coh

if "h2d" in which:
    # PCIe-inclusive view of the headline config: upload of the trial queue (host -> HBM) next to its compute time
    C, N, T = 256, 4096, 250
    host = np.random.default_rng(0).standard_normal((T * N, C), dtype=np.float32)
    torch.from_numpy(host[:1024]).cuda()                      # warm up the copy path
    torch.cuda.synchronize()

    def best(fn, n=3):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            del x
        return min(ts)

    t_page = best(lambda: torch.from_numpy(host).cuda())
    pinned = torch.from_numpy(host).pin_memory()
    t_pin = best(lambda: pinned.cuda(non_blocking=True))
    gb = host.nbytes / 1e9
    res["h2d_upload"] = {"GB": gb, "pageable_GBps": gb / t_page, "pinned_GBps": gb / t_pin,
                         "trials_per_s_pageable": T / t_page, "trials_per_s_pinned": T / t_pin}
    print("h2d", res["h2d_upload"], flush=True)
    del pinned, host

if "granger" in which or "granger256" in which:
    for C, N, T in (((64, 1024, 700), (256, 4096, 300)) if "granger" in which else ((256, 4096, 300),)):
        data = synthdata.ar2_uncoupled_fast(C, N, T, seed=4)
        K = 7
        F = N // 2 + 1
        starts = torch.arange(T, device="cuda", dtype=torch.int64) * N
        plan = be.FFTPlan(N, N, C, windows.dpss(N, 4.096, K) * np.sqrt(N), np.sqrt(2) / N, 0, True, None, "fourier", True)
        acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
        B = 50
        for b0 in range(0, T, B):
            be.csd_accumulate(plan.execute(data, starts[b0:b0 + B]), acc)
        be.csd_finalize(acc, 1.0 / (K * T))
        torch.cuda.synchronize()
        del data
        t0 = time.perf_counter()
        G, meta = be.granger(acc, niter=100)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[f"granger_C{C}_F{F}"] = {"seconds": dt, **meta, "finite": bool(torch.isfinite(G).all())}
        print("granger", C, res[f"granger_C{C}_F{F}"], flush=True)
        del acc, G

os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/configs.json", "w"), indent=1)
