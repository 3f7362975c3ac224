#!/usr/bin/env python
"""Headline benchmark: trials/s for mtmfft + coherence (BASELINE.json configs[2] data shape:
256 channels x 4096 samples x 1000 trials per GPU, 7 DPSS tapers, full 256x256 CSD).

One "step" = one complete pass of the hot path over the rank's in-HBM trial queue:
    for every batch of trials:  detrend -> taper -> FFT (complex spectra, all tapers)   [K1]
                                 acc += X X^H on the fp32 matrix cores                     [K4]
    (N > 1) RCCL all-reduce of the accumulator's packed lower triangle over xGMI           [C1]
    scale + coherence normalisation + Hermitian mirror (one fused pass) -> (F, C, C) float32 [K5]
Trials shard across ranks with no other exchange (weak scaling: 1000 trials per GPU).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--trials", type=int, default=1000, help="trials per GPU")
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=1000, help="trials per FFT/CSD launch pair")
    ap.add_argument("--blocked", action="store_true",
                    help="FFT -> CSD hand-over in the channel-blocked layout (faster FFT stores, slower CSD fetch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(nchan, nsamp, budget_s=30.0):
    """Reference-faithful CPU path (oracle = port of the reference's NumPy/SciPy calls) timed on
    the host: single-trial cross spectra exactly as csd.py:94-102 does them (incl. the
    (K,F,C,C) temporary), the per-trial cost of the reference's compute_sequential loop."""
    from oracle import spy_oracle as O
    rng = np.random.default_rng(0)
    x = rng.normal(size=(nsamp, nchan)).astype(np.float32)
    topt = {"NW": 1.0 * nsamp / 1000.0, "Kmax": 7}
    n = 0
    t0 = time.perf_counter()
    while True:
        O.cross_spectra_cF(x.copy(), samplerate=1000.0, nSamples=nsamp, foi=None, taper="dpss", taper_opt=topt,
                           polyremoval=0, faithful=True)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s / 2 or n >= 8:
            break
    return {"value": n / el, "unit": "trials/s", "cores": 1, "kind": "port",
            "sample": f"{n} trial(s) of {nchan} ch x {nsamp} samples, cross_spectra_cF (mtmfft + (K,F,C,C) outer "
                      f"product + taper mean, as csd.py:94-102), {el:.1f} s on one host core"}


def pmc_traffic(nrows, nfreq, nchan):
    """HBM bytes per CSD launch from the committed counter passes (profiles/r1_pmc_counters_final.txt:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of tools/pmc_harness.cpp, same launch
    shape).  Units are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM) prescribes for 16-byte
    per-lane streaming reads on gfx950.  Counters are only valid for the shape they were taken on
    (first line of the file)."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_counters_final.txt")
    if not os.path.exists(path):
        return None
    total, cur = 0.0, None
    for ln in open(path):
        if ln.startswith("#"):
            if f"rows={nrows} F={nfreq} C={nchan} " not in ln:
                return None
            continue
        if not ln.startswith(" "):
            cur = ln.strip()
            continue
        if cur is None or ("csd_accum_kernel" not in cur and "csd_reduce_parts" not in cur):
            continue
        name, rest = ln.split()[0], ln.split("mean=")[1]
        if name == "FETCH_SIZE":
            total += 2.0 * 1024.0 * float(rest)
        elif name == "WRITE_SIZE":
            total += 1024.0 * float(rest)
    return total or None


class _QuietStdout:
    """RCCL prints a version banner to stdout when the first communicator is created; the contract is ONE JSON
    line on stdout, so file descriptor 1 points at /dev/null while the process group comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(null, 1)
        os.close(null)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: syncopy_amd has no CPU path")
    torch.cuda.set_device(local)
    # SPY_BENCH_FORCE_DIST=1: run the process-group path (init, all-reduce, barrier) even with one rank - lets the
    # collective code be exercised on a 1-GPU box under torch.distributed.run
    dist_on = world > 1 or bool(os.environ.get("SPY_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with _QuietStdout():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            warm = torch.zeros(8, device="cuda")
            dist.all_reduce(warm)                  # creates the communicator (and its banner) here
            torch.cuda.synchronize()

    from scipy.signal import windows
    from syncopy_amd import backend as be
    from syncopy_amd import synthdata

    C, N, T, K = args.channels, args.samples, args.trials, 7
    F = N // 2 + 1
    fs = 1000.0
    NW = 1.0 * N / fs                      # tapsmofrq = 1 Hz -> NW = 4.096, Kmax = 7
    tapers = windows.dpss(N, NW, K) * np.sqrt(N)
    scale = np.sqrt(2) / N
    data = synthdata.ar2_uncoupled_fast(C, N, T, seed=1234 + rank)          # (T*N, C) float32 in HBM
    starts_all = torch.arange(T, device="cuda", dtype=torch.int64) * N
    plan = be.FFTPlan(N, N, C, tapers, scale, detrend=0, demean_taper=False, freq_idx=None, output="fourier",
                      keeptapers=True)
    blocked = args.blocked and plan.set_blocked(True)
    B = min(args.batch, T)
    spec = torch.empty(plan.out_shape(B), dtype=torch.complex64, device="cuda")
    acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    ev_csd, ev_fft = [], []

    def step(timed):
        acc.zero_()
        for b0 in range(0, T, B):
            nb = min(B, T - b0)
            sp = spec[:nb * (K if blocked else 1)]
            if timed:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
            plan.execute(data, starts_all[b0:b0 + nb], out=sp)
            if timed:
                e1.record()
            be.csd_accumulate(sp, acc, blocked=blocked)
            if timed:
                e2.record()
                ev_fft.append((e0, e1, nb))
                ev_csd.append((e1, e2, nb))
        if dist_on:
            # the accumulator carries its lower triangle only: pack -> all-reduce -> unpack (0.54 GB instead of 1.07)
            packed = be.csd_tril_pack(acc)
            dist.all_reduce(torch.view_as_real(packed))
            be.csd_tril_unpack(packed, acc)
        # K5 fused: scale + coherency + |.| + Hermitian mirror straight from the raw accumulator
        return be.coh_from_accumulator(acc, 1.0 / (K * T * world), "abs")

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        coh = step(True)
    fence()
    el = time.perf_counter() - t0
    tmax = torch.tensor([el], device="cuda", dtype=torch.float64)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax.item())

    if rank == 0:
        assert bool(torch.isfinite(coh).all()), "non-finite coherence"
        diag = coh[:, torch.arange(C), torch.arange(C)]
        assert float((diag - 1).abs().max()) < 1e-5, "coherence diagonal must be 1"
        csd_ms = [a.elapsed_time(b) for a, b, _ in ev_csd]
        fft_ms = [a.elapsed_time(b) for a, b, _ in ev_fft]
        rows = [nb * K for _, _, nb in ev_csd]
        flops = [8.0 * r * F * C * (C + 1) / 2 for r in rows]                 # Hermitian-minimal, SURVEY 8d
        achieved = sum(flops) / (sum(csd_ms) * 1e-3) / 1e12
        fft_bytes = sum(nb * (N * C * 4 + K * F * C * 8) for _, _, nb in ev_fft)
        value = world * T * args.steps / el
        line = {
            "metric": "trials/sec for mtmfft+coherence (256 ch x 4096 samples, 7 DPSS tapers, full CSD)",
            "value": value,
            "unit": "trials/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: connectivityanalysis method='coh' on AR(2) AnalogData, "
                            f"{C} ch x {N} samp x {T} trials per GPU, tapsmofrq=1 Hz (NW={NW:.3f}, 7 tapers), "
                            "polyremoval=0, output='abs', inputs resident in HBM",
                "trials_per_gpu": T, "channels": C, "samples": N, "tapers": K, "freqs": F, "batch": B, "handover_layout": "blocked" if blocked else "standard",
                "channel_samples_per_s": value * N * C,
                "fft_kernel": plan.kernel_name,
                "fft_ms_per_trial": sum(fft_ms) / (T * args.steps),
                "fft_stream_GBps": fft_bytes / (sum(fft_ms) * 1e-3) / 1e9,
                "csd_ms_per_trial": sum(csd_ms) / (T * args.steps),
            },
            "roofline": {
                "bound": "mfma",
                "kernel": ("spycsd::csd_accum_kernel<5, 4, %d>" % (1 if C == 256 else 2 if C < 256 else 3) if C <= 512 and not blocked
                           else "spycsd::csd_accum_kernel<5, 4, 0>") + " (+ row-split <1, 1> tail and its reduction)",
                "achieved": achieved,
                "peak": PEAK_MFMA_F32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_MFMA_F32_TFLOPS,
                "flop_per_launch": flops[0],
                "avg_launch_ms": float(np.mean(csd_ms)),
                "algorithmic_hbm_bytes_per_launch": rows[0] * F * C * 8 + 2 * F * (((C + 31) // 32) * ((C + 31) // 32 + 1) // 2) * 1024 * 8,
                "traffic": pmc_traffic(rows[0], F, C),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(C, N)
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
