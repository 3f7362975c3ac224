#!/usr/bin/env python
"""Headline benchmark: trials/s for mtmfft + coherence (BASELINE.json configs[2] data shape:
256 channels x 4096 samples x 1000 trials per GPU, 7 DPSS tapers, full 256x256 CSD).

One "step" = one complete pass of the hot path over the rank's in-HBM trial queue:
    reference-order float32 mean of every trial                                            [K0  seq_mean_kernel]
    detrend -> taper -> FFT (complex spectra, all tapers) + per-channel range              [K1  mtmfft_quad_kernel]
    acc += X X^H: float32 operands as fp16 (hi, lo) pairs on the half-precision matrix
        cores, float32 accumulation; frequencies a pair cannot hold redone in float32      [K4h csdh_kernel]
    (N > 1) RCCL all-reduce of the accumulator's packed lower triangle over xGMI           [C1]
    scale + coherence normalisation + Hermitian mirror (one fused pass) -> (F, C, C) float32 [K5]
Trials shard across ranks with no other exchange (weak scaling: 1000 trials per GPU).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            (no launcher: this file starts the N ranks itself, or exits 2 if the node
                                             has fewer than N GPUs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line, kept under 4 KB: headline, `roofline` (K4h), `selfcheck` (the timed launch against
complex128), at N = 1 also the twins of the headline at the reference's operand precision (`value_f32_products`,
`value_reference_arithmetic`), a compact `secondary` (c2 mtmfft power, other trial lengths, c4 sliding-window FFT and
Morlet wavelets, the front-end- and host-copy-inclusive headline, c5 Wilson / Granger) and a compact `cpu_baseline`
(the oracle on the host cores: BASELINE.md section 4.2).  Everything longer - the full secondary entries with kernel
names, bounds and counter provenance, notes, the CPU-baseline variants - goes to gpurun_out/bench_detail.json.
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
PEAK_MFMA_F16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16 / FP16 MFMA, dense (2:1 sparsity figures are not a peak)
PROFILE_ROUND = "r6"            # profiles/<round>_* written by tools/final_bench.sh on the code of this round
PEAK_F64_TFLOPS = 78.6          # MI355X_MICROARCH.md: FP64 vector / matrix
PEAK_HBM_GBS = 8000.0
# SURVEY 8(d) bound of the headline on one GPU: K4's 3.775e9 algorithmic flops per trial at 157.3 TFLOP/s = 24.0 us
# -> 41.7 k trials/s (the transform stage in front of it is not in this bound: headline_frac prices the WHOLE step,
# K0 + K1 + K4 + K5 back to back, against K4's matrix bound alone)
HEADLINE_BOUND_TRIALS_PER_S = 41.7e3


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
    ap.add_argument("--no-reference-mean", action="store_true",
                    help="detrend with float64 block sums instead of the reference's float32 row-order mean (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the c2 / c4 / c5 secondary measurements")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (= port of the reference's NumPy/SciPy calls) timed on the host cores
# ----------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_worker(args):
    """One process of the one-process-per-core baseline: `n` trials of the coherence ST stage, BLAS/FFT threads = 1.
    mode "faithful": cross_spectra_cF as the reference runs it (materialises the (K,F,C,C) product, csd.py:98);
    "einsum": the oracle's accumulation without that temporary; "blas": best-effort CPU - the same mtmfft, then one
    complex64 matrix product per frequency through BLAS (np.matmul on (F, C, K) x (F, K, C), 64 frequencies at a
    time, accumulated in place into the trial sum)."""
    nchan, nsamp, n, mode, seed = args
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import spy_oracle as O
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(nsamp, nchan)).astype(np.float32)
    topt = {"NW": 1.0 * nsamp / 1000.0, "Kmax": 7}
    acc = None
    t0 = time.perf_counter()
    for _ in range(n):
        if mode == "blas":
            specs, _ = O.mtmfft(O.detrend(x.copy(), 0), 1000.0, nsamp, "dpss", topt)        # (K, F, C) complex64
            A = np.ascontiguousarray(specs.transpose(1, 2, 0))                               # (F, C, K)
            B = np.ascontiguousarray(specs.conj().transpose(1, 0, 2))                        # (F, K, C)
            if acc is None:
                acc = np.zeros((A.shape[0], nchan, nchan), dtype=np.complex64)
            inv = np.float32(1.0 / specs.shape[0])
            for f0 in range(0, A.shape[0], 64):
                blk = np.matmul(A[f0:f0 + 64], B[f0:f0 + 64])
                blk *= inv
                acc[f0:f0 + 64] += blk
        else:
            r, _ = O.cross_spectra_cF(x.copy(), samplerate=1000.0, nSamples=nsamp, foi=None, taper="dpss", taper_opt=topt,
                                      polyremoval=0, faithful=(mode == "faithful"))
            acc = r if acc is None else acc.__iadd__(r)      # the trial sum of compute_sequential (:1022-1032)
    return time.perf_counter() - t0


def _cpu_noop(_):
    import numpy  # noqa: F401  (worker start-up: interpreter + NumPy/SciPy import, kept out of the timed map)
    from oracle import spy_oracle  # noqa: F401
    return 0


def cpu_baseline(nchan, nsamp):
    """BASELINE.md section 4.2 on the GPU box's host: the reference's per-trial coherence ST stage
    (mtmfft + outer product + taper mean, csd.py:94-102) as restated by the oracle,
      (a) one process, one trial at a time = compute_sequential (computational_routine.py:944);
      (b) one process per PHYSICAL core over disjoint trials = the Dask LocalCluster trial map (:926), BLAS threads
          = 1, inputs in RAM, worker start-up not timed; as many processes as half of the usable memory allows
          (cgroup limit respected; the per-process footprints below are measured peaks, rounded up);
    as "reference-faithful" (materialises the (K,F,C,C) product like csd.py:98: ~15 GB per process at 256 x 4096),
    "einsum" (the oracle's accumulation without that temporary, ~3.2 GB) and "best-effort BLAS" (one cgemm per
    frequency, accumulated in place: ~1.4 GB, so every physical core gets a process).
    `value` is (a)-faithful: the reference's own code path on one core."""
    import multiprocessing as mp
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or 1
        avail = psutil.virtual_memory().available
    except Exception:
        phys, avail = os.cpu_count() or 1, 16 << 30
    # the memory this process may really use: a container's cgroup limit can be far below what the host reports
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            lim = open(path).read().strip()
            if lim.isdigit():
                avail = min(avail, int(lim))
        except OSError:
            pass
    budget = avail // 2
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else phys
    ncore = max(1, min(phys, usable))
    t_f = _cpu_worker((nchan, nsamp, 1, "faithful", 0))
    t_e = _cpu_worker((nchan, nsamp, 1, "einsum", 0))
    t_b = _cpu_worker((nchan, nsamp, 1, "blas", 0))
    res_bytes = (nsamp // 2 + 1) * nchan * nchan * 8
    variants = {
        "faithful_1core": {"value": 1.0 / t_f, "cores": 1, "s_per_trial": t_f},
        "einsum_1core": {"value": 1.0 / t_e, "cores": 1, "s_per_trial": t_e},
        "best_effort_blas_1core": {"value": 1.0 / t_b, "cores": 1, "s_per_trial": t_b},
    }
    ctx = mp.get_context("spawn")
    for name, mode, per_proc, ntr in (("faithful_process_per_core", "faithful", 14 * res_bytes, 1),
                                      ("einsum_process_per_core", "einsum", 3 * res_bytes, 1),
                                      ("best_effort_blas_process_per_core", "blas", int(1.3 * res_bytes), 2)):
        nproc = int(min(ncore, budget // max(per_proc, 1)))
        if nproc < 2:
            variants[name] = {"value": None, "cores": 0, "note": "skipped: not enough memory for two processes"}
            continue
        with ctx.Pool(nproc) as pool:
            pool.map(_cpu_noop, range(nproc), chunksize=1)
            t0 = time.perf_counter()
            pool.map(_cpu_worker, [(nchan, nsamp, ntr, mode, 100 + i) for i in range(nproc)], chunksize=1)
            t_all = time.perf_counter() - t0
        variants[name] = {"value": nproc * ntr / t_all, "cores": nproc, "cores_available": ncore, "trials": nproc * ntr,
                          "wall_s": t_all, "GB_per_process_budgeted": per_proc / 1e9}
    best = max((v for v in variants.values() if v.get("value")), key=lambda v: v["value"])
    return {
        "value": 1.0 / t_f, "unit": "trials/s", "cores": 1, "kind": "port",
        "sample": f"1 trial of {nchan} ch x {nsamp} samples, cross_spectra_cF reference-faithful (mtmfft + (K,F,C,C) outer "
                  f"product + taper mean, as csd.py:94-102), {t_f:.1f} s on one host core",
        "cpu_model": _cpu_model(), "physical_cores": phys, "usable_cores": usable,
        "usable_memory_GB": avail / 1e9, "memory_budget_GB": budget / 1e9,
        "best_cpu_trials_per_s": best["value"], "best_cpu_cores": best["cores"],
        "variants": variants,
        "note": "process-per-core variants: BLAS threads = 1, inputs in RAM, worker start-up excluded; one process per "
                "physical core unless half of the usable memory (cgroup limit respected) holds fewer",
    }


# ----------------------------------------------------------------------------------------------------------------
def k4_sources_sha():
    """Hash of the K4 kernel sources: written into the PMC summary by tools/pmc_run.sh, compared here - counters taken on
    other kernel code are not reported."""
    import hashlib
    h = hashlib.sha256()
    for name in ("csdh_kernel.h", "csdh.hip", "csd3m_kernel.h", "csd3m_launch_impl.h", "csd.hip", "csd_kernel.h", "csd_args.h"):
        with open(os.path.join(ROOT, "syncopy_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(nrows, nfreq, nchan):
    """HBM bytes per CSD launch from the committed counter passes (profiles/r*_pmc_counters_final.txt: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs of tools/pmc_harness.cpp, same launch shape; rocprofv3 counter collection
    crashes inside torch's own kernels, so the passes cannot run inside this process).  Units are KiB; FETCH_SIZE is
    doubled as MI355X_MICROARCH.md (HBM) prescribes for 16-byte per-lane streaming reads on gfx950.  Returns
    (bytes or None, provenance): the counters are only reported for the launch shape AND the kernel sources they were
    taken on (first line of the file: shape and `k4_sources_sha`)."""
    sha = k4_sources_sha()
    for name in (PROFILE_ROUND + "_pmc_headline.txt", "r5_pmc_headline.txt"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return None, {"from_profile": None}
    prov = {"from_profile": "profiles/" + name, "k4_sources_sha": sha}
    total, cur = 0.0, None
    for ln in open(path):
        if ln.startswith("#"):
            if f"rows={nrows} F={nfreq} C={nchan} " not in ln:
                prov["refused"] = "profile taken on another launch shape"
                return None, prov
            if f"k4_sources_sha={sha}" not in ln:
                prov["refused"] = "profile taken on other K4 kernel sources (stale)"
                return None, prov
            continue
        if not ln.startswith(" "):
            cur = ln.strip()
            continue
        if cur is None or not any(k in cur for k in ("csd_accum_kernel", "csd_reduce_parts", "csd3m_kernel", "csdh_kernel")):
            continue
        name, rest = ln.split()[0], ln.split("mean=")[1]
        if name == "FETCH_SIZE":
            total += 2.0 * 1024.0 * float(rest)
        elif name == "WRITE_SIZE":
            total += 1024.0 * float(rest)
    return (total or None), prov


PMC_SECONDARY = PROFILE_ROUND + "_pmc_secondary.txt"


def pmc_secondary(mode):
    """HBM bytes per trial of one `secondary` workload from the committed counter file (profiles/<round>_pmc_secondary.txt,
    written by tools/final_bench.sh through the torch-free tools/pmc_harness2.cpp - same plans, shapes and launch
    arguments): sum over the workload's kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 x dispatches, divided by the
    repetitions and trials of the harness run (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    Returns (bytes per trial or None, provenance)."""
    path = os.path.join(ROOT, "profiles", PMC_SECONDARY)
    prov = {"from_profile": "profiles/" + PMC_SECONDARY, "section": mode}
    if not os.path.exists(path):
        return None, {"from_profile": None}
    import re
    total, inside, T, reps, kernels = 0.0, False, None, None, []
    for ln in open(path):
        if ln.startswith("## "):
            inside = ln.split()[1] == mode
            if inside:
                mt, mr = re.search(r"T=(\d+)", ln), re.search(r"reps=(\d+)", ln)
                T, reps = (int(mt.group(1)) if mt else None), (int(mr.group(1)) if mr else None)
            continue
        if not inside:
            continue
        if not ln.startswith(" "):
            kernels.append(ln.split("[")[0].strip())
            continue
        name = ln.split()[0]
        mn, mm = re.search(r"n=\s*(\d+)", ln), re.search(r"mean=([0-9.eE+-]+)", ln)
        if name in ("FETCH_SIZE", "WRITE_SIZE") and mn and mm:
            total += (2.0 if name == "FETCH_SIZE" else 1.0) * 1024.0 * float(mm.group(1)) * int(mn.group(1))
    if not total or not T or not reps:
        return None, prov
    prov.update({"trials": T, "reps": reps, "kernels": kernels})
    return total / (T * reps), prov


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


def _event_ms(torch, fn, reps=3):
    """Average duration of fn() in ms between events on torch's current stream (= the library's stream)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _traffic(mode, algorithmic_bytes):
    """`traffic` fields of a secondary entry: counter bytes per trial, their ratio to the algorithmic bytes, the files a
    reader needs to recompute the entry (kernel durations: profiles/<round>_secondary_kernel_stats.txt, section `mode`)."""
    t, prov = pmc_secondary(mode)
    return {"traffic_bytes_per_trial": t, "traffic_over_algorithmic": (t / algorithmic_bytes) if t else None,
            "traffic_source": prov, "kernel_stats": "profiles/" + PROFILE_ROUND + "_secondary_kernel_stats.txt#" + mode}


def secondary(torch, be, synthdata, data, N, C, T, refmean=True):
    """The other SURVEY 8(d) numbers, inputs resident in HBM, each priced against the bound SURVEY 8(d) names:
    frac = max(bytes / 8 TB/s, flops / peak) / measured time."""
    from scipy.signal import windows
    out = []
    K = 7
    F = N // 2 + 1
    # ---- c2: mtmfft power spectra, 7 tapers, taper mean (configs[1])
    tapers = windows.dpss(N, 1.0 * N / 1000.0, K) * np.sqrt(N)
    plan = be.FFTPlan(N, N, C, tapers, np.sqrt(2) / N, 0, False, None, "pow", False, reference_mean=refmean)
    starts = torch.arange(T, device="cuda", dtype=torch.int64) * N
    buf = torch.empty(plan.out_shape(T), dtype=torch.float32, device="cuda")
    ms = _event_ms(torch, lambda: plan.execute(data, starts, out=buf))
    byt, flop = N * C * 4 + F * C * 4, K * C * 2.5 * N * np.log2(N)
    bound_us = max(byt / (PEAK_HBM_GBS * 1e9), flop / (PEAK_MFMA_F32_TFLOPS * 1e12)) * 1e6
    out.append({"name": "c2 mtmfft power (BASELINE configs[1]): %d ch x %d samp x %d trials, 7 DPSS tapers, taper mean" % (C, N, T),
                "value": T / (ms * 1e-3), "unit": "trials/s", "us_per_trial": 1e3 * ms / T,
                "channel_samples_per_s": T / (ms * 1e-3) * N * C, "kernel": plan.kernel_name,
                "bound": "fft-flop (fp32 vector peak) vs hbm, whichever is larger", "bound_us_per_trial": bound_us,
                "bytes_per_trial": byt, "flop_per_trial": flop, "frac": bound_us / (1e3 * ms / T),
                "hbm_GBps": byt * T / (ms * 1e-3) / 1e9, **_traffic("c2", byt)})
    # ---- the same under precision="reference" (float64 taper product and transform, mtmfft_dec64_kernel.h)
    if plan.set_precision(True):
        ms64 = _event_ms(torch, lambda: plan.execute(data, starts, out=buf))
        out.append({"name": "c2 under precision='reference' (float64 transform, complex64 rounding where mtmfft.py:104-127 rounds)",
                    "value": T / (ms64 * 1e-3), "unit": "trials/s", "us_per_trial": 1e3 * ms64 / T, "kernel": plan.kernel_name,
                    "ratio_to_float32": ms64 / ms, "bound": "fp64 vector %.1f TFLOP/s on the same flop count" % PEAK_F64_TFLOPS,
                    "bound_us_per_trial": flop / (PEAK_F64_TFLOPS * 1e12) * 1e6,
                    "frac": flop / (PEAK_F64_TFLOPS * 1e12) / (1e-3 * ms64 / T), **_traffic("c2f64", byt)})
    del buf, plan
    # ---- c2 at trial lengths that are not powers of two (1 kHz x 2 / 3 / 5 / 10 / 12 s; BASELINE configs[0] is N = 2000):
    # the compile-time schedules K1d (3000 = 3 x 1000 through the radix-3 decimation, 10000 with split exchanges) and,
    # beyond one workgroup's LDS in quad form, the same schedules on channel PAIRS (CfgD::HALF: 12000 through the 6000-point
    # schedule, 16384 - configs[3]'s trial length - through the 8192-point one)
    for N2 in (2000, 3000, 5000, 10000, 12000, 16384):
        T2 = 200
        d2 = synthdata.ar2_uncoupled_fast(C, N2, T2, seed=78)
        tp2 = windows.dpss(N2, 1.0 * N2 / 1000.0, K) * np.sqrt(N2)
        plan = be.FFTPlan(N2, N2, C, tp2, np.sqrt(2) / N2, 0, False, None, "pow", False, reference_mean=refmean)
        st2 = torch.arange(T2, device="cuda", dtype=torch.int64) * N2
        buf = torch.empty(plan.out_shape(T2), dtype=torch.float32, device="cuda")
        ms = _event_ms(torch, lambda: plan.execute(d2, st2, out=buf))
        F2 = N2 // 2 + 1
        byt, flop = N2 * C * 4 + F2 * C * 4, K * C * 2.5 * N2 * np.log2(N2)
        bound_us = max(byt / (PEAK_HBM_GBS * 1e9), flop / (PEAK_MFMA_F32_TFLOPS * 1e12)) * 1e6
        out.append({"name": "c2 shape at N = %d%s: %d ch x %d samp x %d trials, 7 DPSS tapers, taper mean"
                            % (N2, "" if N2 & (N2 - 1) == 0 else " (not a power of two)", C, N2, T2),
                    "value": T2 / (ms * 1e-3), "unit": "trials/s", "us_per_trial": 1e3 * ms / T2,
                    "channel_samples_per_s": T2 / (ms * 1e-3) * N2 * C, "kernel": plan.kernel_name,
                    "bound": "fft-flop (fp32 vector peak) vs hbm, whichever is larger", "bound_us_per_trial": bound_us,
                    "bytes_per_trial": byt, "flop_per_trial": flop, "frac": bound_us / (1e3 * ms / T2),
                    **_traffic("n%d" % N2, byt)})
        if plan.set_precision(True):
            ms64 = _event_ms(torch, lambda: plan.execute(d2, st2, out=buf))
            out[-1]["reference_precision"] = {"us_per_trial": 1e3 * ms64 / T2, "ratio_to_float32": ms64 / ms,
                                              "kernel": plan.kernel_name, **_traffic("n%df64" % N2, byt)}
        del buf, plan, d2
    # ---- c4: 128 ch x 16384 samples (configs[3]); (i) 512-sample Hann windows, 50 % overlap
    C4, N4, T4 = 128, 16384, 200
    d4 = synthdata.ar2_uncoupled_fast(C4, N4, T4, seed=77)
    nperseg, step = 512, 256
    w = windows.hann(nperseg)
    w = w * np.sqrt(4 / 3) * np.sqrt(nperseg / w.sum())
    plan = be.FFTPlan(nperseg, nperseg, C4, w[None], np.sqrt(2) / nperseg, 0, False, None, "pow", False)
    nT = int(np.ceil(N4 / step))
    fr = torch.arange(nT, device="cuda", dtype=torch.int64) * step - nperseg // 2
    tr = torch.arange(T4, device="cuda", dtype=torch.int64) * N4
    st = (tr[:, None] + fr[None, :]).reshape(-1).contiguous()
    lo = tr[:, None].expand(T4, nT).reshape(-1).contiguous()
    hi = (lo + N4).contiguous()
    buf = torch.empty(plan.out_shape(T4 * nT), dtype=torch.float32, device="cuda")
    ms = _event_ms(torch, lambda: plan.execute(d4, st, lo, hi, out=buf))
    byt = N4 * C4 * 4 + nT * (nperseg // 2 + 1) * C4 * 4
    out.append({"name": "c4 mtmconvol (BASELINE configs[3] i): %d ch x %d samp x %d trials, hann nperseg 512, 50 %% overlap, pow" % (C4, N4, T4),
                "value": T4 / (ms * 1e-3), "unit": "trials/s", "us_per_trial": 1e3 * ms / T4, "kernel": plan.kernel_name,
                "bound": "hbm", "bytes_per_trial": byt, "bound_us_per_trial": byt / (PEAK_HBM_GBS * 1e9) * 1e6,
                "frac": byt / (PEAK_HBM_GBS * 1e9) / (1e-3 * ms / T4), "hbm_GBps": byt * T4 / (ms * 1e-3) / 1e9,
                **_traffic("conv", byt)})
    del buf, plan
    # ---- c4 (ii): Morlet wavelets, 25 scales 4 .. 100 Hz, every sample, trial average accumulated on the device
    foi = np.arange(4, 104, 4, dtype=float)
    scales = (1 / foi) * (6 + np.sqrt(38)) / (4 * np.pi)
    plan = be.CWTPlan(N4, C4, scales, 1e-3, 6.0, 0, "pow")
    res = torch.zeros(plan.out_shape(1), dtype=torch.float32, device="cuda")
    ms = _event_ms(torch, lambda: plan.execute(d4, tr, tr, tr + N4, out=res, accumulate=2), reps=2)
    byt = N4 * C4 * 4 + N4 * 25 * C4 * 4          # keeptrials=True accounting of SURVEY 8(d)
    flop_lo = 26 * C4 * 5 * 18816 * np.log2(18816)
    bound_us = max(byt / (PEAK_HBM_GBS * 1e9), flop_lo / (PEAK_MFMA_F32_TFLOPS * 1e12)) * 1e6
    out.append({"name": "c4 wavelet (BASELINE configs[3] ii): %d ch x %d samp x %d trials, Morlet w0=6, 25 scales 4..100 Hz, toi=all, pow, trial average" % (C4, N4, T4),
                "value": T4 / (ms * 1e-3), "unit": "trials/s", "us_per_trial": 1e3 * ms / T4,
                "kernel": "spycwt::cwt2_kernel (4 block groups) + cwt_scatter_kernel",
                "bound": "max(hbm at keeptrials accounting, fft-flop lower estimate)", "bytes_per_trial": byt,
                "flop_per_trial": flop_lo, "bound_us_per_trial": bound_us, "frac": bound_us / (1e3 * ms / T4),
                **_traffic("wav", byt)})
    del res, plan
    # ... and the same two analyses through spy.freqanalysis (argument checks, dry runs, every trial's frames in one launch,
    # trial average on the device, the result copied to the host): warm calls on resident data
    import syncopy_amd as spy
    a4 = spy.AnalogData(d4.cpu().numpy(), samplerate=1000.0,
                        trialdefinition=np.stack([np.arange(T4) * N4, np.arange(1, T4 + 1) * N4, np.zeros(T4)], axis=1))
    del d4

    def _warm_call(fn, reps=3):
        fn()
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            _ = r.data.shape
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    t_conv = _warm_call(lambda: spy.freqanalysis(a4, method="mtmconvol", taper="hann", t_ftimwin=0.512, toi=0.5, keeptrials=False))
    t_wav = _warm_call(lambda: spy.freqanalysis(a4, method="wavelet", wavelet="Morlet", width=6, foi=np.arange(4.0, 104.0, 4.0),
                                                toi="all", keeptrials=False), reps=2)
    for e in out:
        if e["name"].startswith("c4 mtmconvol"):
            e["front_end_warm_call_s"], e["front_end_us_per_trial"] = t_conv, 1e6 * t_conv / T4
        if e["name"].startswith("c4 wavelet"):
            e["front_end_warm_call_s"], e["front_end_us_per_trial"] = t_wav, 1e6 * t_wav / T4
    del a4
    spy.release_device_buffers()
    # ---- headline through the front end: spy.connectivityanalysis(method="coh") on host-resident AnalogData.  First
    # call = PCIe-inclusive (trial queue uploaded host -> HBM, plans and tapers built); warm call = front-end inclusive
    # (argument checks, dry run, plan lookup, kernels, copy of the 0.54 GB result to the host), queue resident
    import syncopy_amd as spy
    host = data.cpu().numpy()
    trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
    adata = spy.AnalogData(host, samplerate=1000.0, trialdefinition=trl)
    spy.release_device_buffers()                       # as in a fresh process: no cached device blocks, every buffer of
    torch.cuda.empty_cache()                           # the call (4.2 GB queue, spectra, result) is a real hipMalloc
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = spy.connectivityanalysis(adata, method="coh", tapsmofrq=1, polyremoval=0)
    torch.cuda.synchronize()
    t_cold = time.perf_counter() - t0                  # first analysis of the process: + DPSS tables (SciPy eigenproblem),
    del res, adata                                     # plan construction, code-object loading, device allocations
    adata = spy.AnalogData(host, samplerate=1000.0, trialdefinition=trl)       # the same recording as a NEW object: uploaded again
    ts, tcopy = [], []
    main_stream = torch.cuda.current_stream()
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = spy.connectivityanalysis(adata, method="coh", tapsmofrq=1, polyremoval=0)
        main_stream.synchronize()                      # the result is complete in HBM (the copies of its frequency ranges to
        ts.append(time.perf_counter() - t0)           # the host run on a side stream, the last of them still under way)
        shape = list(res.data.shape)                   # ... `.data` waits for that last range: the array is the landing block
        tcopy.append(time.perf_counter() - t0)
        del res
    tb = []
    torch.cuda.synchronize()
    for _ in range(2):                                 # the same call back to back, results dropped unread, no synchronisation
        t0 = time.perf_counter()                       # between the analyses: 4 calls per measurement
        for _ in range(4):
            res = spy.connectivityanalysis(adata, method="coh", tapsmofrq=1, polyremoval=0)
            del res
        torch.cuda.synchronize()
        tb.append((time.perf_counter() - t0) / 4)
    out.append({"name": "headline through the front end, result left in HBM: spy.connectivityanalysis(method='coh', tapsmofrq=1) on %d ch x %d samp x %d trials of host-resident AnalogData" % (C, N, T),
                "value": T / min(ts[1:]), "unit": "trials/s", "warm_call_s": min(ts[1:]), "warm_call_with_host_copy_s": min(tcopy[1:]),
                "value_definition": "trials / warm_call_s: the call has returned and the result is complete in HBM (main stream "
                                    "synchronised); its copy to the host runs frequency range by frequency range on a side "
                                    "stream under the cross-spectral products (backend.coh_pipeline), `.data` waits for the last range",
                "value_with_host_copy": T / min(tcopy[1:]),
                "back_to_back_warm_call_s": min(tb), "back_to_back_trials_per_s": T / min(tb),
                "first_call_s": ts[0], "first_call_with_host_copy_s": tcopy[0], "cold_process_first_call_s": t_cold,
                "pcie_inclusive_trials_per_s": T / ts[0], "result_shape": shape,
                "note": "back_to_back_*: four calls in a row, results dropped unread, one synchronisation at the end, per call.  "
                        "warm_call_s: argument checks, dry run, plan lookup, the 16-trial look of precision='auto', kernels, "
                        "result complete in HBM; warm_call_with_host_copy_s: until `.data` has been read.  first_call_s: the "
                        "first analysis of a recording that still sits in host memory - the %.1f GB trial queue goes up in "
                        "chunks on a copy stream, the transforms and CSD updates of chunk k run under the PCIe copy of chunk "
                        "k + 1 (backend.Upload; the bus alone needs %.0f ms at 57 GB/s); cold_process_first_call_s: the same "
                        "as the very first analysis of the process (DPSS tables, plans, code objects, every device buffer a fresh hipMalloc)"
                        % (host.nbytes / 1e9, host.nbytes / 57e9 * 1e3)})
    del adata, host
    spy.release_device_buffers()
    # ---- c5: Wilson / Granger AV stage on the CSD of the resident trials (demean_taper as method='granger' sets it)
    plan = be.FFTPlan(N, N, C, tapers, np.sqrt(2) / N, 0, True, None, "fourier", True, reference_mean=refmean)
    acc = torch.zeros((F, C, C), dtype=torch.complex64, device="cuda")
    Tg = T
    for timed_pass in (False, True):                   # (the first pass allocates the 14.7 GB spectra buffer)
        acc.zero_()
        torch.cuda.synchronize()
        t_st = time.perf_counter()
        for b0 in range(0, Tg, 500):
            be.csd_accumulate(plan.execute(data, starts[b0:b0 + 500]), acc)
        be.csd_finalize(acc, 1.0 / (K * Tg))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dt_st = t0 - t_st
    G, meta = be.granger(acc, niter=100)
    torch.cuda.synchronize()
    dt_first = time.perf_counter() - t0            # first call of the process: + code-object load and scratch allocation
    t0 = time.perf_counter()
    G, meta = be.granger(acc, niter=100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = be.granger_stats() if hasattr(be, "granger_stats") else {}
    iters = stats.get("iterations")
    flop_it = 2 * (F - 1) * 8.0 * C ** 3 * 6            # inverse + 4 GEMMs on the 2(F-1) lag-domain bins
    flop_it_exec = F * 8.0 * C ** 3 * 5                 # what K6 executes: inverse + 4 products on the F rfft bins
    entry = {"name": "c5 (BASELINE configs[4]) on one GPU: AV stage regularize_csd + wilson_sf + granger on %d x %d x %d (value), after the ST stage of %d trials with demean_taper (st_stage_s)" % (F, C, C, T),
             "value": dt, "unit": "s", "higher_is_better": False, "converged": meta["converged"],
             "max_rel_err": meta["max rel. err"], "cond0": meta["initial cond. num"], "iterations": iters,
             "kernel": "spywil::zinv_mfma_kernel / zgemm_mfma_kernel<0..3> / plus4_kernel", "bound": "fp64 %.1f TFLOP/s" % PEAK_F64_TFLOPS,
             "flop_per_iteration": flop_it, "executed_flop_per_iteration": flop_it_exec,
             "st_stage_s": dt_st, "st_trials": Tg, "st_plus_av_s": dt_st + dt, "first_call_s": dt_first,
             "kernel_stats": "profiles/" + PROFILE_ROUND + "_wilson_kernel_stats.csv", "counters": "profiles/" + PROFILE_ROUND + "_wilson_pmc.txt",
             "note": "frac prices the flops the kernels execute (conjugate symmetry: F of the reference's 2(F-1) bins) "
                     "against the fp64 matrix peak; algorithmic_frac uses SURVEY 8(d)'s count for the full spectrum"}
    if iters:
        entry["frac"] = iters * flop_it_exec / (PEAK_F64_TFLOPS * 1e12) / dt
        entry["algorithmic_frac"] = iters * flop_it / (PEAK_F64_TFLOPS * 1e12) / dt
    entry.update({k: v for k, v in stats.items() if k != "iterations"})
    out.append(entry)
    assert bool(torch.isfinite(G).all()), "non-finite Granger values"
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, the environment
    torch.distributed.run would give them) and return the worst exit code.  Rank 0 inherits stdout - its ONE JSON line is
    this command's line.  Refuses (exit 2) when the node has fewer than N GPUs: a line with another n_gpus than the one
    asked for is never printed."""
    import subprocess
    n = args.gpus
    if not os.environ.get("SPY_BENCH_LAUNCH_TEST"):
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible: not running\n" % (n, have))
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for q in pending:           # a rank that died leaves the others in a collective: end them (exact PIDs)
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def launch_test(args):
    """SPY_BENCH_LAUNCH_TEST=1: the launch / rendezvous / barrier / max-over-ranks plumbing of this file with a gloo process
    group and NO device work (tests/test_bench_launch.py runs `bench.py --gpus 2` this way on the CPU container).  The line
    says so in `data` and carries no value."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    seen = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"metric": "launch test", "value": None, "unit": "trials/s", "n_gpus": world, "ranks_seen": int(seen.item()),
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(el.item()) / max(args.steps, 1),
                          "data": "launch-test (no device work)"}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def _short(x, digits=4):
    """Numbers of the compact line: `digits` significant figures."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _short(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, digits) for v in x]
    return x


SECONDARY_KEYS = (("headline under precision", "ref_transforms_k4h"), ("c2 mtmfft power", "c2"), ("c2 under precision", "c2_f64"),
                  ("c2 shape at N = 2000", "n2000"), ("c2 shape at N = 3000", "n3000"), ("c2 shape at N = 5000", "n5000"),
                  ("c2 shape at N = 10000", "n10000"), ("c2 shape at N = 12000", "n12000"), ("c2 shape at N = 16384", "n16384"),
                  ("c4 mtmconvol", "c4_mtmconvol"), ("c4 wavelet", "c4_wavelet"), ("headline through the front end", "front_end"),
                  ("c5 (BASELINE", "c5_granger_av"))


def compact_secondary(entries):
    """One short record per `secondary` entry for the final line: us/trial (or the entry's own unit), the fraction of the
    bound it is priced against and counter traffic over algorithmic bytes.  The full entries go to the detail file."""
    out = {}
    for e in entries:
        key = next((k for pre, k in SECONDARY_KEYS if e["name"].startswith(pre)), None)
        if key is None:
            continue
        if key == "front_end":
            out[key] = _short({"warm": e["value"], "with_host_copy": e["value_with_host_copy"], "back_to_back": e["back_to_back_trials_per_s"],
                               "pcie_inclusive": e["pcie_inclusive_trials_per_s"], "unit": "trials/s"})
        elif key == "c5_granger_av":
            out[key] = _short({"s": e["value"], "iterations": e.get("iterations"), "frac": e.get("frac"), "st_stage_s": e["st_stage_s"]})
        elif key == "ref_transforms_k4h":
            out[key] = _short({"trials_per_s": e["value"]})
        else:
            rec = {"us_per_trial": e["us_per_trial"], "frac": e.get("frac")}
            if e.get("traffic_over_algorithmic"):
                rec["traffic_x"] = e["traffic_over_algorithmic"]
            if "reference_precision" in e:
                rec["f64_us_per_trial"] = e["reference_precision"]["us_per_trial"]
            out[key] = _short(rec, 3)
    return out


DETAIL_FILE = os.path.join("gpurun_out", "bench_detail.json")
MAX_LINE_BYTES = 4000


def emit(line, detail):
    """The contract's ONE JSON line, under 4 KB so that no capture window cuts it (round 5's 20 KB line reached the driver
    truncated); everything else (`secondary` in full, notes, CPU-baseline variants, provenance of every counter figure)
    goes to gpurun_out/bench_detail.json, which travels back with the run."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, DETAIL_FILE), "w") as fh:
            json.dump(detail, fh, indent=1)
        line["detail"] = DETAIL_FILE
    except OSError as exc:
        line["detail"] = "not written: %s" % exc
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("secondary", "step", "selfcheck_twins"):
        if len(text) <= MAX_LINE_BYTES:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= MAX_LINE_BYTES, len(text)
    print(text, flush=True)


def main():
    args = parse()
    if os.environ.get("WORLD_SIZE") is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        raise SystemExit(2)
    if os.environ.get("SPY_BENCH_LAUNCH_TEST"):
        raise SystemExit(launch_test(args))
    import torch
    import torch.distributed as dist

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
    refmean = not args.no_reference_mean
    plan = be.FFTPlan(N, N, C, tapers, scale, detrend=0, demean_taper=False, freq_idx=None, output="fourier",
                      keeptapers=True, reference_mean=refmean)
    blocked = args.blocked and plan.set_blocked(True)
    B = min(args.batch, T)
    spec = torch.empty(plan.out_shape(B), dtype=torch.complex64, device="cuda")
    # Two accumulators: with a process group the sum over ranks (and the coherence pass behind it) of step i runs on a
    # side stream while the main stream already transforms the trials of step i + 1 - every step is still one
    # complete analysis (FFT, CSD, ONE all-reduce of the packed lower triangle, coherence), the steps are pipelined.
    accs = [torch.zeros((F, C, C), dtype=torch.complex64, device="cuda") for _ in range(2 if dist_on else 1)]
    comm_stream = torch.cuda.Stream() if dist_on else None
    done = [None, None]                    # per accumulator: its last collective + coherence pass has finished
    rec_main = {"csd": [], "fft": [], "coll": []}
    nstep = [0]
    # per-channel range of the spectra (largest |re|, |im|), left by the transform kernel for K4h's operand scaling
    absmax = torch.zeros(C, dtype=torch.float32, device="cuda") if (C == 256 and not blocked) else None

    def step(rec, fft_plan=None, split=True):
        """One complete analysis of the rank's T trials.  `rec`: event lists to append this step's K1 / K4 / collective
        brackets to, or None (untimed).  `split=False`: the cross-spectral products on the float32 matrix instructions."""
        fft_plan = fft_plan or plan
        slot = nstep[0] % len(accs)
        nstep[0] += 1
        acc = accs[slot]
        main = torch.cuda.current_stream()
        if done[slot] is not None:
            main.wait_event(done[slot])
        acc.zero_()
        if absmax is not None:
            absmax.zero_()
        for b0 in range(0, T, B):
            nb = min(B, T - b0)
            sp = spec[:nb * (K if blocked else 1)]
            if rec is not None:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
            fft_plan.execute(data, starts_all[b0:b0 + nb], out=sp, absmax=absmax if split else None)
            if rec is not None:
                e1.record()
            be.csd_accumulate(sp, acc, blocked=blocked, absmax=absmax if (split and fft_plan.tracked_absmax) else None, split=split)
            if rec is not None:
                e2.record()
                rec["fft"].append((e0, e1, nb))
                rec["csd"].append((e1, e2, nb))
        if not dist_on:
            # K5 fused: scale + coherency + |.| + Hermitian mirror straight from the raw accumulator
            return be.coh_from_accumulator(acc, 1.0 / (K * T * world), "abs")
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready)
            # the accumulator carries its lower triangle only: the product's collective (backend.csd_allreduce_ =
            # spyhip_allreduce_csd: pack -> RCCL all-reduce on the library's communicator -> unpack, 0.54 GB of 1.07)
            if rec is not None:
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
            be.csd_allreduce_(acc)
            if rec is not None:
                c1.record()
                rec["coll"].append((c0, c1, F * (C * (C + 1) // 2) * 8))
            coh = be.coh_from_accumulator(acc, 1.0 / (K * T * world), "abs")
            done[slot] = torch.cuda.Event()
            done[slot].record(comm_stream)
        return coh

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    fsel = sorted({0, 1, 2, F // 4, F // 2, F - 3, F - 2, F - 1})

    def selfcheck_of(coh, reference_transforms=False):
        """Parity at the depth that was timed (K * T rows): raw accumulator and coherence of a handful of frequencies
        against complex128 products of the spectra of the LAST step (still in `spec`), and trial 0's transform against a
        float64 taper + rfft of the detrended float32 trial (mtmfft.py:96-127); criterion of tests/parity.py.  A
        collective on every rank (the reference sum needs every rank's spectra); the figures are rank 0's."""
        acc = accs[(nstep[0] - 1) % len(accs)]
        spec3 = spec.reshape(-1, F, C)
        tril = torch.tril(torch.ones(C, C, dtype=torch.bool, device="cuda"))
        worst_csd = worst_coh = 0.0
        for f in fsel:
            x = spec3[:, f, :].to(torch.complex128)
            ref = x.T @ x.conj()
            if dist_on:
                dist.all_reduce(torch.view_as_real(ref))          # the accumulator holds the sum over ranks
            got = acc[f].to(torch.complex128)
            tol = 1e-5 * ref.abs() + 1e-6 * ref.abs().max()
            worst_csd = max(worst_csd, float((((got - ref).abs() / tol)[tril]).max()))
            d = torch.sqrt(torch.diagonal(ref).real)
            cref = ref.abs() / torch.outer(d, d)
            tolc = 1e-5 * cref + 1e-6 * cref.max()
            worst_coh = max(worst_coh, float(((coh[f].to(torch.float64) - cref).abs() / tolc).max()))
        x0 = data[:N].to(torch.float64)
        x0 = x0 - x0.mean(0, keepdim=True)
        ref = torch.fft.rfft(x0[None] * torch.from_numpy(tapers).cuda()[:, :, None], dim=1) * scale      # (K, F, C)
        got = spec3[:K].to(torch.complex128)
        tol = 1e-5 * ref.abs() + 1e-6 * ref.abs().max()
        worst_fft = float(((got - ref).abs() / tol).max())
        return {"max_err_over_tol": max(worst_csd, worst_coh, worst_fft), "csd": worst_csd, "coherence": worst_coh,
                "fft": worst_fft, "rows": int(K * T * world)}

    for _ in range(args.warmup):
        step(None)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        coh = step(rec_main)
    fence()
    el = time.perf_counter() - t0
    tmax = torch.tensor([el], device="cuda", dtype=torch.float64)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax.item())

    # one more analysis, fenced on both sides: the un-pipelined latency of a single analysis (the timed steps above
    # overlap the collective / coherence tail of step i with the transforms of step i + 1 when N > 1)
    fence()
    t1 = time.perf_counter()
    coh = step(None)
    fence()
    latency = time.perf_counter() - t1

    can_check = not blocked and B == T
    selfcheck = selfcheck_of(coh) if can_check else None
    if rank == 0 and selfcheck:
        assert selfcheck["max_err_over_tol"] <= 1.0, selfcheck
    fallbacks = be.csd_split_fallbacks() if (C == 256 and not blocked) else None

    # ---- the twins of the headline (N = 1 only; never `value`): the same step with
    #   (a) the cross-spectral products on the FLOAT32 matrix instructions (csd3m_kernel): float32 operands as the reference's
    #       complex64 products have them (csd.py:98), float32 transforms;
    #   (b) float64 taper product + transform rounded to complex64 where mtmfft.py:104-127 rounds, AND float32 products:
    #       the reference's arithmetic end to end ("reference arithmetic");
    #   (c) float64 transforms with K4h (what precision="reference" runs by default).
    twins = {}
    if not blocked and not dist_on:
        plan64 = be.FFTPlan(N, N, C, tapers, scale, detrend=0, demean_taper=False, freq_idx=None, output="fourier",
                            keeptapers=True, reference_mean=refmean)
        has64 = plan64.set_precision(True)
        ntw = max(2, min(args.steps, 5))
        for key, fp, split in (("f32_products", plan, False), ("reference_arithmetic", plan64 if has64 else None, False),
                               ("reference_transforms_k4h", plan64 if has64 else None, True)):
            if fp is None:
                continue
            step(None, fp, split)
            fence()
            rec = {"csd": [], "fft": [], "coll": []}
            tt = time.perf_counter()
            for _ in range(ntw):
                coh_t = step(rec, fp, split)
            fence()
            el_t = time.perf_counter() - tt
            twins[key] = {"value": T * ntw / el_t, "unit": "trials/s", "ms_per_step": 1e3 * el_t / ntw, "steps": ntw,
                          "fft_kernel": fp.kernel_name, "csd_kernel": be.csd_kernel_name(C, blocked) if split else "spycsd::csd3m_kernel<256, 8> (float32 operands)",
                          "fft_ms_per_step": sum(a.elapsed_time(b) for a, b, _ in rec["fft"]) / ntw,
                          "csd_ms_per_step": sum(a.elapsed_time(b) for a, b, _ in rec["csd"]) / ntw,
                          "selfcheck": selfcheck_of(coh_t) if can_check else None,
                          "max_abs_diff_to_headline_coherence": float((coh_t - coh).abs().max())}
            if rank == 0 and twins[key]["selfcheck"]:
                assert twins[key]["selfcheck"]["max_err_over_tol"] <= 1.0, (key, twins[key]["selfcheck"])
            del coh_t
        del plan64

    if rank == 0:
        assert bool(torch.isfinite(coh).all()), "non-finite coherence"
        diag = coh[:, torch.arange(C), torch.arange(C)]
        assert float((diag - 1).abs().max()) < 1e-5, "coherence diagonal must be 1"
        ev_csd, ev_fft, ev_coll = rec_main["csd"], rec_main["fft"], rec_main["coll"]
        csd_ms = [a.elapsed_time(b) for a, b, _ in ev_csd]
        fft_ms = [a.elapsed_time(b) for a, b, _ in ev_fft]
        rows = [nb * K for _, _, nb in ev_csd]
        flops = [8.0 * r * F * C * (C + 1) / 2 for r in rows]                 # Hermitian-minimal, SURVEY 8d
        achieved = sum(flops) / (sum(csd_ms) * 1e-3) / 1e12
        fft_bytes = sum(nb * (N * C * 4 + K * F * C * 8) for _, _, nb in ev_fft)
        # matrix flops the 3-multiplication kernel really issues: 136 sub-tiles x 3 MFMAs of 16x16x4 per 4 rows
        # (every multiple of 16 up to 256 channels: floor(256 / C) frequencies per workgroup, nb (nb + 1) / 2 sub-tiles each)
        is3m = (C == 256 or not blocked)
        nsub = ((C + 15) // 16) * ((C + 15) // 16 + 1) // 2
        if C > 512:       # 256-channel blocks: Hermitian product per block, a 256-sub-tile rectangle per pair of blocks
            blocks = [min(256, C - 256 * i) for i in range((C + 255) // 256)]
            nsub = sum(((b + 15) // 16) * ((b + 15) // 16 + 1) // 2 for b in blocks) + 256 * (len(blocks) * (len(blocks) - 1) // 2)
        executed = ((rows[0] + 3) // 4) * F * nsub * 3 * 2048.0 if is3m else None
        # K4h (256 channels, standard layout): 12 v_mfma_f32_16x16x32_f16 (16384 flop) per 16 x 16 sub-tile and chunk of 32
        # rows on the frequencies of the full rounds of workgroups; the frequencies beyond them run the float32 tail
        k4h = C == 256 and not blocked and not os.environ.get("SPYHIP_CSD_F32")
        if k4h:
            ncu = torch.cuda.get_device_properties(0).multi_processor_count
            rem = F % ncu
            f_main = F - rem if (F > ncu and 0 < rem and 4 * rem <= ncu) else F
            executed = ((rows[0] + 31) // 32) * f_main * 136 * 12 * 16384.0
        value = world * T * args.steps / el
        coll = {"executed": bool(dist_on), "backend": None}
        if dist_on:
            coll["backend"] = ("RCCL through torch.distributed (the library's communicator could not be created on every rank)"
                               if getattr(be, "_lib_comm_failed", None) else "RCCL, library communicator (spyhip_allreduce_csd)")
        if ev_coll:
            coll.update({"bytes": ev_coll[0][2], "pack_allreduce_unpack_ms": float(np.mean([a.elapsed_time(b) for a, b, _ in ev_coll]))})
        traffic, traffic_prov = pmc_traffic(rows[0], F, C)
        avg_ms = float(np.mean(csd_ms))
        hbm_bytes = rows[0] * F * C * 8 + (2 * F * nsub * 256 * 8 if (is3m or k4h) else 2 * F * (((C + 31) // 32) * ((C + 31) // 32 + 1) // 2) * 1024 * 8)
        src = (traffic_prov.get("from_profile") or "none") + ((" (" + traffic_prov["refused"] + ")") if traffic_prov.get("refused") else "")
        if k4h:
            ex_tf = executed / (avg_ms * 1e-3) / 1e12
            roofline = {
                "bound": "mfma", "kernel": "spycsd::csdh_kernel", "achieved": ex_tf, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": ex_tf / PEAK_MFMA_F16_TFLOPS, "avg_launch_ms": avg_ms,
                "executed_mfma_flop_per_launch": executed, "algorithmic_flop_per_launch": flops[0],
                "algorithmic_TFLOPs": achieved, "fp32_equivalent_frac": achieved / PEAK_MFMA_F32_TFLOPS,
                "algorithmic_hbm_bytes_per_launch": hbm_bytes, "hbm_GBps": hbm_bytes / (avg_ms * 1e-3) / 1e9,
                "traffic": traffic, "traffic_source": src, "frequencies_left_to_float32_kernels": fallbacks,
            }
            roofline_notes = {
                "kernel": "spycsd::csdh_kernel (+ its float32 stand-in on flagged frequencies, the row-split <1, 1> tail and its reduction)",
                "note": "achieved = executed fp16 matrix flops per launch / avg_launch_ms: ceil(rows / 32) chunks x frequencies of "
                        "the full rounds x 136 sub-tiles x 12 v_mfma_f32_16x16x32_f16 x 16384 flop (= SQ_INSTS_VALU_MFMA_F16 / "
                        "SQ_INSTS_MFMA x 16384 of profiles/%s_pmc_headline.txt); peak = dense FP16 / BF16 MFMA.  The chip does not "
                        "hold 2.4 GHz under this load: GRBM_GUI_ACTIVE / duration in the same file gives the clock the fraction "
                        "should also be read against.  algorithmic_TFLOPs / fp32_equivalent_frac: the same launch in the units of "
                        "SURVEY 8(d) (8 flop per complex multiply-accumulate on the Hermitian-minimal triangle) against the float32 "
                        "matrix peak the survey priced K4 with." % PROFILE_ROUND,
                "kernel_stats": "profiles/%s_bench_final_kernel_stats.csv (Name = spycsd::csdh_kernel; AverageNs + the tail rows must agree with avg_launch_ms)" % PROFILE_ROUND,
                "fp32_peak": PEAK_MFMA_F32_TFLOPS, "hbm_frac": hbm_bytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic_source": traffic_prov,
            }
        else:
            ex = ((executed / flops[0]) if executed else 1.0) * achieved
            roofline = {
                "bound": "mfma", "kernel": be.csd_kernel_name(C, blocked), "achieved": ex, "peak": PEAK_MFMA_F32_TFLOPS,
                "unit": "TFLOP/s", "frac": ex / PEAK_MFMA_F32_TFLOPS, "avg_launch_ms": avg_ms,
                "executed_mfma_flop_per_launch": executed, "algorithmic_flop_per_launch": flops[0],
                "algorithmic_TFLOPs": achieved, "algorithmic_frac": achieved / PEAK_MFMA_F32_TFLOPS,
                "algorithmic_hbm_bytes_per_launch": hbm_bytes, "traffic": traffic, "traffic_source": src,
            }
            roofline_notes = {
                "note": "achieved = flops the kernel EXECUTES per launch (ceil(rows / 4) x F x sub-tiles x 3 MFMAs x 2048 flop) / "
                        "avg_launch_ms; algorithmic_* = SURVEY 8(d)'s credited flops (8 per complex multiply-accumulate on the "
                        "Hermitian-minimal triangle)", "traffic_source": traffic_prov}
        # the whole step against the bytes SURVEY 8(d) counts for it (input once, accumulator once) and against what the
        # two-kernel formulation moves (spectra written by K1 and read by K4)
        step_ms = 1e3 * el / args.steps
        alg_step = T * N * C * 4 + F * C * C * 8
        form_step = T * (N * C * 4 + 2 * K * F * C * 8) + 2 * F * C * C * 8
        step_info = {"fft_ms": sum(fft_ms) / args.steps, "csd_ms": sum(csd_ms) / args.steps,
                     "fft_stream_GBps": fft_bytes / (sum(fft_ms) * 1e-3) / 1e9,
                     "algorithmic_GB": alg_step / 1e9, "formulation_GB": form_step / 1e9,
                     "hbm_frac_of_formulation": form_step / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
        if coll.get("pack_allreduce_unpack_ms"):
            step_info["allreduce_ms"] = coll["pack_allreduce_unpack_ms"]
            step_info["allreduce_GB"] = coll["bytes"] / 1e9
        line = {
            "metric": "trials/sec for mtmfft+coherence (256 ch x 4096 samples, 7 DPSS tapers, full CSD)",
            "value": value,
            "unit": "trials/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 transforms; CSD operands fp16 hi+lo (22-bit), f32 accumulate" if k4h else "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: connectivityanalysis method='coh', AR(2) AnalogData, "
                            f"{C} ch x {N} samp x {T} trials per GPU, tapsmofrq=1 Hz ({K} DPSS tapers), polyremoval=0, "
                            "output='abs', inputs resident in HBM",
                "trials_per_gpu": T, "channels": C, "samples": N, "tapers": K, "freqs": F, "batch": B,
                "fft_kernel": plan.kernel_name + (" + seq_mean_kernel" if refmean else ""),
                "collective": {"executed": bool(dist_on), "bytes": coll.get("bytes"), "ms": coll.get("pack_allreduce_unpack_ms")},
            },
            "roofline": roofline,
            "step": step_info,
            "analysis_latency_ms": 1e3 * latency,
            "selfcheck": selfcheck,
        }
        detail = {"line": None, "roofline_notes": roofline_notes, "collective": coll, "twins": twins,
                  "selfcheck_criterion": "|a-b| <= 1e-5 |b| + 1e-6 max|b| against complex128 / float64 references of the same inputs; frequencies %s" % fsel,
                  "headline_frac_vs_survey_bound": value / (world * HEADLINE_BOUND_TRIALS_PER_S),
                  "channel_samples_per_s": value * N * C,
                  "dtype_note": "float32 transforms (the reference's float64 ones: twins.reference_*); the cross-spectral products take "
                                "each float32 operand as an exact-to-22-bits fp16 (hi, lo) pair on the fp16 matrix cores with float32 "
                                "accumulation (hi hi' + hi lo' + lo hi'; dropped term <= 2^-22), frequencies whose range a pair cannot "
                                "hold are redone by the float32 kernel; `selfcheck` = this launch against complex128 under the parity criterion"}
        if twins:
            # the same step at the reference's operand precision, beside `value` (never instead of it)
            if "f32_products" in twins:
                line["value_f32_products"] = twins["f32_products"]["value"]
            if "reference_arithmetic" in twins:
                line["value_reference_arithmetic"] = twins["reference_arithmetic"]["value"]
            line["selfcheck_twins"] = {k: (v["selfcheck"] or {}).get("max_err_over_tol") for k, v in twins.items()}
        if world == 1 and not args.no_secondary:
            del spec
            sec = secondary(torch, be, synthdata, data, N, C, T, refmean)
            detail["secondary"] = sec
            line["secondary"] = compact_secondary(sec)
            if "reference_transforms_k4h" in twins:
                line["secondary"]["ref_transforms_k4h"] = {"trials_per_s": twins["reference_transforms_k4h"]["value"]}
            fe = line["secondary"].get("front_end")
            if fe:
                line["value_with_host_copy"] = fe["with_host_copy"]
                line["value_front_end"] = fe["warm"]
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(C, N)
            detail["cpu_baseline"] = cb
            line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                    "sample": "1 trial of %d ch x %d samp, cross_spectra_cF as csd.py:94-102, one core" % (C, N),
                                    "cpu_model": cb["cpu_model"], "best_cpu_trials_per_s": cb["best_cpu_trials_per_s"],
                                    "best_cpu_cores": cb["best_cpu_cores"]}
        line = _short(line, 5)
        detail["line"] = dict(line)
        emit(line, detail)
    if dist_on:
        torch.cuda.synchronize()
        be.shutdown_library_comm()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
