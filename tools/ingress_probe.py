"""Where the first call on host-resident data spends its time (development aid): the upload alone for several staging
configurations, the taper tables, the whole front-end call.  python tools/ingress_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import syncopy_amd as spy  # noqa: E402
from syncopy_amd import backend as be  # noqa: E402

C, N, T = 256, 4096, 1000
rng = np.random.default_rng(0)
host = rng.standard_normal((T * N, C), dtype=np.float32)
torch.cuda.init()
torch.zeros(1, device="cuda")
for threads in (4, 8, 16):
    for chunk_mb in (64, 128, 256):
        be._COPY_THREADS, be._H2D_CHUNK, be._copy_pool = threads, chunk_mb << 20, None
        be._pin.pop("h2d", None)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = be.to_device(host, torch.device("cuda", 0))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            del d
        print(f"to_device {threads:2d} threads, {chunk_mb:3d} MiB chunks: first {ts[0] * 1e3:6.1f} ms, best {min(ts) * 1e3:6.1f} ms = "
              f"{host.nbytes / min(ts) / 1e9:5.1f} GB/s", flush=True)
be._COPY_THREADS, be._H2D_CHUNK, be._copy_pool = 8, 128 << 20, None
be._pin.pop("h2d", None)
pin = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
print(f"pinned -> device, 4 x 1 GiB: {4 * (1 << 30) / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
t0 = time.perf_counter()
from syncopy_amd.specest.tapers import taper_table
tp = taper_table("dpss", N, N, {"NW": 4.096, "Kmax": 7})
print(f"DPSS taper table: {1e3 * (time.perf_counter() - t0):.1f} ms")
trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
for rep in range(2):
    adata = spy.AnalogData(host, samplerate=1000.0, trialdefinition=trl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = spy.connectivityanalysis(adata, method="coh", tapsmofrq=1, polyremoval=0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res2 = spy.connectivityanalysis(adata, method="coh", tapsmofrq=1, polyremoval=0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"front end, fresh AnalogData #{rep}: first call {1e3 * (t1 - t0):.1f} ms, second {1e3 * (t2 - t1):.1f} ms")
    del adata, res, res2
x = spy.synthdata.ar2_uncoupled_fast(C, N, 400, seed=3)
acc = torch.zeros((N // 2 + 1, C, C), dtype=torch.complex64, device="cuda")
plan = be.FFTPlan(N, N, C, tp, np.sqrt(2) / N, 0, True, None, "fourier", True, reference_mean=True)
st = torch.arange(400, device="cuda", dtype=torch.int64) * N
be.csd_accumulate(plan.execute(x, st), acc)
be.csd_finalize(acc, 1.0 / 2800)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    be.granger(acc)
    torch.cuda.synchronize()
    print(f"granger call {rep}: {time.perf_counter() - t0:.3f} s")
