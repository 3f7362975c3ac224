"""spy.connectivityanalysis(method='coh') on 256 ch x 4096 samples x 1000 resident trials: per-call times with the result read
on the host, left in HBM (main stream synchronised), and in chains without synchronisation - with the frequency-range
pipeline of the host copy (backend.coh_pipeline) and without it.  python tools/frontend_chain_probe.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import syncopy_amd as spy
from syncopy_amd import backend as be, synthdata

C, N, T = 256, 4096, 1000
data = synthdata.ar2_uncoupled_fast(C, N, T, seed=1234)
host = data.cpu().numpy()
del data
trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
adata = spy.AnalogData(host, samplerate=1000.0, trialdefinition=trl)
main = torch.cuda.current_stream()
keep = be.frequency_ranges


def call():
    return spy.connectivityanalysis(adata, method="coh", tapsmofrq=1, polyremoval=0)


for label, fr in (("pipeline", keep), ("one piece", lambda *a, **k: None), ("pipeline", keep), ("one piece", lambda *a, **k: None)):
    be.frequency_ranges = fr
    for _ in range(3):
        r = call(); _ = r.data.shape; del r
    torch.cuda.synchronize()
    t_read, t_hbm = [], []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = call(); main.synchronize(); t_hbm.append(time.perf_counter() - t0)
        _ = r.data.shape; t_read.append(time.perf_counter() - t0); del r
    chains = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            r = call(); del r
        torch.cuda.synchronize(); chains.append((time.perf_counter() - t0) / 8)
    reads = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            r = call(); _ = r.data.shape; del r
        reads.append((time.perf_counter() - t0) / 8)
    print("%-10s result in HBM %.2f ms   read on the host %.2f ms   chain of 8 unread %.2f ms/call (%s)   chain of 8 read %.2f ms/call"
          % (label, 1e3 * min(t_hbm), 1e3 * min(t_read), 1e3 * min(chains), " ".join("%.1f" % (1e3 * c) for c in chains), 1e3 * min(reads)), flush=True)
be.frequency_ranges = keep
