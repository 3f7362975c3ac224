"""cProfile of the warm spy.connectivityanalysis(method='coh') call on the headline shape (development aid)."""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import syncopy_amd as spy
from syncopy_amd import synthdata

C, N, T = 256, 4096, 1000
host = synthdata.ar2_uncoupled_fast(C, N, T, seed=5).cpu().numpy()
trl = np.stack([np.arange(T) * N, np.arange(1, T + 1) * N, np.zeros(T)], axis=1)
data = spy.AnalogData(host, samplerate=1000.0, trialdefinition=trl)
for i in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = spy.connectivityanalysis(data, method="coh", tapsmofrq=1, polyremoval=0)
    t1 = time.perf_counter()
    shape = res.data.shape
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"call {i}: returned after {1e3 * (t1 - t0):.1f} ms, result on the host after {1e3 * (t2 - t0):.1f} ms")
pr = cProfile.Profile()
pr.enable()
res = spy.connectivityanalysis(data, method="coh", tapsmofrq=1, polyremoval=0)
shape = res.data.shape
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# the call alone (result left in HBM): where the host time goes, by own time
pr = cProfile.Profile()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr.enable()
res = spy.connectivityanalysis(data, method="coh", tapsmofrq=1, polyremoval=0)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"profiled call: returned after {1e3 * (t1 - t0):.1f} ms, device idle after {1e3 * (t2 - t0):.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
