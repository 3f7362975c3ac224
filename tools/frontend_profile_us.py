"""Own time per function of the warm spy.connectivityanalysis(method='coh') call in microseconds (development aid)."""
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
for i in range(4):
    res = spy.connectivityanalysis(data, method="coh", tapsmofrq=1, polyremoval=0)
    torch.cuda.synchronize()
    del res
pr = cProfile.Profile()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr.enable()
res = spy.connectivityanalysis(data, method="coh", tapsmofrq=1, polyremoval=0)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"call returned after {1e3 * (t1 - t0):.2f} ms, device idle after {1e3 * (t2 - t0):.2f} ms")
st = pstats.Stats(pr).stats
rows = sorted(((v[2], v[3], v[0], k) for k, v in st.items()), reverse=True)[:32]
for tt, ct, nc, k in rows:
    print(f"{1e6 * tt:8.0f} us own {1e6 * ct:8.0f} us cum {nc:6d} calls  {k[0].split('/')[-1]}:{k[1]} {k[2]}")
