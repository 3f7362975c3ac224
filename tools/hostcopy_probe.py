import time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n = 512 << 20
src = torch.empty(n, dtype=torch.uint8, pin_memory=True).numpy()
src[:] = 1
pool = ThreadPoolExecutor(8)
def par(dst, src, k):
    step = (n + k - 1) // k
    fs = [pool.submit(np.copyto, dst[o:o + step], src[o:o + step]) for o in range(0, n, step)]
    [f.result() for f in fs]
for k in (1, 2, 4, 8):
    ts = []
    for rep in range(3):
        dst = np.empty(n, dtype=np.uint8)
        t0 = time.perf_counter(); par(dst, src, k); ts.append(time.perf_counter() - t0)
    dst2 = np.empty(n, dtype=np.uint8); dst2[:] = 0
    t0 = time.perf_counter(); par(dst2, src, k); tw = time.perf_counter() - t0
    print(f"{k} threads: fresh pages {n / min(ts) / 1e9:.1f} GB/s, touched pages {n / tw / 1e9:.1f} GB/s")
