import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
os.environ["SPY_FUZZ_SCALE"] = "20"
import numpy as np, scipy.signal as sps
import syncopy_amd as spy
import test_gpu_fuzz as T
from oracle_routines import ORACLE_FREQ
seed = int(sys.argv[1]); off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(5000 + off + seed)
polyremoval = [None, 0, 1][int(rng.integers(0, 3))]
data, lengths = T._make(rng, ragged=False, offsets=polyremoval is not None)
n = lengths[0]
assert not rng.integers(0, 2)
kw = dict(method="superlet", order_max=int(rng.integers(2, 6)), order_min=1, c_1=int(rng.integers(1, 4)),
          adaptive=bool(rng.integers(0, 2)), foi=np.sort(rng.uniform(20, 300, size=int(rng.integers(2, 5)))),
          toi="all", output=str(rng.choice(["pow", "abs"])), keeptrials=True)
kw["polyremoval"] = polyremoval
print(kw, lengths, data.data.shape)
ref = spy.freqanalysis(data, compute_method="sequential", routine_classes=ORACLE_FREQ, **kw).data
keep = sps.fftconvolve
sps.fftconvolve = lambda a, b, mode="full", axes=None: keep(np.asarray(a, dtype=np.float64), b, mode=mode, axes=axes)
exact = spy.freqanalysis(data, compute_method="sequential", routine_classes=ORACLE_FREQ, **kw).data
sps.fftconvolve = keep
g32 = spy.freqanalysis(data, **kw).data
g64 = spy.freqanalysis(data, precision="reference", **kw).data
s32 = spy.freqanalysis(data, compute_method="sequential", **kw).data
s64 = spy.freqanalysis(data, compute_method="sequential", precision="reference", **kw).data
tol = 1e-5 * np.abs(ref) + 5e-6 * np.abs(ref).max()
for name, a in (("batched f32", g32), ("batched f64", g64), ("per-trial f32", s32), ("per-trial f64", s64), ("float64 oracle", exact)):
    r = np.abs(a.astype(np.float64) - ref) / tol
    i = np.unravel_index(r.argmax(), r.shape)
    print(f"{name:16s} max err/tol vs reference {r.max():.3g} at {tuple(int(v) for v in i)}: value {a[i]:.6g} ref {ref[i]:.6g} exact {exact[i]:.6g}; vs float64 oracle {float((np.abs(a.astype(np.float64) - exact) / tol).max()):.3g}")
print("identical batched/per-trial f32:", np.array_equal(g32, s32), " f64:", np.array_equal(g64, s64))
