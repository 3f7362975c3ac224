"""Granger case of tests/test_gpu_fuzz.py (family SPY_FUZZ_OFFSET, seed argv[1]): kernels against the oracle at the bins next to
DC, both at the default and at a tighter stopping tolerance.  Development aid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import syncopy_amd as spy
from oracle_routines import ORACLE_CONN

seed = int(sys.argv[1])
OFFSET = int(os.environ.get("SPY_FUZZ_OFFSET", "0"))
rng = np.random.default_rng(7000 + OFFSET + seed)
nchan = int(rng.integers(2, 7))
adj = np.zeros((nchan, nchan))
for _ in range(int(rng.integers(1, nchan + 1))):
    i, j = rng.choice(nchan, size=2, replace=False)
    adj[i, j] = float(rng.uniform(0.1, 0.3))
data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=int(rng.choice([500, 1000, 1024])), nTrials=int(rng.integers(30, 60)),
                                 seed=int(rng.integers(1, 10000)))
kw = dict(method="granger", tapsmofrq=float(rng.choice([3.0, 5.0])))
if rng.integers(0, 2):
    kw["pad"] = "nextpow2"
print("channels", nchan, "adj", adj.tolist(), kw, "trials", len(data.trials), "samples", data.trials[0].shape[0])
got = spy.connectivityanalysis(data, **kw)
ref = spy.connectivityanalysis(data, **kw, compute_method="sequential", routine_classes=ORACLE_CONN)
print("info got", dict(got.info), "\ninfo ref", dict(ref.info))
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for f in (0, 1, 2, 3, 10):
    print("f =", f, "\n got\n", got.data[0, f], "\n ref\n", ref.data[0, f])
# the cross-spectral matrix both sides factorise (oracle's), through the oracle and the kernels at several tolerances
csd = spy.connectivityanalysis(data, method="csd", tapsmofrq=kw["tapsmofrq"], pad=kw.get("pad", "maxperlen"), compute_method="sequential",
                               routine_classes=ORACLE_CONN)
print("csd (not demean_taper) shape", csd.data.shape)
