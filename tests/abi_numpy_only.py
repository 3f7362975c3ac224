"""Run by tests/test_gpu_abi_numpy.py in a fresh interpreter: mtmfft + coherence of BASELINE config 1 through
libspyhip.so with NOTHING but NumPy/SciPy + ctypes in the process (PyTorch is never imported), checked against the
vectors the real reference produced (tests/golden/c1.npz).  Also: the library's own RCCL path (communicator of one
rank per visible GPU ... here one rank) and the trial queue's exact-indexing checks."""
import os
import sys

os.environ["SPY_NO_TORCH"] = "1"       # a NumPy-only host by declaration: syncopy_amd/_lib.py must not import torch

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from parity import assert_parity
    from syncopy_amd import abi, synthdata
    from syncopy_amd.specest.tapers import spec_scale, taper_table
    assert "torch" not in sys.modules

    z = np.load(os.path.join(ROOT, "tests", "golden", "c1.npz"))
    data = synthdata.ar2_network(AdjMat=np.zeros((16, 16)), nSamples=2000, nTrials=20, seed=42)
    x, si = np.asarray(data.data), data.sampleinfo
    assert np.array_equal(x[:2000], z["trial0"]) and np.array_equal(si, z["sampleinfo"].astype(np.int64))
    N = 2000
    tapers = taper_table("dpss", N, N, {"NW": 4.0, "Kmax": 7})          # tapsmofrq = 2 Hz at 1 kHz
    scale = spec_scale(N, N)

    dev = abi.Device(0)
    q = dev.upload_trials(x, si)
    pw = dev.mtmfft(q, tapers, scale, N, detrend=0, output="pow", keeptapers=False)
    assert pw.shape == (20, 1, 1001, 16) and pw.dtype == np.float32
    assert_parity(pw, z["pow"], what="c1 pow through the NumPy-only host")
    coh = dev.coherence(q, tapers, scale, N, detrend=0, output="abs")
    assert_parity(coh[None], z["coh_abs"], what="c1 coherence through the NumPy-only host")

    # sums over ranks inside the library (RCCL): communicator of ONE rank - the collective path runs and must be
    # the identity, bit for bit
    uid = dev.comm_unique_id()
    assert len(uid) == 128
    dev.comm_init(uid, 0, 1)
    assert dev.has_comm
    coh2 = dev.coherence(q, tapers, scale, N, detrend=0, output="abs")
    assert np.array_equal(coh, coh2)
    buf = abi.Buffer(dev, (1000,), np.float64)
    vals = np.random.default_rng(0).normal(size=1000)
    abi.check(dev.lib.spyhip_upload(dev.handle, buf.ptr, vals.ctypes.data, vals.nbytes), "upload")
    abi.check(dev.lib.spyhip_allreduce(dev.handle, buf.ptr, 1000, 1), "allreduce f64")
    dev.synchronize()
    assert np.array_equal(buf.numpy(), vals)
    dev.comm_destroy()
    assert not dev.has_comm

    # exact trial indexing: reordered, repeated, overlapping trials are the rows they name - nothing else
    si2 = np.array([[4000, 6000], [0, 2000], [0, 2000], [1000, 3000]], dtype=np.int64)
    q2 = dev.upload_trials(x, si2)
    p2 = dev.mtmfft(q2, tapers, scale, N, detrend=0, output="pow", keeptapers=False)
    assert np.array_equal(p2[0], pw[2]) and np.array_equal(p2[1], pw[0]) and np.array_equal(p2[2], pw[0])
    q1 = dev.upload_trials(x[1000:3000], np.array([[0, 2000]], dtype=np.int64))
    assert np.array_equal(dev.mtmfft(q1, tapers, scale, N, output="pow")[0], p2[3])
    try:
        dev.upload_trials(x, np.array([[39000, 41000]], dtype=np.int64))
    except abi.SpyHipError as exc:
        assert "outside" in str(exc)
    else:
        raise AssertionError("a trial beyond the data must be refused")
    assert "torch" not in sys.modules
    print("abi numpy-only ok")


if __name__ == "__main__":
    main()
