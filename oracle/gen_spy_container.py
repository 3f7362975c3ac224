# -*- coding: utf-8 -*-
"""
Fixture generator for the `.spy` container tests: the REAL reference writes a small container with
`spy.save` - an AnalogData object (3 trials, trigger offsets, named channels), the same recording with the time axis
second (dimord ["channel", "time"]), and the reference's own freqanalysis / connectivityanalysis results on it
(power spectrum, coherence, complex cross-spectral density).  tests/test_spy_container.py reads these files with the
package's own HDF5 reader, and the GPU tests recompute the results from the loaded recording.

Run through ``oracle/make_golden.sh --container`` (same interpreter and stand-in modules as gen_golden.py).
With ``--check <dir>`` it instead LOADS a container written by syncopy_amd.io.save with the unmodified reference
(checksum on, mode r+) and recomputes a spectrum from it: the egress direction, build container only.

TEST INFRASTRUCTURE ONLY - see oracle/spy_oracle.py header.
"""
import os
import sys

import numpy as np

import syncopy as spy


def generate(out):
    rng = np.random.default_rng(0)
    t = np.arange(600) / 500.0
    x = rng.standard_normal((600, 5)).astype(np.float32)
    x[:, 1] += np.sin(2 * np.pi * 40 * t).astype(np.float32)
    x[:, 2] += 0.5 * x[:, 1] + 3.0
    trl = np.array([[0, 200, -50], [200, 400, -50], [400, 600, -50]])
    chans = ["a", "b", "c", "d", "e"]
    d = spy.AnalogData(data=x, samplerate=500.0, trialdefinition=trl, channel=chans)
    spy.save(d, container=os.path.join(out, "session.spy"), overwrite=True)
    dT = spy.AnalogData(data=np.ascontiguousarray(x.T), samplerate=500.0, trialdefinition=trl, channel=chans,
                        dimord=["channel", "time"])
    spy.save(dT, container=os.path.join(out, "session.spy"), tag="chantime", overwrite=True)
    s = spy.freqanalysis(d, method="mtmfft", tapsmofrq=5)
    spy.save(s, container=os.path.join(out, "session.spy"), tag="pow", overwrite=True)
    c = spy.connectivityanalysis(d, method="coh", tapsmofrq=5)
    spy.save(c, container=os.path.join(out, "session.spy"), tag="coh", overwrite=True)
    q = spy.connectivityanalysis(d, method="csd", tapsmofrq=5)
    spy.save(q, container=os.path.join(out, "session.spy"), tag="csd", overwrite=True)
    sel = spy.freqanalysis(d, method="mtmfft", tapsmofrq=5, select={"trials": [0, 2], "channel": ["b", "d"],
                                                                      "latency": [-0.05, 0.2]})
    spy.save(sel, container=os.path.join(out, "session.spy"), tag="powsel", overwrite=True)
    print(sorted(os.listdir(os.path.join(out, "session.spy"))))


def check(path):
    objs = spy.load(path, checksum=True)
    for name, o in sorted(objs.items()):
        print(name, type(o).__name__, o.data.shape, o.data.dtype, o.dimord)
    analog = [o for o in objs.values() if type(o).__name__ == "AnalogData"]
    spec = [o for n, o in objs.items() if n.endswith("_pow.spectral")]
    if analog and spec:
        again = spy.freqanalysis(analog[0], **{k: v for k, v in spec[0].cfg["freqanalysis"].items()
                                               if k in ("method", "tapsmofrq", "output", "keeptrials")})
        err = np.max(np.abs(again.data[()] - spec[0].data[()])) / np.max(np.abs(again.data[()]))
        print(f"reference spectrum of the loaded recording vs the stored result: max rel. deviation {err:.2e}")
        assert err < 1e-5
    print("CHECK OK")


if __name__ == "__main__":
    if sys.argv[1] == "--check":
        check(sys.argv[2])
    else:
        generate(sys.argv[1])
