"""`.spy` containers through the hot path on the GPU: a recording the REAL reference saved is loaded
(memory-mapped), staged into the in-HBM trial queue and analysed; the results are compared with the results the
reference itself stored next to it (tests/golden/spy_container, oracle/gen_spy_container.py) and written back."""
import os

import numpy as np
import pytest

import syncopy_amd as spy
from parity import assert_parity

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CONT = os.path.join(HERE, "golden", "spy_container", "session.spy")


@pytest.mark.parametrize("tag", [None, "chantime"])
def test_loaded_recording_reproduces_the_stored_reference_results(tag, tmp_path):
    data = spy.load(os.path.join(CONT, "session.analog" if tag is None else "session_chantime.analog"))
    ref = spy.load(CONT, dataclass=["spectral", "crossspectral"])
    pw = spy.freqanalysis(data, method="mtmfft", tapsmofrq=5)
    assert_parity(pw.data, np.array(ref["session_pow.spectral"].data), what="pow")
    assert list(pw.channel) == list(ref["session_pow.spectral"].channel)
    assert np.array_equal(pw.trialdefinition, ref["session_pow.spectral"].trialdefinition)
    assert np.allclose(pw.freq, ref["session_pow.spectral"].freq)
    sel = spy.freqanalysis(data, method="mtmfft", tapsmofrq=5,
                           select={"trials": [0, 2], "channel": ["b", "d"], "latency": [-0.05, 0.2]})
    assert_parity(sel.data, np.array(ref["session_powsel.spectral"].data), what="pow with selection")
    assert list(sel.channel) == ["b", "d"]
    coh = spy.connectivityanalysis(data, method="coh", tapsmofrq=5)
    assert_parity(coh.data, np.array(ref["session_coh.crossspectral"].data), rtol=2e-5, atol_rel=2e-6, what="coh")
    csd = spy.connectivityanalysis(data, method="csd", tapsmofrq=5)
    assert_parity(csd.data, np.array(ref["session_csd.crossspectral"].data), what="csd")
    # egress: what the GPU computed goes back into a container and comes out unchanged
    cont = str(tmp_path / "out.spy")
    spy.save(pw, container=cont, tag="pow")
    spy.save(csd, container=cont, tag="csd")
    back = spy.load(cont, checksum=True)
    assert np.array_equal(back["out_pow.spectral"].data, pw.data)
    assert np.array_equal(back["out_csd.crossspectral"].data, csd.data)
    assert back["out_pow.spectral"].cfg["freqanalysis"]["method"] == "mtmfft"
    assert list(back["out_csd.crossspectral"].channel_i) == list(data.channel)


@pytest.mark.parametrize("layout", ["time_channel_f32", "channel_time_f32", "time_channel_f64"])
def test_memmap_staging_of_a_file_larger_than_the_pinned_buffers(layout, tmp_path):
    """to_device streams a mapped file through 2 x 256 MiB pinned buffers: 300 MB here, so the double-buffering, the
    last partial block, the float64 -> float32 conversion and the transposition of channel-major files all run."""
    import torch
    nchan, ntime = 96, 800_000 if "f32" in layout else 400_000
    rng = np.random.default_rng(1)
    x = rng.standard_normal((ntime, nchan), dtype=np.float32)
    host = x if layout.startswith("time") else np.ascontiguousarray(x.T)
    if layout.endswith("f64"):
        host = host.astype(np.float64)
    dimord = ["time", "channel"] if layout.startswith("time") else ["channel", "time"]
    trl = np.array([[k * 4000, (k + 1) * 4000, 0] for k in range(ntime // 4000)])
    spy.save(spy.AnalogData(host, samplerate=1000.0, trialdefinition=trl, dimord=dimord), filename=str(tmp_path / "big"))
    del host
    data = spy.load(str(tmp_path / "big.analog"))
    assert isinstance(data.data, np.memmap) and data.data.nbytes > (256 << 20)
    dev = data.device_data()
    assert dev.shape == (ntime, nchan) and dev.dtype == torch.float32
    assert torch.equal(dev.cpu(), torch.from_numpy(x))
    # and the analysis of a few trials of it equals the analysis of the same trials held in memory
    a = spy.freqanalysis(data, method="mtmfft", tapsmofrq=2, select={"trials": [0, 7, ntime // 4000 - 1]})
    mem = spy.AnalogData(x, samplerate=1000.0, trialdefinition=trl)
    b = spy.freqanalysis(mem, method="mtmfft", tapsmofrq=2, select={"trials": [0, 7, ntime // 4000 - 1]})
    assert np.array_equal(a.data, b.data)
