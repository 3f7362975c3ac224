"""A rank stages only the rows of its own trial shard (AnalogData.shard_span / backend.to_device(rows=) / datatype.device_rows;
the reference's workers read only their slab, shared/kwarg_decorators.py:684-735).  One GPU: the rank / world size are
faked without a process group (collectives are no-ops then), so what is checked is the upload of a sub-span - plain copy
and the chunked background Upload - and the shard-local row arithmetic of the kernels' callers."""
import numpy as np
import pytest

import syncopy_amd as spy
from parity import assert_parity

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture
def as_rank_1_of_3(monkeypatch):
    from syncopy_amd import backend, parallel
    backend.require_gpu()
    monkeypatch.setattr(parallel, "world", lambda: (1, 3))
    return parallel


def test_partial_sums_from_a_staged_shard(as_rank_1_of_3):
    adj = np.zeros((16, 16))
    adj[0, 1] = 0.3
    data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=300, nTrials=7, seed=5, samplerate=200)
    got = spy.connectivityanalysis(data, method="csd", tapsmofrq=3).data          # rank 1: trials 3, 4 of 7, divided by 7
    assert data.staged_rows == (3 * 300, 5 * 300) and tuple(data._device.shape) == (600, 16)
    assert torch.equal(data._device.cpu(), torch.from_numpy(np.asarray(data.data[900:1500], dtype=np.float32)))
    pw = spy.freqanalysis(data, method="mtmfft", tapsmofrq=3, keeptrials=False).data
    as_rank_1_of_3.world = lambda: (0, 1)
    ref = spy.connectivityanalysis(data, method="csd", tapsmofrq=3, select={"trials": [3, 4]}).data
    assert data.staged_rows == (0, 2100)                                           # a new span: staged again, whole recording
    assert_parity(got, (ref * (2.0 / 7.0)).astype(np.complex64), what="CSD partial sum of the shard")
    refp = spy.freqanalysis(data, method="mtmfft", tapsmofrq=3, keeptrials=False, select={"trials": [3, 4]}).data
    assert_parity(pw, (refp * (2.0 / 7.0)).astype(np.float32), what="power partial sum of the shard")


def test_chunked_upload_of_a_span(as_rank_1_of_3):
    """Above 64 MB the ingress goes through backend.Upload (pinned staging, copy stream, marks per chunk): the shard's
    rows arrive bit for bit, wait_rows speaks shard-local rows, and the consumer behind the copy gets the right trials."""
    rng = np.random.default_rng(0)
    C, N, T = 256, 4096, 60                                                       # 252 MB; rank 1 of 3: trials 20 ... 39
    host = rng.standard_normal((T * N, C)).astype(np.float32)
    data = spy.AnalogData(data=host, samplerate=1000.0,
                          trialdefinition=np.stack([np.arange(T) * N, (np.arange(T) + 1) * N, np.zeros(T)], axis=1))
    dev = data.device_data(partial=True)
    up = data.upload_in_flight()
    assert data.staged_rows == (20 * N, 40 * N) and tuple(dev.shape) == (20 * N, C)
    if up is not None:
        assert up.row0 == 20 * N and up.nrows == 20 * N
        up.wait_rows(5 * N)
    torch.cuda.current_stream().synchronize()
    assert torch.equal(dev[:5 * N].cpu(), torch.from_numpy(host[20 * N:25 * N]))
    dev = data.device_data()
    assert torch.equal(dev.cpu(), torch.from_numpy(host[20 * N:40 * N]))
    from syncopy_amd.datatype import device_rows
    assert device_rows(data)[20] == (0, N) and device_rows(data)[39] == (19 * N, 20 * N)
    coh = spy.connectivityanalysis(data, method="csd", tapsmofrq=2)               # first analysis of fresh host data:
    as_rank_1_of_3.world = lambda: (0, 1)                                          # consumed behind the copy
    ref = spy.connectivityanalysis(data, method="csd", tapsmofrq=2, select={"trials": list(range(20, 40))})
    assert_parity(coh.data, (ref.data * (20.0 / 60.0)).astype(np.complex64), what="CSD of the shard behind its upload")
