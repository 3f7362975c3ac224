"""The RCCL leg on hardware (VERDICT r1 "what's weak" 2): compute_hip's csd_tril_pack -> all_reduce -> unpack and
bench.py's distributed branch executed under an initialised "nccl" process group (one rank per visible GPU; on the
1-GPU box a group of one rank with the collectives forced on) and compared with the group-less result."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import syncopy_amd as spy

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, nproc, extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)


def _ngpu():
    import torch
    from syncopy_amd import backend
    backend.require_gpu()
    return torch.cuda.device_count()


def test_front_ends_under_nccl_group(tmp_path):
    n = min(_ngpu(), 2)
    out = tmp_path / "nccl.npz"
    r = _launch([os.path.join(ROOT, "tests", "nccl_worker.py"), str(out)], n, {"SPY_FORCE_COLLECTIVE": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    assert int(z["world"]) == n and bool(z["pack_allreduce_unpack_bit_exact"])
    assert bool(z["library_comm_up"]) and bool(z["library_allreduce_bit_exact"])
    assert bool(z["fallback_taken"]) and bool(z["fallback_allreduce_bit_exact"])     # a communicator that cannot be created
    # every rank staged the rows of its own trial shard and nothing else: disjoint spans in rank order, together the
    # recording; the bytes over PCIe add up to ONE copy of it (not one per rank)
    st = z["staged_rows_and_bytes_per_rank"]
    assert st.shape == (n, 3) and st[0, 0] == 0 and st[-1, 1] == int(z["recording_rows"])
    assert all(st[r, 1] == st[r + 1, 0] for r in range(n - 1))
    assert int(st[:, 2].sum()) == int(z["recording_rows"]) * 37 * 4
    adj = np.zeros((37, 37))
    adj[0, 1] = adj[5, 30] = 0.3
    data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=1024, nTrials=11, seed=3, samplerate=500)
    ref = {"coh": spy.connectivityanalysis(data, method="coh", tapsmofrq=3).data,
           "csd": spy.connectivityanalysis(data, method="csd", tapsmofrq=3).data,
           "ppc": spy.connectivityanalysis(data, method="ppc", tapsmofrq=3).data,
           "corr": spy.connectivityanalysis(data, method="corr").data,
           "pow_avg": spy.freqanalysis(data, method="mtmfft", tapsmofrq=3, keeptrials=False).data,
           "pow": spy.freqanalysis(data, method="mtmfft", tapsmofrq=3).data}
    for k, v in ref.items():
        if n == 1:
            # one rank: the all-reduce of one contribution is the identity - the collective path must not change a bit
            assert np.array_equal(z[k], v), k
        else:
            from parity import assert_parity
            assert_parity(z[k], v, what=k)
    assert np.array_equal(z["pow"], ref["pow"])           # stacked trials: bit-identical for any number of ranks
    # Granger: the group-less run is ONE spyhip_granger call, the group run the stepped, frequency-sharded sequence
    gr = spy.connectivityanalysis(data, method="granger", tapsmofrq=3)
    assert bool(z["granger_info"][0]) and gr.info["converged"]
    assert z["granger_info"][2] == gr.info["reg. factor"]
    np.testing.assert_allclose(z["granger_info"][3], gr.info["initial cond. num"], rtol=1e-6)
    np.testing.assert_allclose(z["granger"], gr.data, rtol=1e-4, atol=1e-6)


def test_c5_granger_sharded_under_nccl_group(tmp_path):
    """BASELINE configs[4] as one front-end call under the process group (trial shards + RCCL all-reduce of the CSD +
    frequency-sharded Wilson): two ranks whenever two GPUs are visible, else a group of one with the collectives forced
    on; compared with the group-less call on the same data."""
    from test_gpu_production import c5_dataset
    n = min(_ngpu(), 2)
    T = 2560
    out = tmp_path / "c5.npz"
    r = _launch([os.path.join(ROOT, "tests", "nccl_worker.py"), str(out)], n, {"SPY_FORCE_COLLECTIVE": "1", "SPY_NCCL_C5": str(T)})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    assert int(z["world"]) == n and bool(z["c5_info"][0]) and z["c5_info"][1] < 5e-6 and z["c5_info"][2] == 0
    ref = spy.connectivityanalysis(c5_dataset(T), method="granger", tapsmofrq=1)
    assert ref.info["converged"]
    np.testing.assert_allclose(z["c5_info"][3], ref.info["initial cond. num"], rtol=1e-5)
    np.testing.assert_allclose(z["c5_sample"], ref.data[0, ::64, :16, :16], rtol=1e-4, atol=1e-6)
    spy.release_device_buffers()


def test_bench_distributed_branch():
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, nccl), on the GPUs present."""
    n = min(_ngpu(), 2)
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--trials", "60",
                 "--no-cpu-baseline", "--no-secondary"], n, {"SPY_BENCH_FORCE_DIST": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["config"]["collective"]["executed"] is True
