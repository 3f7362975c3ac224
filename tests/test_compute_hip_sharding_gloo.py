"""compute_hip's OWN trial sharding under a process group, on the CPU: the product
classes (not the oracle-bound ones) run their real `compute_hip` in two and three gloo ranks with the device
primitives replaced by CPU stand-ins (FFT through the oracle, the accumulation as an einsum, the collective through
gloo), so the index math - which rows each rank transforms, which accumulator they land in, the ONE sum over ranks and
the division by the GLOBAL trial and taper count - is what is tested, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import syncopy_amd as spy
from oracle import spy_oracle as O
from parity import assert_parity


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    adj = np.zeros((4, 4))
    adj[0, 1] = 0.3
    return spy.synthdata.ar2_network(AdjMat=adj, nSamples=300, nTrials=7, seed=11, samplerate=200)


def _patch():
    """CPU stand-ins for the device primitives compute_hip touches."""
    from syncopy_amd import backend, parallel
    from syncopy_amd.datatype import AnalogData
    from syncopy_amd.specest import hip_spectral as hs
    calls = {"rows": [], "staged": []}

    def device_data(self, device=None, partial=False):
        # the product's own span / origin logic (AnalogData.shard_span, datatype.device_rows), a host tensor instead of
        # the upload: only this rank's rows exist in it, so every index below must be shard-local
        span = self.shard_span()
        self._row_origin, self.staged_rows = span[0], span
        calls["staged"].append(span)
        return torch.from_numpy(np.ascontiguousarray(self.data[span[0]:span[1]], dtype=np.float32))

    def run_mtmfft_batches(dev, rows, chans, nfft, taper, taper_opt, demean_taper, ft_compat, polyremoval, freq_idx, output,
                           keeptapers, max_bytes=0, blocked=False, reuse=False, upload=None):
        calls["rows"].extend(rows)
        assert all(0 <= a and b <= dev.shape[0] for a, b in rows), (rows, dev.shape)
        x = dev.numpy()
        specs = []
        for a, b in rows:
            trl = x[a:b] if chans is None else x[a:b][:, chans]
            if polyremoval is not None:
                trl = O.detrend(np.array(trl), polyremoval)
            s, _ = O.mtmfft(np.array(trl), 200.0, nfft, taper, taper_opt, demean_taper=demean_taper)
            specs.append(s if freq_idx is None else s[:, freq_idx])
        spec = torch.from_numpy(np.stack(specs))
        spec.spyhip_blocked = False
        spec.spyhip_ntaper = spec.shape[1]
        yield np.arange(len(rows)), spec

    def csd_accumulate(spec, acc, blocked=False, absmax=None, **kw):
        s = spec.reshape(-1, spec.shape[-2], spec.shape[-1])
        acc += torch.einsum("rfi,rfj->fij", s, s.conj())
        return acc

    def csd_finalize(acc, scale):
        acc *= scale
        return acc

    def csd_allreduce_(acc):
        return parallel.allreduce_sum_(acc)

    AnalogData.device_data = device_data
    AnalogData.upload_in_flight = lambda self: None
    hs.run_mtmfft_batches = run_mtmfft_batches
    backend.csd_accumulate, backend.csd_finalize, backend.csd_allreduce_ = csd_accumulate, csd_finalize, csd_allreduce_
    backend.require_gpu = lambda: None
    backend.to_host = lambda t: t.cpu().numpy()
    return calls


def _st_stage(keeptrials):
    """The product's CrossSpectra routine through its hip entry point: (result, rows this rank transformed)."""
    from syncopy_amd.connectivity.ST_compRoutines import CrossSpectra
    from syncopy_amd.datatype import CrossSpectralData, trial_rows
    calls = _patch()
    data = _data()
    st = CrossSpectra(samplerate=200.0, nSamples=None, foi=None, taper="dpss", taper_opt={"NW": 3.0, "Kmax": 5},
                      demean_taper=False, polyremoval=0, timeAxis=0)
    out = CrossSpectralData(dimord=CrossSpectra.dimord)
    st.initialize(data, out._stackingDim, chan_per_worker=None, keeptrials=keeptrials)
    st.compute(data, out, parallel=False, log_dict={}, method="hip")
    o = data._row_origin
    return (np.asarray(out.data), [(a + o, b + o) for a, b in calls["rows"]], trial_rows(data), list(st.my_trials()),
            calls["staged"][-1])


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        for kt in (False, True):
            val, rows, all_rows, mine, staged = _st_stage(kt)
            res[f"staged{int(kt)}"] = np.array(staged)
            res[f"val{int(kt)}"] = val
            res[f"rows{int(kt)}"] = np.array(rows)
            res[f"mine{int(kt)}"] = np.array(mine)
            res["all_rows"] = np.array(all_rows)
        np.savez(os.path.join(tmp, f"rank{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


def _reference(keeptrials):
    """Per-trial CSDs from the oracle, averaged (or kept) as the reference does."""
    data = _data()
    out = []
    for trl in data.trials:
        s, _ = O.mtmfft(O.detrend(np.array(trl), 0), 200.0, None, "dpss", {"NW": 3.0, "Kmax": 5})
        out.append(np.einsum("kfi,kfj->fij", s, s.conj()) / s.shape[0])
    out = np.stack(out)
    return out if keeptrials else out.mean(axis=0, keepdims=True)


@pytest.mark.parametrize("world", [2, 3])
def test_compute_hip_shards_trials_and_sums_once(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    all_rows = [tuple(r) for r in z[0]["all_rows"]]
    for kt in (0, 1):
        # contiguous, disjoint shards in rank order that cover every trial exactly once - and each rank transformed
        # exactly the rows of its shard
        mine = [list(z[r][f"mine{kt}"]) for r in range(world)]
        assert sum(mine, []) == list(range(len(all_rows)))
        for r in range(world):
            assert [tuple(x) for x in z[r][f"rows{kt}"]] == [all_rows[k] for k in mine[r]]
        # what each rank staged: exactly the rows of its own trials - pairwise disjoint, in rank order, and together the
        # whole recording (these trials neither overlap nor leave gaps)
        staged = [tuple(int(v) for v in z[r][f"staged{kt}"]) for r in range(world)]
        for r in range(world):
            assert staged[r] == (all_rows[mine[r][0]][0], all_rows[mine[r][-1]][1])
        assert staged[0][0] == 0 and staged[-1][1] == all_rows[-1][1]
        assert all(staged[r][1] == staged[r + 1][0] for r in range(world - 1))
        ref = _reference(bool(kt)).astype(np.complex64)
        for r in range(world):
            assert z[r][f"val{kt}"].shape == ref.shape
            assert_parity(z[r][f"val{kt}"], ref, what=f"rank {r} keeptrials={kt}")
            assert np.array_equal(z[r][f"val{kt}"], z[0][f"val{kt}"])          # every rank holds the same result


def _peak_worker(rank, world, port, tmp):
    """precision="auto" look of freqanalysis with more ranks than trials: the rank with an empty shard must still enter
    the all-reduce of the ratio (it used to return early and the others' MAX met its next SUM)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from syncopy_amd import parallel
        from syncopy_amd.specest import hip_spectral as hs

        def run_mtmfft(dev, rows, chans, nfft, *a, **k):          # power spectra with a peak outside the kept band
            p = torch.ones((1, 65, 4))
            p[0, 3] = 1e4
            return [p.clone() for _ in rows]

        hs.run_mtmfft = run_mtmfft
        rows = [(0, 128), (128, 256)]                              # two trials, three ranks: rank 2 has none
        lo, hi = parallel.my_shard(len(rows))
        ans = hs.selection_hides_peak(None, rows[lo:hi], None, None, "hann", None, 0, np.arange(20, 40), 65)
        t = torch.tensor([1.0 if ans else 0.0])
        parallel.allreduce_sum_(t)                                  # the next collective of the real call sequence
        np.savez(os.path.join(tmp, f"peak{rank}.npz"), ans=np.array(ans), total=t.numpy())
    finally:
        dist.destroy_process_group()


def test_precision_look_with_an_empty_shard(tmp_path):
    world = 3
    mp.spawn(_peak_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = [np.load(tmp_path / f"peak{r}.npz") for r in range(world)]
    assert all(bool(x["ans"]) for x in z), "every rank takes the same (float64) branch"
    assert all(float(x["total"][0]) == world for x in z)
