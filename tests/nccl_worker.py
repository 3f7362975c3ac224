"""Worker of tests/test_gpu_nccl.py (launched by torch.distributed.run, one rank per visible GPU): the product's
coherence path under an initialised "nccl" (= RCCL) process group.  With SPY_FORCE_COLLECTIVE=1 a group of ONE rank
still runs csd_tril_pack -> all_reduce -> csd_tril_unpack, so the collective leg executes on a 1-GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    try:
        import syncopy_amd as spy
        from syncopy_amd import backend as be
        from syncopy_amd import parallel
        assert parallel.collective_active()
        adj = np.zeros((37, 37))
        adj[0, 1] = adj[5, 30] = 0.3
        data = spy.synthdata.ar2_network(AdjMat=adj, nSamples=1024, nTrials=11, seed=3, samplerate=500)
        res = {}
        res["coh"] = spy.connectivityanalysis(data, method="coh", tapsmofrq=3).data
        res["csd"] = spy.connectivityanalysis(data, method="csd", tapsmofrq=3).data
        res["ppc"] = spy.connectivityanalysis(data, method="ppc", tapsmofrq=3).data
        res["corr"] = spy.connectivityanalysis(data, method="corr").data
        res["pow_avg"] = spy.freqanalysis(data, method="mtmfft", tapsmofrq=3, keeptrials=False).data
        res["pow"] = spy.freqanalysis(data, method="mtmfft", tapsmofrq=3).data
        # AV stage with the frequencies sharded over the ranks (spyhip_wilson_* steps, wilson_sharded.py)
        gr = spy.connectivityanalysis(data, method="granger", tapsmofrq=3)
        res["granger"] = gr.data
        res["granger_info"] = np.array([gr.info["converged"], gr.info["max rel. err"], gr.info["reg. factor"],
                                        gr.info["initial cond. num"]], dtype=np.float64)
        # bench.py's own sequence on a raw accumulator, bit-compared with the untouched lower triangle
        g = torch.Generator(device="cuda").manual_seed(1)
        spec = torch.view_as_complex(torch.randn((21, 130, 256, 2), generator=g, device="cuda"))
        acc = torch.zeros((130, 256, 256), dtype=torch.complex64, device="cuda")
        be.csd_accumulate(spec, acc)
        before = acc.clone()
        packed = be.csd_tril_pack(acc)
        dist.all_reduce(torch.view_as_real(packed))
        acc.fill_(7.0)
        be.csd_tril_unpack(packed, acc)
        ii, jj = np.tril_indices(256)
        world = dist.get_world_size()
        same = torch.equal(torch.view_as_real(acc[:, ii, jj].contiguous()),
                           torch.view_as_real((before[:, ii, jj] * world).contiguous()))
        res["pack_allreduce_unpack_bit_exact"] = np.array(same)
        # the product's ONE collective path: backend.csd_allreduce_ = spyhip_allreduce_csd on the library's own RCCL
        # communicator (bootstrapped through this process group); same bits as the torch.distributed sum above
        acc2 = before.clone()
        be.csd_allreduce_(acc2)
        res["library_allreduce_bit_exact"] = np.array(torch.equal(
            torch.view_as_real(acc2[:, ii, jj].contiguous()), torch.view_as_real(acc[:, ii, jj].contiguous())))
        res["library_comm_up"] = np.array(bool(be._lib_comm))
        # the communicator cannot be created (here: its init entry point made to fail on this rank): every rank learns it
        # through the process group, nobody waits in a collective the others never enter, and the sum takes the
        # torch.distributed route with the same bits
        be.shutdown_library_comm()
        ctx = be.context()
        real_init = ctx.lib.spyhip_comm_init
        ctx.lib.spyhip_comm_init = lambda *a: -1
        try:
            acc3 = before.clone()
            be.csd_allreduce_(acc3)
        finally:
            ctx.lib.spyhip_comm_init = real_init
        res["fallback_allreduce_bit_exact"] = np.array(torch.equal(
            torch.view_as_real(acc3[:, ii, jj].contiguous()), torch.view_as_real(acc[:, ii, jj].contiguous())))
        res["fallback_taken"] = np.array(bool(be._lib_comm_failed) and not be._lib_comm)
        be._lib_comm_failed.clear()                  # (the analyses below use the library's communicator again)
        if os.environ.get("SPY_NCCL_C5"):
            # BASELINE configs[4] through the front end, trial-sharded over the ranks of this group: every rank holds
            # the recording, transforms its contiguous shard of the trials, ONE all-reduce of the packed CSD triangle,
            # then the frequency-sharded Wilson factorisation (tests/test_gpu_production.py holds the group-less twin)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from test_gpu_production import c5_dataset
            T5 = int(os.environ["SPY_NCCL_C5"])
            big = c5_dataset(T5)
            g5 = spy.connectivityanalysis(big, method="granger", tapsmofrq=1)
            res["c5_info"] = np.array([g5.info["converged"], g5.info["max rel. err"], g5.info["reg. factor"],
                                       g5.info["initial cond. num"]], dtype=np.float64)
            res["c5_sample"] = np.ascontiguousarray(g5.data[0, ::64, :16, :16])
            del big, g5
            spy.release_device_buffers()
        # what this rank staged into HBM for the analyses above: only the rows of its own trial shard
        # (AnalogData.shard_span: the reference's workers read only their slab, kwarg_decorators.py:684-735)
        lo, hi = data.staged_rows
        mine = torch.tensor([lo, hi, (hi - lo) * data.data.shape[1] * 4], dtype=torch.int64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        res["staged_rows_and_bytes_per_rank"] = torch.stack(allr).cpu().numpy()
        res["recording_rows"] = np.array(data.data.shape[0])
        res["world"] = np.array(world)
        torch.cuda.synchronize()
        if dist.get_rank() == 0:
            np.savez(out_path, **res)
    finally:
        try:
            from syncopy_amd import backend
            torch.cuda.synchronize()
            backend.shutdown_library_comm()
        finally:
            dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
