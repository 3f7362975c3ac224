"""Trial sharding across GPUs: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Semantics of the reference's trial-parallel map (computational_routine.py:806-942): trials are
independent; outputs with keeptrials=True are concatenated, outputs with keeptrials=False are
one sum-reduction (the reference's mutex-guarded `+=`, kwarg_decorators.py:723-735) followed by a
single division by the global trial count.  Here: contiguous trial ranges per rank (remainder
to the low ranks), ONE all-reduce of the accumulator, no other data-path collective.
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    """(rank, world_size); (0, 1) without an initialised process group."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collective_active():
    """True when sums over ranks have to run: more than one rank - or a process group of ONE rank with
    SPY_FORCE_COLLECTIVE=1, which lets the pack -> all-reduce -> unpack path be executed (and bit-compared with the
    group-less result) on a 1-GPU box."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or bool(os.environ.get("SPY_FORCE_COLLECTIVE"))


def shard_bounds(n, size):
    """Contiguous ranges [lo, hi) of n trials for `size` ranks, remainder to the low ranks."""
    base, rem = divmod(n, size)
    bounds, lo = [], 0
    for r in range(size):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def my_shard(n):
    rank, size = world()
    return shard_bounds(n, size)[rank]


def allreduce_sum_(t):
    """In-place sum over ranks of a torch tensor (complex tensors go as interleaved floats);
    fixed reduction algorithm of the backend => identical result on every rank."""
    if collective_active():
        dist.all_reduce(torch.view_as_real(t) if t.is_complex() else t, op=dist.ReduceOp.SUM)
    return t


def allreduce_sum_numpy(a):
    """Sum over ranks of a host array (used by the sequential / oracle-bound engine path)."""
    _, size = world()
    if size == 1:
        return a
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    allreduce_sum_(t)
    return t.cpu().numpy()


def gather_trials(local):
    """Concatenate per-rank host arrays along axis 0 in rank order (keeptrials=True outputs)."""
    _, size = world()
    if size == 1:
        return local
    parts = [None] * size
    dist.all_gather_object(parts, local)
    return np.concatenate([p for p in parts if p is not None and p.shape[0] > 0], axis=0)


def allreduce_max(value):
    """Largest value of a host scalar over the ranks (condition numbers, convergence errors)."""
    if not collective_active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def exchange(send, recv_shapes):
    """Personalised all-to-all: rank q receives send[q] of every rank.  `send`: one contiguous tensor per rank,
    `recv_shapes`: the shape of what rank r sends to me, per r.  Point-to-point (isend / irecv), so it runs on RCCL
    and on gloo alike; complex tensors travel as interleaved reals."""
    rank, size = world()
    if size == 1 or not collective_active():
        return [send[0]]
    out = [torch.empty(tuple(recv_shapes[r]), dtype=send[rank].dtype, device=send[rank].device) for r in range(size)]
    out[rank].copy_(send[rank])
    ops = []
    view = (lambda t: torch.view_as_real(t)) if send[rank].is_complex() else (lambda t: t)
    for r in range(size):
        if r == rank:
            continue
        if send[r].numel():
            ops.append(dist.P2POp(dist.isend, view(send[r]), r))
        if out[r].numel():
            ops.append(dist.P2POp(dist.irecv, view(out[r]), r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def allgather_cat(local, sizes, dim=0):
    """Concatenate per-rank tensors whose extent along `dim` is sizes[r] (all-gather with padding to the largest)."""
    rank, size = world()
    if size == 1 or not collective_active():
        return local
    big = max(sizes)
    shape = list(local.shape)
    shape[dim] = big
    pad = torch.zeros(shape, dtype=local.dtype, device=local.device)
    pad.narrow(dim, 0, local.shape[dim]).copy_(local)
    parts = [torch.empty_like(pad) for _ in range(size)]
    cplx = local.is_complex()
    dist.all_gather([torch.view_as_real(p) for p in parts] if cplx else parts,
                    torch.view_as_real(pad) if cplx else pad)
    return torch.cat([parts[r].narrow(dim, 0, sizes[r]) for r in range(size)], dim=dim)
