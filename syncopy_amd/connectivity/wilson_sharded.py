"""Frequency-sharded Wilson factorisation + Granger causality (wilson_sf.py:16-254, granger.py:10-79).

The AV stage of `method="granger"` is one big batched computation over the rfft bins of the trial-averaged CSD
(2.3 s on one GPU at 2049 x 256 x 256 in round 1, 57x the ST stage of 1000 trials): at 8 GPUs it IS the run time
unless it is sharded too.  Every rank keeps a contiguous range of bins; regularisation, Cholesky factor, inverse,
products and the convergence error are per frequency and stay local.  Only the plus operator couples the
frequencies (an FFT along the frequency axis per matrix entry, wilson_sf.py:169,182): around it g is transposed
from "my frequencies x all entries" to "all frequencies x my entries" and back - two personalised all-to-alls of
(R-1)/R of the 16 F n^2 bytes per iteration over xGMI - and three small quantities are reduced over the ranks:
gamma_0 (sum), the condition number and the error (max).  psi0 (n x n) is replicated.

The orchestration below is backend-neutral: `prim` supplies the per-shard steps - `HipPrims` runs them through the C
ABI (spyhip_wilson_*); the CPU tests bind the NumPy oracle instead and run the same code under gloo with two ranks.
"""
import ctypes as C

import numpy as np
import torch

from .. import parallel


def granger_sharded(csd_local, f_lo, nftot, prim, rtol=5e-6, niter=100, cond_max=1e4, eps_max=1e-1):
    """csd_local: complex64 (nf, n, n), the bins [f_lo, f_lo + nf) of the nftot rfft bins of the trial-averaged CSD
    (a torch tensor on the device the primitives work on).  Returns (granger float32 (nf, n, n) for the local bins,
    info dict as granger_cF's metadata, H (nf, n, n), Sigma (n, n))."""
    rank, size = parallel.world() if parallel.collective_active() else (0, 1)
    nf, n = int(csd_local.shape[0]), int(csd_local.shape[1])
    nn = n * n
    fb = parallel.shard_bounds(nftot, size)
    if fb[rank] != (f_lo, f_lo + nf):
        raise ValueError(f"rank {rank} holds bins [{f_lo}, {f_lo + nf}) but the contiguous partition gives {fb[rank]}")
    eb = parallel.shard_bounds(nn, size)
    ne = eb[rank][1] - eb[rank][0]

    # ---- regularize_csd (wilson_sf.py:197-254): the ladder decisions are taken on the global maximum
    A, c = prim.cond(csd_local, 0.0)
    cond0 = parallel.allreduce_max(c)
    factor = 0
    if not cond0 < cond_max:
        factor = -1
        for eps in np.logspace(-10, np.log10(eps_max), 15):
            A, c = prim.cond(csd_local, float(eps))
            if parallel.allreduce_max(c) < cond_max:
                factor = float(eps)
                break

    U, gpart = prim.init(A, f_lo, nftot)
    gamma0 = parallel.allreduce_sum_(gpart)
    converged, err = False, float("inf")
    iterations = 0
    for attempt in range(2):          # second attempt: pivoted inverse, if a block inverse met a tiny pivot anywhere
        psi0, psi = prim.psi0(gamma0.clone(), nf)
        tiny = False
        for it in range(niter):
            iterations = it + 1
            g, t = prim.g(psi, U, pivoted=bool(attempt))
            tiny = parallel.allreduce_max(float(t)) > 0
            if tiny:
                break
            # g: (my frequencies, all entries) -> (all frequencies, my entries)
            g2 = g.reshape(nf, nn)
            recv = parallel.exchange([g2[:, lo:hi].contiguous() for lo, hi in eb],
                                     [(b - a, ne) for a, b in fb])
            gpe, g0e = prim.plus(torch.cat(recv, dim=0))
            # g+: back to (my frequencies, all entries); the zero-lag coefficients of all entries to everybody
            recv = parallel.exchange([gpe[a:b].contiguous() for a, b in fb], [(nf, hi - lo) for lo, hi in eb])
            gp = torch.cat(recv, dim=1).reshape(nf, n, n)
            g0 = parallel.allgather_cat(g0e, [hi - lo for lo, hi in eb]).reshape(n, n)
            err = parallel.allreduce_max(prim.update(psi, gp, g0, psi0, A))
            if err < rtol:
                converged = True
                break
        if not tiny:
            break
    G, H, Sigma = prim.finish(A, psi, psi0)
    info = {"converged": bool(converged), "max rel. err": float(err), "reg. factor": factor,
            "initial cond. num": float(cond0), "iterations": iterations}
    return G, info, H, Sigma


class HipPrims:
    """The per-shard steps on the device through the C ABI (include/spyhip.h, "K6 in steps")."""

    def __init__(self, device=None):
        from .. import backend
        self.be = backend
        self.ctx = backend.context(device)
        self.dev = torch.device("cuda", self.ctx.device)
        self._work = None

    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def work(self, nf, n):
        need = 3 * nf * n * n
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.complex128, device=self.dev)
        return self._work

    def _call(self, name, *args):
        self.ctx.bind_stream()
        rc = getattr(self.ctx.lib, name)(self.ctx.handle, *args)
        if rc < 0:
            self.be.check(rc, name)
        return rc

    def cond(self, csd_local, eps):
        nf, n, _ = csd_local.shape
        A = torch.empty((nf, n, n), dtype=torch.complex128, device=self.dev)
        out = C.c_double()
        self._call("spyhip_wilson_cond", self._p(csd_local.contiguous()), nf, n, float(eps), self._p(A),
                   self._p(self.work(nf, n)), C.byref(out))
        return A, out.value

    def init(self, A, f_lo, nftot):
        nf, n, _ = A.shape
        U = torch.empty_like(A)
        gpart = torch.empty((n, n), dtype=torch.complex128, device=self.dev)
        self._call("spyhip_wilson_init", self._p(A), nf, n, int(f_lo), int(nftot), self._p(U), self._p(gpart))
        return U, gpart

    def psi0(self, gamma0, nf):
        n = gamma0.shape[0]
        psi0 = torch.empty((n, n), dtype=torch.complex128, device=self.dev)
        psi = torch.empty((nf, n, n), dtype=torch.complex128, device=self.dev)
        self._call("spyhip_wilson_psi0", self._p(gamma0), n, nf, self._p(psi0), self._p(psi))
        return psi0, psi

    def g(self, psi, U, pivoted):
        nf, n, _ = psi.shape
        g = torch.empty_like(psi)
        rc = self._call("spyhip_wilson_g", self._p(psi), self._p(U), nf, n, int(bool(pivoted)), self._p(self.work(nf, n)),
                        self._p(g))
        return g, rc == 1

    def plus(self, ge):
        nftot, nent = ge.shape
        ge = ge.contiguous()
        gp = torch.empty_like(ge)
        g0 = torch.zeros((nent,), dtype=torch.complex128, device=self.dev)
        self._call("spyhip_wilson_plus", self._p(ge), int(nftot), int(nent), self._p(gp), self._p(g0))
        return gp, g0

    def update(self, psi, gp, g0, psi0, A):
        nf, n, _ = psi.shape
        err = C.c_double()
        self._call("spyhip_wilson_update", self._p(psi), self._p(gp.contiguous()), self._p(g0.contiguous()), self._p(psi0),
                   self._p(A), nf, n, self._p(self.work(nf, n)), C.byref(err))
        return err.value

    def finish(self, A, psi, psi0):
        nf, n, _ = psi.shape
        G = torch.empty((nf, n, n), dtype=torch.float32, device=self.dev)
        H = torch.empty((nf, n, n), dtype=torch.complex128, device=self.dev)
        Sigma = torch.empty((n, n), dtype=torch.complex128, device=self.dev)
        self._call("spyhip_wilson_finish", self._p(A), self._p(psi), self._p(psi0), nf, n, self._p(self.work(nf, n)),
                   self._p(G), self._p(H), self._p(Sigma))
        return G, H, Sigma


def granger_hip_sharded(csd, rtol=5e-6, niter=100, cond_max=1e4, eps_max=1e-1):
    """The AV stage of method='granger' with the frequencies sharded over the ranks of the process group: `csd` is
    the FULL trial-averaged CSD (F, n, n) complex64 every rank holds after the ST stage's all-reduce; each rank
    factorises its range of bins and the Granger values are gathered.  Returns (granger (F, n, n) float32, info)."""
    rank, size = parallel.world() if parallel.collective_active() else (0, 1)
    F, n = int(csd.shape[0]), int(csd.shape[1])
    if F < 8 * size or n * n < size:
        # a handful of bins: not worth two exchanges per iteration (and no rank may hold an empty shard) -
        # every rank factorises the whole spectrum, identically
        from .. import backend
        return backend.granger(csd, rtol=rtol, niter=niter, cond_max=cond_max, eps_max=eps_max)
    fb = parallel.shard_bounds(F, size)
    lo, hi = fb[rank]
    G, info, _, _ = granger_sharded(csd[lo:hi].contiguous(), lo, F, HipPrims(csd.device), rtol, niter, cond_max, eps_max)
    return parallel.allgather_cat(G, [b - a for a, b in fb]), info
