# -*- coding: utf-8 -*-
"""
The drop-in run under the reference's OWN engine (build container only; launched by oracle/check_dropin.sh).

What is checked: syncopy_amd's compute functions bound into the reference's compute classes exactly as INTEGRATION.md
section B prescribes -

    MultiTaperFFT.computeFunction         = staticmethod(process_io(mtmfft_cF))          specest/compRoutines.py:207
    CrossSpectra.computeFunction          = staticmethod(process_io(cross_spectra_cF))   connectivity/ST_compRoutines.py:447
    NormalizeCrossSpectra.computeFunction = staticmethod(process_io(normalize_csd_cF))   connectivity/AV_compRoutines.py:134
    GrangerCausality.computeFunction      = staticmethod(process_io(granger_cF))         connectivity/AV_compRoutines.py:449
    MultiTaperFFTConvol.computeFunction   = staticmethod(process_io(mtmconvol_cF))       specest/compRoutines.py:417
    WaveletTransform.computeFunction      = staticmethod(process_io(wavelet_cF))         specest/compRoutines.py:598
    SpectralDyadicProduct.computeFunction = staticmethod(process_io(spectral_dyadic_product_cF))   connectivity/ST_compRoutines.py:120

- survive the real `spy.freqanalysis` / `spy.connectivityanalysis`: the dry run of ComputationalRoutine.initialize
(shared/computational_routine.py:240-340), the trial loop of compute / compute_sequential (:724-731, :944-1036) with
process_io's HDF5 slabs (shared/kwarg_decorators.py:587-739), the trial average, h5_add_metadata / process_metadata and
check_freq_hashes (shared/metadata.py:135-222, :297) on the `freqs_hash` every cF returns, and the Granger `.info`
entries; the time-frequency routines with their per-trial positional arguments (`soi` / `preselect`, `postselect`:
computational_routine.py:985-990) and the `_make_trialdef` metadata (specest/compRoutines.py:813); SpectralData input
chained into connectivityanalysis; and ONE analysis with parallel=True on a dask.distributed LocalCluster of two
workers (the dict-of-HDF5-coordinates branch of process_io, kwarg_decorators.py:646-737, and pickling of the bound
compute function).  Results are compared with the fixtures the unmodified reference wrote (tests/golden/c1.npz, conn5.npz,
tf_variants.npz, chain.npz).

Two interpreters, because no single one in this image has both worlds: the reference needs h5py / dask
(/opt/conda/bin/python3.9, no torch), the product's compute functions need torch (/usr/bin/python3, no h5py).  The
reference process (this file, client role) therefore binds PROXIES whose calls travel over a pipe to a server process
(this file, `--serve`) that runs syncopy_amd's real compute functions.  There is no GPU in the build container: inside the
server the DEVICE PRIMITIVES the compute functions reach for - the transform of a batch of trials, the cross-spectral
accumulation, scaling / normalisation, the Wilson factorisation, host <-> device copies - are replaced by CPU stand-ins
built on the oracle, as tests/test_compute_hip_sharding_gloo.py does; everything above them (argument handling, dry-run
contract, frequency selection, taper bookkeeping, metadata, return shapes and dtypes) is the product's own code.

TEST INFRASTRUCTURE ONLY - never imported by the product; the reference never leaves the container.
"""
import io
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CFS = {
    "mtmfft_cF": "syncopy_amd.specest.compRoutines",
    "cross_spectra_cF": "syncopy_amd.connectivity.ST_compRoutines",
    "normalize_csd_cF": "syncopy_amd.connectivity.AV_compRoutines",
    "granger_cF": "syncopy_amd.connectivity.AV_compRoutines",
    "mtmconvol_cF": "syncopy_amd.specest.compRoutines",
    "wavelet_cF": "syncopy_amd.specest.compRoutines",
    "spectral_dyadic_product_cF": "syncopy_amd.connectivity.ST_compRoutines",
}


# ------------------------------------------------------------------------------------------------ wire format
# (no pickles across numpy 1.26 <-> 2.x: arrays travel as .npy, the rest as a small tagged tree)
def _pack(obj, out):
    if isinstance(obj, np.ndarray) or isinstance(obj, np.generic):
        buf = io.BytesIO()
        np.save(buf, np.asarray(obj), allow_pickle=False)
        b = buf.getvalue()
        out.write(b"A" + struct.pack("<q", len(b)) + b)
    elif obj is None:
        out.write(b"N")
    elif isinstance(obj, bool):
        out.write(b"T" if obj else b"F")
    elif isinstance(obj, int):
        out.write(b"I" + struct.pack("<q", obj))
    elif isinstance(obj, float):
        out.write(b"D" + struct.pack("<d", obj))
    elif isinstance(obj, str):
        b = obj.encode()
        out.write(b"S" + struct.pack("<q", len(b)) + b)
    elif isinstance(obj, (tuple, list)):
        out.write((b"U" if isinstance(obj, tuple) else b"L") + struct.pack("<q", len(obj)))
        for v in obj:
            _pack(v, out)
    elif isinstance(obj, dict):
        out.write(b"M" + struct.pack("<q", len(obj)))
        for k, v in obj.items():
            _pack(str(k), out)
            _pack(v, out)
    elif isinstance(obj, slice):
        out.write(b"X")
        for v in (obj.start, obj.stop, obj.step):
            _pack(None if v is None else int(v), out)
    elif isinstance(obj, np.dtype) or (isinstance(obj, type) and issubclass(obj, np.generic)):
        _pack("dtype:" + np.dtype(obj).str, out)
    elif type(obj).__module__.startswith("syncopy.specest.wavelets"):
        # the wavelet function object of the reference's method_kwargs (freqanalysis.py:893-897): class names + parameters
        _pack({"__wavelet__": [c.__name__ for c in type(obj).__mro__ if c is not object],
               "w0": float(getattr(obj, "w0", 6.0)), "m": int(getattr(obj, "m", 0))}, out)
    else:
        raise TypeError("cannot send %r" % type(obj))


def _unpack(inp):
    tag = inp.read(1)
    if not tag:
        raise EOFError
    if tag == b"A":
        (n,) = struct.unpack("<q", inp.read(8))
        return np.load(io.BytesIO(inp.read(n)), allow_pickle=False)
    if tag == b"N":
        return None
    if tag in (b"T", b"F"):
        return tag == b"T"
    if tag == b"I":
        return struct.unpack("<q", inp.read(8))[0]
    if tag == b"D":
        return struct.unpack("<d", inp.read(8))[0]
    if tag == b"S":
        (n,) = struct.unpack("<q", inp.read(8))
        s = inp.read(n).decode()
        return np.dtype(s[6:]) if s.startswith("dtype:") else s
    if tag in (b"U", b"L"):
        (n,) = struct.unpack("<q", inp.read(8))
        items = [_unpack(inp) for _ in range(n)]
        return tuple(items) if tag == b"U" else items
    if tag == b"X":
        return slice(_unpack(inp), _unpack(inp), _unpack(inp))
    if tag == b"M":
        (n,) = struct.unpack("<q", inp.read(8))
        d = {_unpack(inp): _unpack(inp) for _ in range(n)}
        if "__wavelet__" in d:
            # an object with the reference's class names and attributes, as syncopy_amd's wavelet_cF reads them
            cls = None
            for name in reversed(d["__wavelet__"]):
                cls = type(name, (cls,) if cls else (object,), {})
            obj = cls()
            obj.w0, obj.m = d["w0"], d["m"]
            return obj
        return d
    raise ValueError("bad tag %r" % tag)


# ------------------------------------------------------------------------------------------------ server role
def serve():
    """System interpreter: syncopy_amd's compute functions on CPU stand-ins of the device primitives."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch
    from oracle import spy_oracle as O
    from syncopy_amd import backend
    from syncopy_amd.specest import hip_spectral as hs

    inp, out = sys.stdin.buffer, os.fdopen(os.dup(1), "wb")
    sys.stdout = sys.stderr                                       # stray prints must not corrupt the pipe

    # ---- device primitives -> CPU (the oracle's arithmetic); tensors stay torch CPU tensors
    torch.Tensor.cuda = lambda self, *a, **k: self
    backend.require_gpu = lambda: None
    backend.to_host = lambda t: t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    def spectra(x, nfft, taper, taper_opt, demean_taper, polyremoval):
        if polyremoval is not None and polyremoval is not False:
            x = O.detrend(np.array(x), polyremoval)
        s, _ = O.mtmfft(np.array(x), 1.0, nfft, taper, taper_opt, demean_taper=demean_taper)
        return s                                                  # (K, F, C) complex64; samplerate only labels the axis

    def run_mtmfft(dev, rows, chans, nfft, taper, taper_opt, demean_taper, ft_compat, polyremoval, freq_idx, output, keeptapers):
        res = []
        for a, b in rows:
            x = dev.numpy()[a:b]
            x = x if chans is None else x[:, chans]
            s = spectra(x, nfft, taper, taper_opt, demean_taper, polyremoval)
            s = s if freq_idx is None else s[:, freq_idx]
            s = O.convert_output(s, output)
            if not keeptapers:
                s = s.mean(axis=0, keepdims=True)
            res.append(torch.from_numpy(np.ascontiguousarray(s)))
        return res

    def run_mtmfft_batches(dev, rows, chans, nfft, taper, taper_opt, demean_taper, ft_compat, polyremoval, freq_idx, output,
                           keeptapers, max_bytes=0, blocked=False, reuse=False, upload=None):
        specs = run_mtmfft(dev, rows, chans, nfft, taper, taper_opt, demean_taper, ft_compat, polyremoval, freq_idx, output, True)
        spec = torch.stack(specs)
        spec.spyhip_blocked, spec.spyhip_ntaper, spec.spyhip_absmax = False, spec.shape[1], None
        yield np.arange(len(rows)), spec

    def csd_accumulate(spec, acc, blocked=False, absmax=None, **kw):
        s = spec.reshape(-1, spec.shape[-2], spec.shape[-1])
        acc += torch.einsum("rfi,rfj->fij", s, s.conj())
        return acc

    def csd_finalize(acc, scale):
        acc *= scale
        return acc

    def coh_normalize(csd, output="abs"):
        return torch.from_numpy(np.ascontiguousarray(O.normalize_csd(csd.numpy()[None], output)[0]))

    def granger(csd, rtol=5e-6, niter=100, cond_max=1e4, eps_max=1e-1, want_factors=False):
        res, meta = O.granger_cF(csd.numpy()[None], rtol=rtol, nIter=niter, cond_max=cond_max)
        info = {k.split("--")[0]: v.item() if hasattr(v, "item") else v for k, v in meta.items()}
        return torch.from_numpy(np.ascontiguousarray(res[0], dtype=np.float32)), info

    def run_stft(dev, row0, s0, s1, frames, nperseg, step, boundary, chans, taper, taper_opt, polyremoval, fidx, output, keeptapers):
        x = dev.numpy()[row0 + s0:row0 + s1]
        x = x if chans is None else x[:, chans]
        dk = "constant" if polyremoval == 0 else "linear" if polyremoval == 1 else False
        ftr, _ = O.mtmconvol(x, 1.0, nperseg, nperseg - step, taper, taper_opt, "zeros" if boundary else None, bool(boundary), dk)
        s = ftr[np.asarray(frames, dtype=np.int64)]
        s = s if fidx is None else s[:, :, fidx]
        s = O.convert_output(s, output)
        if not keeptapers:
            s = np.nanmean(s, axis=1, keepdims=True)
        return torch.from_numpy(np.ascontiguousarray(s))

    class CWTPlan:
        """backend.CWTPlan on the oracle's cwt: detrend the whole trial, transform the pre-selected samples, keep the
        post-selected ones (the contract of spyhip_cwt_exec, include/spyhip.h)."""

        def __init__(self, nsig, nchan, scales, dt, w0=6.0, detrend=None, output="pow", tpos=None, ntime_out=None, device=None,
                     sl_cycles=None, k_sd=5.0, family=None, order=None):
            self.nsig, self.nchan, self.scales, self.dt, self.w0 = int(nsig), int(nchan), np.asarray(scales, dtype=float), dt, w0
            self.detrend, self.output, self.family, self.order = detrend, output, family, order
            self.tpos = None if tpos is None else np.asarray(tpos)
            self.ntime_out = self.nsig if tpos is None else int(ntime_out)
            self.out_dtype = torch.complex64 if output == "fourier" else torch.float32

        def set_precision(self, reference=True):
            return True

        def out_shape(self, nseg):
            return (nseg, self.ntime_out, self.scales.size, self.nchan)

        def execute(self, data, seg_start, trial_lo, trial_hi, chan_idx=None, out=None, accumulate=False):
            res = []
            for st, lo, hi in zip(seg_start.tolist(), trial_lo.tolist(), trial_hi.tolist()):
                x = data.numpy()[lo:hi]
                x = x if chan_idx is None else x[:, chan_idx.numpy()]
                x = O.detrend(np.array(x), self.detrend)
                sp = O.cwt(x[st - lo:st - lo + self.nsig], 1.0 / self.dt, self.scales, self.w0, self.family, self.order).transpose(1, 0, 2)
                if self.tpos is not None:
                    keep = np.nonzero(self.tpos >= 0)[0]
                    sel = np.empty((self.ntime_out,) + sp.shape[1:], dtype=sp.dtype)
                    sel[self.tpos[keep]] = sp[keep]
                    sp = sel
                res.append(O.convert_output(sp, self.output))
            res = torch.from_numpy(np.ascontiguousarray(np.stack(res)))
            if out is None:
                return res
            if accumulate == 2:
                out += res.sum(dim=0, keepdim=True)
            elif accumulate:
                out += res
            else:
                out.copy_(res)
            return out

    hs.run_mtmfft, hs.run_mtmfft_batches, hs.run_stft = run_mtmfft, run_mtmfft_batches, run_stft
    backend.CWTPlan = CWTPlan
    backend.csd_accumulate, backend.csd_finalize = csd_accumulate, csd_finalize
    backend.coh_normalize, backend.granger = coh_normalize, granger

    class Faux:
        """shape / dtype stand-in of a trial for the dry run (the reference hands its FauxTrial, base_data.py:1458)"""

        def __init__(self, shape, dtype):
            self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)

        @property
        def T(self):
            return Faux(self.shape[::-1], self.dtype)

    funcs = {name: getattr(importlib.import_module(mod), name) for name, mod in CFS.items()}
    calls = {name: 0 for name in funcs}
    while True:
        try:
            name, args, kwargs = _unpack(inp)
        except EOFError:
            break
        if name == "__stats__":
            _pack(("ok", {k: int(v) for k, v in calls.items()}), out)
            out.flush()
            continue
        if name == "__signature__":                               # [(parameter, has a default, default)] of a compute function
            import inspect
            sig = inspect.signature(funcs[args[0]])
            _pack(("ok", [[k, v.default is not v.empty, None if v.default is v.empty else v.default]
                          for k, v in sig.parameters.items()]), out)
            out.flush()
            continue
        try:
            calls[name] += 1
            if isinstance(args[0], dict) and "faux_shape" in args[0]:
                args[0] = Faux(args[0]["faux_shape"], args[0]["faux_dtype"])
            res = funcs[name](*args, **kwargs)
            _pack(("ok", res), out)
        except Exception as exc:                                  # noqa: BLE001 - reported to the client, which raises
            import traceback
            _pack(("error", "%s: %s\n%s" % (type(exc).__name__, exc, traceback.format_exc())), out)
        out.flush()


# ------------------------------------------------------------------------------------------------ client role
class Server:
    """The product's interpreter behind a pipe, started on first use.  Pickles to an unstarted Server: a Dask worker that
    receives the bound compute function starts its own."""

    def __init__(self):
        self.p = None

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.p = None

    def start(self):
        # a clean environment: nothing of the conda interpreter (its MKL / library paths) may leak into the system one
        env = {k: v for k, v in os.environ.items() if k in ("PATH", "HOME", "TMPDIR", "LANG", "HSA_ENABLE_IPC_MODE_LEGACY")}
        env["PATH"] = "/usr/local/sbin:/usr/local/bin:/usr/sbin:/usr/bin:/sbin:/bin"
        env["OMP_NUM_THREADS"] = "2"
        self.p = subprocess.Popen(["/usr/bin/python3", os.path.abspath(__file__), "--serve"], stdin=subprocess.PIPE,
                                  stdout=subprocess.PIPE, env=env)

    def call(self, name, *args, **kwargs):
        if self.p is None:
            self.start()
        _pack((name, list(args), kwargs), self.p.stdin)
        self.p.stdin.flush()
        status, res = _unpack(self.p.stdout)
        if status != "ok":
            raise RuntimeError("syncopy_amd." + name + " failed in the server:\n" + res)
        return res

    def close(self):
        if self.p is not None:
            self.p.stdin.close()
            self.p.wait(timeout=60)
            self.p = None


def excess(a, b, rtol=1e-5, atol_rel=1e-6):
    a, b = np.asarray(a), np.asarray(b)
    if not np.iscomplexobj(b) and np.isnan(b).any():                # (windows a trial does not fill: NaN on both sides)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        a, b = np.nan_to_num(a), np.nan_to_num(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    tol = rtol * np.abs(b) + atol_rel * np.abs(b).max()
    return float((np.abs(a.astype(np.complex128) - b) / np.where(tol == 0, np.finfo(np.float32).tiny, tol)).max())


def main():
    import syncopy as spy
    from syncopy import synthdata
    from syncopy.shared.kwarg_decorators import process_io
    from syncopy.specest import compRoutines as ref_spec
    from syncopy.connectivity import ST_compRoutines as ref_st, AV_compRoutines as ref_av

    srv = Server()

    import inspect

    def proxy(name, ref_cF):
        # ComputationalRoutine.__init__ builds its cfg from the compute function's SIGNATURE (get_defaults,
        # shared/tools.py:346-376; computational_routine.py:138-144): the proxy carries the product function's own, and
        # that must be the reference's - same parameters, same order, same defaults
        params = [inspect.Parameter(k, inspect.Parameter.POSITIONAL_OR_KEYWORD,
                                    default=d if has else inspect.Parameter.empty) for k, has, d in srv.call("__signature__", name)]
        ours = inspect.Signature(params)
        theirs = inspect.signature(ref_cF)
        assert [(p.name, p.default) for p in ours.parameters.values()] == \
               [(p.name, p.default) for p in theirs.parameters.values()], (name, str(ours), str(theirs))
        say("signature of syncopy_amd's %s equals the reference's: %s" % (name, ours))

        def cF(trl_dat, *args, **kwargs):
            if isinstance(trl_dat, np.ndarray):
                first = trl_dat
            elif hasattr(trl_dat, "shape") and hasattr(trl_dat, "dtype") and not hasattr(trl_dat, "__array__"):
                first = {"faux_shape": [int(v) for v in trl_dat.shape], "faux_dtype": np.dtype(trl_dat.dtype)}   # dry run
            else:
                first = np.asarray(trl_dat)                        # an h5py dataset slab
            res = srv.call(name, first, *args, **kwargs)
            if kwargs.get("noCompute"):
                shape, dtype = res                                # the dry-run contract: (shape, dtype)
                return tuple(int(v) for v in shape), np.dtype(dtype)
            if isinstance(res, tuple):                            # (array, metadata): values as the reference's cFs give them
                return res[0], {k: np.asarray(v) for k, v in res[1].items()}
            return res
        cF.__name__ = name
        cF.__signature__ = ours
        return process_io(cF)

    log = []

    def say(msg):
        print(msg, flush=True)
        log.append(msg)

    seen = []                                                      # (where, entries, distinct freqs_hash values)

    def spy_on(mod, where):
        orig = mod.check_freq_hashes

        def check_freq_hashes(metadata, out):
            hashes = {bytes(np.asarray(v).tobytes()) for k, v in metadata.items() if "freqs_hash" in k}
            seen.append((where, len(metadata), len(hashes)))
            return orig(metadata, out)
        mod.check_freq_hashes = check_freq_hashes

    spy_on(ref_spec, "MultiTaperFFT")
    spy_on(ref_st, "CrossSpectra")
    ref_spec.MultiTaperFFT.computeFunction = staticmethod(proxy("mtmfft_cF", ref_spec.mtmfft_cF))
    ref_st.CrossSpectra.computeFunction = staticmethod(proxy("cross_spectra_cF", ref_st.cross_spectra_cF))
    ref_av.NormalizeCrossSpectra.computeFunction = staticmethod(proxy("normalize_csd_cF", ref_av.normalize_csd_cF))
    ref_av.GrangerCausality.computeFunction = staticmethod(proxy("granger_cF", ref_av.granger_cF))
    ref_spec.MultiTaperFFTConvol.computeFunction = staticmethod(proxy("mtmconvol_cF", ref_spec.mtmconvol_cF))
    ref_spec.WaveletTransform.computeFunction = staticmethod(proxy("wavelet_cF", ref_spec.wavelet_cF))
    ref_st.SpectralDyadicProduct.computeFunction = staticmethod(proxy("spectral_dyadic_product_cF", ref_st.spectral_dyadic_product_cF))

    g1 = np.load(os.path.join(ROOT, "tests", "golden", "c1.npz"))
    g5 = np.load(os.path.join(ROOT, "tests", "golden", "conn5.npz"))
    worst = 0.0

    # ---- BASELINE config 1 through the reference's front ends, the product's compute functions underneath
    c1 = synthdata.ar2_network(nTrials=20, nSamples=2000, AdjMat=np.zeros((16, 16)), seed=42)
    s = spy.freqanalysis(c1, method="mtmfft", tapsmofrq=2)
    e = excess(np.array(s.data), g1["pow"])
    worst = max(worst, e)
    say("freqanalysis(mtmfft, tapsmofrq=2) on config 1: %s %s, err/tol vs tests/golden/c1.npz[pow] = %.3g; freq axis equal: %s"
        % (s.data.shape, s.data.dtype, e, bool(np.array_equal(s.freq, g1["freq"]))))
    assert np.array_equal(s.freq, g1["freq"]) and tuple(s.data.shape) == g1["pow"].shape and e <= 1.0

    coh = spy.connectivityanalysis(c1, method="coh", tapsmofrq=2)
    e = excess(np.array(coh.data), g1["coh_abs"])
    worst = max(worst, e)
    say("connectivityanalysis(coh, tapsmofrq=2) on config 1: %s %s, err/tol vs c1.npz[coh_abs] = %.3g"
        % (coh.data.shape, coh.data.dtype, e))
    assert e <= 1.0
    csd = spy.connectivityanalysis(c1, method="csd", tapsmofrq=2, foilim=[0, 60])
    e = excess(np.array(csd.data), g1["csd_foilim_0_60"])
    worst = max(worst, e)
    say("connectivityanalysis(csd, foilim=[0, 60]) on config 1: %s %s, err/tol vs c1.npz[csd_foilim_0_60] = %.3g; freq axis equal: %s"
        % (csd.data.shape, csd.data.dtype, e, bool(np.array_equal(csd.freq, g1["csd_freq"]))))
    assert e <= 1.0 and np.array_equal(csd.freq, g1["csd_freq"])

    # ---- the freqs_hash every trial's cF call returned went through h5_add_metadata, metadata_from_hdf5_file and
    # check_freq_hashes (metadata.py:11-60, :179-222, :297) inside process_metadata - a mismatch raises there; `seen` is
    # what the reference's own check was handed
    assert [n for _, n, _ in seen] == [20, 20, 20] and all(h == 1 for _, _, h in seen), seen
    say("check_freq_hashes ran inside process_metadata of %s: %s metadata entries each, one distinct freqs_hash per analysis"
        % (", ".join(w for w, _, _ in seen), seen[0][1]))

    # ---- Granger on the coupled 5-channel network: values and the .info entries process_metadata extracts
    d5 = spy.AnalogData(data=[np.array(t) for t in g5["data"]], samplerate=float(g5["samplerate"]))
    gr = spy.connectivityanalysis(d5, method="granger", tapsmofrq=3)
    got = np.array(gr.data)
    ref = g5["granger"]
    off = ~np.eye(5, dtype=bool)
    # (the reference's own tolerance for this estimate is atol = 1e-2, tests/test_connectivity.py; both sides stop at rtol = 5e-6)
    err = float(np.abs(got[0][:, off] - ref[0][:, off]).max())
    err_away = float(np.abs(got[0][3:][:, off] - ref[0][3:][:, off]).max())
    info = np.array([gr.info["converged"], gr.info["max rel. err"], gr.info["reg. factor"], gr.info["initial cond. num"]], dtype=float)
    say("connectivityanalysis(granger, tapsmofrq=3) on the 5-channel network: max |diff| to conn5.npz[granger] = %.3g (%.3g "
        "away from the three bins next to DC); .info = converged %s, max rel. err %.3g, reg. factor %s, initial cond. num "
        "%.6g (fixture: %s)" % (err, err_away, bool(info[0]), info[1], info[2], info[3], g5["granger_info"].tolist()))
    # Wilson's iteration converges quadratically and stops at the first error below rtol = 5e-6: here the fixture's run
    # stopped at 4.0e-6 and this one - spectra from another NumPy build, 1e-7 apart - took one more step to 4e-13; the
    # two estimates then differ at the bins next to DC by less than the reference's own tolerance for them (atol 1e-2,
    # syncopy/tests/test_connectivity.py) and agree elsewhere
    assert err <= 1e-2 and err_away <= 2e-3 and bool(info[0]) and info[2] == g5["granger_info"][2]
    assert abs(info[3] - g5["granger_info"][3]) <= 1e-4 * g5["granger_info"][3]

    # ---- time-frequency routines: per-trial positional arguments (soi / preselect, postselect: computational_routine.py:
    # 985-990 hands argv[k][trial] to the cF), the dry run on the reference's FauxTrial, _make_trialdef in process_metadata
    gt = np.load(os.path.join(ROOT, "tests", "golden", "tf_variants.npz"))
    tf = spy.AnalogData(data=[np.array(t) for t in gt["data"]], samplerate=float(gt["samplerate"]))
    tf.trialdefinition = gt["trialdefinition"]
    tf_cases = [
        ("conv_hann_half", dict(method="mtmconvol", taper="hann", t_ftimwin=0.5, toi=0.5)),
        ("conv_dpss_keep", dict(method="mtmconvol", tapsmofrq=2, t_ftimwin=0.4, toi=0.5, keeptapers=True, output="fourier", foilim=[0, 120])),
        ("conv_all", dict(method="mtmconvol", taper="hann", t_ftimwin=0.1, toi="all", foi=[20, 40, 60], polyremoval=1)),
        ("conv_toi_equi", dict(method="mtmconvol", taper="hann", t_ftimwin=0.05, toi=np.arange(-0.5, 0.5, 0.01))),
        ("conv_toi_irreg", dict(method="mtmconvol", taper="hann", t_ftimwin=0.3, toi=np.array([-0.6, -0.45, 0.0, 0.31]))),
        ("wav_all", dict(method="wavelet", wavelet="Morlet", width=6, foi=np.arange(10, 110, 10), toi="all")),
        ("wav_toi", dict(method="wavelet", wavelet="Morlet", width=4, foi=np.array([8.0, 33.0, 150.0]), toi=np.arange(-0.8, 0.8, 0.05), output="fourier")),
        ("wav_auto_scales", dict(method="wavelet", wavelet="Morlet", toi="all", output="abs", keeptrials=False)),
    ]
    for name, opts in tf_cases:
        r = spy.freqanalysis(tf, **opts)
        got = np.array(r.data)
        # conv_all detrends linearly: the product fits in float64, the reference in float32 - the
        # fixture's own distance from the exact fit is the tolerance there, as in tests/test_gpu_golden.py
        e = excess(got, gt[name], rtol=2e-5 if name == "conv_all" else 1e-5, atol_rel=4e-6 if name == "conv_all" else 1e-6)
        worst = max(worst, e)
        ok_axes = bool(np.array_equal(r.freq, gt[name + "_freq"]) and np.array_equal(r.trialdefinition, gt[name + "_trialdef"])
                       and np.allclose(np.array(r.time[0]), gt[name + "_time0"], rtol=0, atol=1e-12))
        say("freqanalysis(%s) on the time-frequency fixture: %s %s, err/tol vs tf_variants.npz[%s] = %.3g; freq axis, "
            "trialdefinition (_make_trialdef) and time axis equal: %s"
            % (", ".join("%s=%s" % (k, ("array(%d)" % len(v)) if isinstance(v, np.ndarray) else repr(v)) for k, v in opts.items()),
               got.shape, got.dtype, name, e, ok_axes))
        assert got.shape == gt[name].shape and e <= 1.0 and ok_axes, (name, got.shape, gt[name].shape, e)

    # ---- SpectralData input: freqanalysis(output="fourier", keeptapers=True) chained into connectivityanalysis
    # (SpectralDyadicProduct, ST_compRoutines.py:30-117, connectivity_analysis.py:475-538)
    gc = np.load(os.path.join(ROOT, "tests", "golden", "chain.npz"))
    n5j = spy.AnalogData(data=[np.array(t) for t in gc["data"]], samplerate=float(gc["samplerate"]))
    spec5 = spy.freqanalysis(n5j, method="mtmfft", tapsmofrq=3, output="fourier", keeptapers=True, foilim=[0, 60])
    e0 = excess(np.array(spec5.data[0:1]), gc["spec_first_trial"])
    for key, opts in (("chain_coh", dict(method="coh")), ("chain_coh_complex", dict(method="coh", output="complex")),
                      ("chain_csd", dict(method="csd"))):
        r = spy.connectivityanalysis(spec5, **opts)
        e = excess(np.array(r.data), gc[key])
        worst = max(worst, e, e0)
        say("connectivityanalysis(%s) on SpectralData (fourier, keeptapers; first trial err/tol %.3g): %s %s, err/tol vs chain.npz[%s] = %.3g"
            % (", ".join("%s=%r" % kv for kv in opts.items()), e0, r.data.shape, r.data.dtype, key, e))
        assert e <= 1.0 and e0 <= 1.0

    # ---- ONE analysis with parallel=True on a dask.distributed LocalCluster (the reference's own test setup:
    # syncopy/tests/conftest.py:41,60): the bound compute function is pickled to the workers, process_io takes its
    # dict-of-HDF5-coordinates branch there (kwarg_decorators.py:646-737: the worker opens the source file, reads the
    # trial's slab, calls the cF, writes the result under the target's lock), the workers start their own product server
    import dask.distributed as dd
    cluster = dd.LocalCluster(n_workers=2, threads_per_worker=1, processes=True, dashboard_address=None)
    client = dd.Client(cluster)
    try:
        sp = spy.freqanalysis(c1, method="mtmfft", tapsmofrq=2, parallel=True)
        e = excess(np.array(sp.data), g1["pow"])
        cp = spy.connectivityanalysis(c1, method="coh", tapsmofrq=2, parallel=True)
        e2 = excess(np.array(cp.data), g1["coh_abs"])
        wv = spy.freqanalysis(tf, method="wavelet", wavelet="Morlet", width=6, foi=np.arange(10, 110, 10), toi="all", parallel=True)
        e3 = excess(np.array(wv.data), gt["wav_all"])
        worst = max(worst, e, e2, e3)
        say("parallel=True on dd.LocalCluster(n_workers=2): freqanalysis(mtmfft) err/tol vs c1.npz[pow] = %.3g, "
            "connectivityanalysis(coh) vs c1.npz[coh_abs] = %.3g, freqanalysis(wavelet) vs tf_variants.npz[wav_all] = %.3g; "
            "log of the first: %s" % (e, e2, e3, "parallel" if "parallel" in str(sp.cfg).lower() or "parallel" in sp.log.lower() else "(no mention)"))
        assert e <= 1.0 and e2 <= 1.0 and e3 <= 1.0
    finally:
        client.close()
        cluster.close()

    stats = srv.call("__stats__")
    say("compute-function calls served by syncopy_amd: " + ", ".join("%s %d" % kv for kv in sorted(stats.items())))
    assert all(v > 0 for v in stats.values())
    say("(calls served inside the Dask workers went to the workers' own servers and are not in this count)")
    srv.close()
    say("drop-in check passed: worst err/tol over the spectral fixtures %.3g" % worst)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            fh.write("# oracle/check_dropin.sh: syncopy_amd's compute functions under the reference's own ComputationalRoutine engine\n")
            fh.write("\n".join(log) + "\n")


if __name__ == "__main__":
    if "--serve" in sys.argv:
        serve()
    else:
        main()
