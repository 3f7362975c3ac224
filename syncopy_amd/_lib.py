"""ctypes binding of libspyhip.so (declarations follow include/spyhip.h).

There is deliberately NO fallback: if the HIP library is missing or cannot be
loaded, every compute entry point raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libspyhip.so")

_lib = None

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol of include/spyhip.h
SIGNATURES = {
    "spyhip_version": (C.c_int, []),
    "spyhip_last_error": (C.c_char_p, []),
    "spyhip_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "spyhip_ctx_destroy": (C.c_int, [vp]),
    "spyhip_ctx_set_stream": (C.c_int, [vp, vp]),
    "spyhip_ctx_trim": (C.c_int, [vp]),
    "spyhip_ctx_synchronize": (C.c_int, [vp]),
    "spyhip_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "spyhip_free": (C.c_int, [vp, vp]),
    "spyhip_memset": (C.c_int, [vp, vp, C.c_int, C.c_size_t]),
    "spyhip_upload": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "spyhip_download": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "spyhip_queue_upload": (C.c_int, [vp, vp, C.c_int64, C.c_int, c_i64p, C.c_int, C.POINTER(vp)]),
    "spyhip_queue_destroy": (C.c_int, [vp]),
    "spyhip_queue_data": (vp, [vp]),
    "spyhip_queue_segments": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]),
    "spyhip_comm_unique_id": (C.c_int, [vp]),
    "spyhip_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "spyhip_comm_destroy": (C.c_int, [vp]),
    "spyhip_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "spyhip_allreduce_csd": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "spyhip_allreduce": (C.c_int, [vp, vp, C.c_int64, C.c_int]),
    "spyhip_fft_plan_set_precision": (C.c_int, [vp, C.c_int]),
    "spyhip_cwt_plan_set_precision": (C.c_int, [vp, C.c_int]),
    "spyhip_cwt_plan_set_direct": (C.c_int, [vp, C.c_int]),
    "spyhip_fft_plan_set_reference_mean": (C.c_int, [vp, C.c_int]),
    "spyhip_fft_plan_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, C.c_double, C.c_int,
                                         C.c_int, c_i32p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "spyhip_fft_plan_destroy": (C.c_int, [vp]),
    "spyhip_fft_exec": (C.c_int, [vp, vp, C.c_int64, vp, vp, vp, vp, C.c_int, vp]),
    "spyhip_fft_plan_kernel_name": (C.c_char_p, [vp]),
    "spyhip_fft_plan_set_blocked": (C.c_int, [vp, C.c_int]),
    "spyhip_csd_accumulate": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "spyhip_csd_set_phase_exact": (C.c_int, [vp, C.c_int]),
    "spyhip_csd_accumulate_split": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp]),
    "spyhip_csd_accumulate_split_range": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int]),
    "spyhip_csd_split_fallbacks": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "spyhip_fft_plan_set_absmax": (C.c_int, [vp, vp]),
    "spyhip_csd_accumulate_blocked": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "spyhip_coh_from_accumulator": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, vp]),
    "spyhip_ppc_accumulate": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "spyhip_ppc_accumulate_csd": (C.c_int, [vp, vp, C.c_int, C.c_int64, vp]),
    "spyhip_ppc_finalize": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, vp]),
    "spyhip_jack_coh_accumulate": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int64,
                                             vp, vp]),
    "spyhip_ccov_nfft": (C.c_int, [C.c_int]),
    "spyhip_ccov_normalize": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "spyhip_ccov_from_accumulator": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp]),
    "spyhip_csd_tril_pack": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "spyhip_csd_tril_unpack": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "spyhip_csd_finalize": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double]),
    "spyhip_coh_normalize": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "spyhip_cwt_plan_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_f64p, C.c_double, C.c_double, C.c_int,
                                         C.c_int, c_i32p, C.c_int, C.POINTER(vp)]),
    "spyhip_cwt_plan_create_sl": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_f64p, C.c_double, C.c_double,
                                            C.c_double, C.c_int, C.c_int, c_i32p, C.c_int, C.POINTER(vp)]),
    "spyhip_cwt_plan_create_family": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_f64p, C.c_double, C.c_int, C.c_double,
                                                C.c_double, C.c_int, C.c_int, c_i32p, C.c_int, C.POINTER(vp)]),
    "spyhip_slt_combine": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, C.c_int,
                                     C.c_int]),
    "spyhip_spec_convert": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "spyhip_cwt_plan_destroy": (C.c_int, [vp]),
    "spyhip_cwt_exec": (C.c_int, [vp, vp, C.c_int64, vp, vp, vp, vp, C.c_int, vp, C.c_int]),
    "spyhip_granger": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, vp, vp, vp,
                                 c_f64p]),
    "spyhip_granger_last_iterations": (C.c_int, [vp]),
    "spyhip_wilson_cond": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double, vp, vp, c_f64p]),
    "spyhip_wilson_init": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "spyhip_wilson_psi0": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "spyhip_wilson_g": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "spyhip_wilson_plus": (C.c_int, [vp, vp, C.c_int, C.c_int64, vp, vp]),
    "spyhip_wilson_update": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, c_f64p]),
    "spyhip_wilson_finish": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "spyhip_trial_mean_f32": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64]),
    "spyhip_trial_mean_c64": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64]),
    "spyhip_axis_nanmean": (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, vp]),
}


class SpyHipError(RuntimeError):
    pass


def load():
    """Load libspyhip.so (once)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpyHipError(
            f"{LIB_PATH} is missing: build it with `python -m syncopy_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # A host that uses PyTorch has libamdhip64 loaded already and ours binds to that runtime; a NumPy-only host
    # (syncopy_amd/abi.py) gets the runtime through the library's own dependencies.  Loading order matters when both
    # are used in one process: if torch is imported AFTER this library, torch binds to the system libamdhip64 this
    # library pulled in (same SONAME) instead of the runtime bundled with its wheel.  So: torch already imported ->
    # nothing to do; torch importable and not opted out (SPY_NO_TORCH=1, what the NumPy-only tests set) -> import it
    # first; otherwise the process is NumPy-only by declaration.
    import sys
    if "torch" not in sys.modules and not os.environ.get("SPY_NO_TORCH"):
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().spyhip_last_error().decode("utf-8", "replace")
        if rc == -6:
            # a Cholesky factorisation met a matrix that is not positive definite (granger.hip: check_info): the reference's
            # np.linalg.cholesky raises this type with this text (wilson_sf.py:76,144-151)
            import numpy as np
            raise np.linalg.LinAlgError("Matrix is not positive definite") from SpyHipError(f"{what} failed (code {rc}): {msg}")
        raise SpyHipError(f"{what} failed (code {rc}): {msg}")
