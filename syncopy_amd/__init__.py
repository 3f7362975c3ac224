"""syncopy_amd: MI355X-native spectral estimation / cross-spectral connectivity hot path
behind Syncopy's ComputationalRoutine / computeFunction plug-in surface.

    import syncopy_amd as spy
    spec = spy.freqanalysis(data, method="mtmfft", tapsmofrq=2)
    coh = spy.connectivityanalysis(data, method="coh", tapsmofrq=2)

All arithmetic runs in libspyhip.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
__version__ = "0.1.0"

from .datatype import AnalogData, CrossSpectralData, SpectralData  # noqa: F401
from . import synthdata  # noqa: F401
from .io import load, save  # noqa: F401
from .shared.kwarg_decorators import StructDict, get_defaults  # noqa: F401

# The front ends keep their tensors in PyTorch (device memory, streams, torch.distributed) and are imported on first
# use; `syncopy_amd.abi` drives the same library with NumPy + ctypes only and never pulls torch in.
_LAZY = {"freqanalysis": ".specest.freqanalysis", "connectivityanalysis": ".connectivity.connectivity_analysis",
         "mean": ".statistics.summary_stats"}


def release_device_buffers():
    """Free the device / pinned buffers the package caches between calls (spectra hand-over buffer, staging buffers)."""
    from . import backend
    backend.release_buffers()


def __getattr__(name):
    if name in _LAZY:
        import importlib
        value = getattr(importlib.import_module(_LAZY[name], __name__), name)
        globals()[name] = value
        return value
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
