"""syncopy_amd: MI355X-native spectral estimation / cross-spectral connectivity hot path
behind Syncopy's ComputationalRoutine / computeFunction plug-in surface.

    import syncopy_amd as spy
    spec = spy.freqanalysis(data, method="mtmfft", tapsmofrq=2)
    coh = spy.connectivityanalysis(data, method="coh", tapsmofrq=2)

All arithmetic runs in libspyhip.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
__version__ = "0.1.0"

from .datatype import AnalogData, CrossSpectralData, SpectralData  # noqa: F401
from .specest.freqanalysis import freqanalysis  # noqa: F401
from .connectivity.connectivity_analysis import connectivityanalysis  # noqa: F401
from . import synthdata  # noqa: F401
from .shared.kwarg_decorators import StructDict, get_defaults  # noqa: F401
