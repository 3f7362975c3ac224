"""Module-wide constants (values of syncopy/shared/const_def.py:12-62)."""
import numpy as np
from scipy.signal import windows

spectralDTypes = {
    "pow": np.float32, "abs": np.float32, "real": np.float32, "imag": np.float32, "angle": np.float32,
    "absreal": np.float32, "absimag": np.float32, "fourier": np.complex64, "complex": np.complex64,
}

availableTapers = [w for w in windows.__all__ if w not in ("get_window", "exponential", "dpss")]
availablePaddingOpt = ["maxperlen", "nextpow2"]
availableMethods = ("mtmfft", "mtmconvol", "wavelet", "superlet", "welch")
connectivityMethods = ("coh", "corr", "csd", "granger", "ppc")
connectivity_outputs = {"abs", "pow", "complex", "fourier", "angle", "real", "imag"}
