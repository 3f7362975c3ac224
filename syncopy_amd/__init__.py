"""syncopy_amd: MI355X-native spectral estimation / cross-spectral connectivity hot path
behind Syncopy's ComputationalRoutine / computeFunction plug-in surface."""
__version__ = "0.1.0"
