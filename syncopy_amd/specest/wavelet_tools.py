"""Morlet scale/period conversions (specest/wavelets/wavelets.py:89-101) and the
'optimal' scale set of Torrence & Compo as used by the reference (specest/wavelet.py:52-106)."""
import numpy as np


def morlet_scale_from_period(period, w0=6.0):
    return (period * (np.sqrt(w0 * w0 + 2) + w0)) / (4.0 * np.pi)


def morlet_fourier_period(s, w0=6.0):
    return 4 * np.pi * s / (w0 + (2 + w0 ** 2) ** 0.5)


# wavelet functions of the reference (freqanalysis.py:55): name -> (kernel family of the C ABI, order taken from the call?)
WAVELET_FAMILY = {"Morlet": ("Morlet", False), "Paul": ("Paul", True), "DOG": ("DOG", True), "Ricker": ("DOG", False),
                  "Marr": ("DOG", False), "Mexican_hat": ("DOG", False)}


def family_scale_from_period(family, order=None, w0=6.0):
    """period -> scale of a wavelet family (wavelets.py:93-101 Morlet, :181-185 Paul, :303-307 DOG)."""
    if family == "Paul":
        return lambda period: period * (2 * order + 1) / (4 * np.pi)
    if family == "DOG":
        return lambda period: period * np.sqrt(order + 0.5) / (2 * np.pi)
    return lambda period: morlet_scale_from_period(period, w0)


def family_fourier_period(family, order=None, w0=6.0):
    """scale -> Fourier period (wavelets.py:89-91, :178-179, :300-301)."""
    if family == "Paul":
        return lambda s: 4 * np.pi * s / (2 * order + 1)
    if family == "DOG":
        return lambda s: 2 * np.pi * s / (order + 0.5) ** 0.5
    return lambda s: morlet_fourier_period(s, w0)


def optimal_wavelet_scales(nSamples, dt, w0=6.0, dj=0.25, s0=None, scale_from_period=None):
    if s0 is None:
        s0 = morlet_scale_from_period(2 * dt, w0) if scale_from_period is None else scale_from_period(2 * dt)
    J = int((1 / dj) * np.log2(nSamples * dt / s0))
    return (s0 * 2 ** (dj * np.arange(0, J + 1)))[::-1]


def superlet_steps(scales, order_max, order_min=1, c_1=3, adaptive=False):
    """The factors of the superlet geometric mean as (cycles, s0, exponents) steps: the transform with `cycles`
    cycles over scales[s0:] enters the product raised to `exponents` (one per scale of that transform; the first step
    initialises).  Multiplicative SLT (specest/superlet.py:97-117): every order over all scales, exponent
    1/n_orders.  Fractional adaptive SLT (:120-182): the order grows linearly with frequency from order_min to
    order_max (:378-395); integer part i of the order -> product of the first i wavelets, the next one enters with
    the fractional part as weight, everything to the power 1/(order - order_min + 1)."""
    scales = np.asarray(scales, dtype=np.float64)
    if not adaptive:
        cycles = c_1 * np.arange(order_min, order_max + 1)
        n_ord = order_max + 1 - order_min
        return [(float(c), 0, np.full(scales.size, 1 / n_ord)) for c in cycles]
    fois = 1 / (2 * np.pi * scales)
    orders = order_min + (order_max - order_min) * (fois - fois[0]) / (fois[-1] - fois[0])
    orders_int = np.int32(np.floor(orders))
    cycles = c_1 * np.unique(orders_int)
    exponents = 1 / (orders - order_min + 1)
    jumps = np.where(np.diff(orders_int))[0]
    if len(cycles) != len(jumps) + 1:
        raise ValueError("superlet orders are not monotonous in the scales (frequencies must be sorted low to high)")
    alphas = orders % orders_int
    steps = [(float(cycles[0]), 0, exponents.copy())]
    last = 1
    for i, jump in enumerate(jumps):
        span = slice(last, jump + 1)
        expo = np.concatenate((alphas[span] * exponents[span], exponents[jump + 1:]))
        steps.append((float(cycles[i + 1]), int(last), expo))
        last = int(jump) + 1
    return steps
