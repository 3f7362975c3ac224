"""Index matching helpers of the frontends."""
import numpy as np

from .errors import SPYValueError


def best_match(source, selection, span=False, tol=None, squash_duplicates=False):
    """Closest elements of `source` for every entry of `selection` (or all elements inside the
    closed interval `selection` if `span`).  Behaviour of syncopy/shared/tools.py:224-343:
    ties go to the right neighbour, duplicates are removed keeping the first occurrence."""
    source = np.asarray(source)
    if np.issubdtype(type(selection), np.number):
        selection = [selection]
    selection = np.asarray(selection, dtype=float)
    if tol is not None:
        if not all(np.all(np.abs(source - v) < tol) for v in selection):
            raise SPYValueError(f"all elements of `selection` within a {tol:2.4f}-band around `source`",
                                varname="selection", actual="values deviating further")
    if span:
        idx = np.nonzero((source >= selection[0]) & (source <= selection[1]))[0]
        order = None
    else:
        order = None
        src = source
        if source.size > 1 and np.diff(source).min() < 0:
            order = np.argsort(source)
            src = source[order]
        idx = np.searchsorted(src, selection, side="left")
        lo = np.abs(selection - src[np.maximum(idx - 1, 0)])
        hi = np.abs(selection - src[np.minimum(idx, src.size - 1)])
        shift = (idx == src.size) | (lo < hi)
        idx[shift] -= 1
    if squash_duplicates:
        _, first = np.unique(idx.astype(np.intp), return_index=True)
        idx = idx[np.sort(first)]
    if order is not None:
        idx = order[idx]
    return source[idx], idx
