"""Small helpers (prograph/utils/general.py of the reference)."""
import numpy as np


def flatten(lst):
    out = []
    for sub in lst:
        out.extend(sub)
    return out


def check_symmetric(a, rtol=1e-05, atol=1e-08):
    return np.allclose(a, a.T, rtol=rtol, atol=atol)
