"""
Minkowski distance operator (prograph/distance/minkowski.py:8-41 of the reference).

Out of the hot-path scope: a dense floating-point contraction kept as the
reference's own torch expression so that `build_graph(distance=minkowski, ...)` keeps working
through the generic distance protocol.  Runs on whatever device the operands live on.
"""
import torch

from .utils import clean_input


def minkowski(X, Y, p=2, similarity=False):
    X, Y = clean_input(X, Y)
    diff = X - Y[:, None, :]
    distances = torch.pow(torch.sum(torch.pow(diff, exponent=p), axis=2), exponent=1 / p)
    if similarity:
        distances = 1 / (1 + distances)
    return distances
