"""
Minkowski distance operator (prograph/distance/minkowski.py:8-41 of the reference).

p = 2 on fp16 device tensors - what `build_graph(representation="Embedded", distance=minkowski)`
feeds it (prograph/prograph.py:726-764) - runs on the HIP kernel `pg_minkowski_dense`, which rounds
every elementwise step to fp16 exactly like the reference's fp16 tensor expression (see
prograph_amd/csrc/pg_mink.hip for the tolerance).  Any other p or dtype is evaluated with the
reference's own torch expression on the device the operands live on: the distance-operator protocol
stays open, and nothing here runs on a CPU path of its own.
"""
import torch

from .. import _native
from .utils import clean_input


def minkowski(X, Y, p=2, similarity=False):
    X, Y = clean_input(X, Y)
    # (the kernel takes up to 65 535 blocks of 16 rows of Y per launch; empty / wider operands: the torch expression)
    if (p == 2 and X.dtype == torch.float16 and Y.dtype == torch.float16 and X.is_cuda and Y.is_cuda and X.shape[1] > 0
            and X.shape[0] > 0 and 0 < Y.shape[0] <= 65535 * 16):
        return _native.minkowski_dense(_native.pack_f16(X), _native.pack_f16(Y), similarity=similarity)
    diff = X - Y[:, None, :]
    distances = torch.pow(torch.sum(torch.pow(diff, exponent=p), axis=2), exponent=1 / p)
    if similarity:
        distances = 1 / (1 + distances)
    return distances
