"""
prograph_amd — MI355X-native graph-construction hot path of acmater/prograph.

`Prograph(file, ...)`, `build_graph(eps=... | k=...)`, `indexing(...)`, `pgraph("sklearn" |
"pytorch", ...)` and the `distance(X, Y) -> (M, N)` operator protocol keep the reference's
surface; pairwise Hamming, epsilon/kNN adjacency assembly and the distance-k / position-mask
queries run as hand-written HIP kernels for gfx950 behind the C ABI in include/prograph_hip.h.
"""
from .prograph import Prograph
from .protein import Protein
from . import distance
from .graph import CSRGraph, KNNGraph

__all__ = ["Prograph", "Protein", "distance", "CSRGraph", "KNNGraph"]
