"""
Banded Levenshtein kNN over tokenised, zero right-padded sequences — NOT in the reference
(acmater/prograph ships only hamming / minkowski); this is BASELINE.json configs[4], defined by
this build: d = min(edit distance, band+1), canonical (d, index) order, rank 0 dropped.
Parity is unpinned: the kernels are checked against the build's own oracle only.
"""
from .. import _native


def levenshtein_knn(tokens, k, band=8, **kw):
    return _native.levenshtein_knn(tokens, k, band=band, **kw)
