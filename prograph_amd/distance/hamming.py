"""
Hamming distance operator — the reference's plug-in contract
`distance(X (N,D), Y (M,D), similarity=False) -> (M,N)` (prograph/distance/hamming.py:8-39),
computed by the HIP dense kernel (`pg_hamming_dense`) instead of the broadcast
`torch.sum(X != Y[:,None,:], axis=2)`.

Same conventions as the reference: the result is (M, N) — row m is Y[m] against every row
of X (the reference's docstring says N x M, its code and tests say M x N) — `torch.int64`,
on the device of the inputs; `ValueError` on an empty operand; zero right-padding when the
second dimensions differ; `similarity=True` returns 1/(1+d).

Byte tokens (integer valued, 0..255) take the native kernel at any D: one record holds 255
tokens (5 bit planes) or 128 (8 planes); longer sequences are cut into column segments whose
distances the kernel accumulates into the same (M, N) matrix.  Anything else is not tokenised
sequence data (floats with fractions, wide integers) and is evaluated with the same torch
expression as the reference, on the GPU; there is no CPU path.
"""
import weakref

import torch

from .. import _native
from .utils import clean_input

# The reference calls distance(X, batch) once per batch of 8 rows with the SAME X
# (prograph/prograph.py:731-732): remember, for the last few X operands that live on the device,
# whether they are byte tokens, their largest token and their packed form, so that such loops
# validate and pack the big matrix once.  An entry belongs to ONE tensor object: it holds a weak
# reference to it and only answers for that very object at the same in-place version (storage
# addresses are reused by the caching allocator, so address + shape alone would hand a freed
# operand's distances to the next tensor allocated in its place).
_X_CACHE = {}
_X_CACHE_MAX = 4


def _x_entry(X, Xd, cacheable):
    key = (X.data_ptr(), tuple(X.shape), X.dtype, str(X.device)) if cacheable else None
    for k in [k for k, e in _X_CACHE.items() if e["ref"]() is None]:
        del _X_CACHE[k]                                    # operands that died
    ent = _X_CACHE.get(key) if key is not None else None
    if ent is not None and (ent["ref"]() is not X or ent["version"] != X._version):
        del _X_CACHE[key]                                  # another tensor at that address, or edited in place
        ent = None
    if ent is None:
        xb = _as_byte_tokens(Xd)
        ent = {"xb": xb, "max": int(xb.max()) if xb is not None else None, "planes": {}}
        if key is not None:
            ent["ref"], ent["version"] = weakref.ref(X), X._version
            if len(_X_CACHE) >= _X_CACHE_MAX:
                _X_CACHE.pop(next(iter(_X_CACHE)))
            _X_CACHE[key] = ent
    return ent


def _segments(d, bits):
    w = _native.MAX_L_5BIT if bits == _native.BITS_5 else _native.MAX_L
    if d <= w:
        return [(0, d)]
    w = w // 32 * 32                              # whole 32-token groups per segment
    return [(a, min(d, a + w)) for a in range(0, d, w)]


def _x_planes(ent, bits, seg):
    key = (bits, seg)
    if key not in ent["planes"]:
        ent["planes"][key] = _native.pack(ent["xb"][:, seg[0]:seg[1]], bits=bits)
    return ent["planes"][key]


def _as_byte_tokens(T):
    """uint8 view of an integer valued tensor in 0..255, or None."""
    if T.dtype == torch.uint8:
        return T
    if T.dtype == torch.bool:
        return T.to(torch.uint8)
    if T.is_floating_point():
        ok = torch.isfinite(T).all() and (T == T.floor()).all() and (T >= 0).all() and (T <= 255).all()
    else:
        ok = (T >= 0).all() and (T <= 255).all()
    return T.to(torch.uint8) if bool(ok) else None


def hamming(X, Y, similarity=False):
    X, Y = clean_input(X, Y)
    home = X.device
    dev = _native.device()
    Xd, Yd = X.to(dev), Y.to(dev)
    # only a caller-owned tensor that already lives on the device has a stable identity
    ent = _x_entry(X, Xd, X.device == dev and X.shape[0] >= 1024)
    yb = _as_byte_tokens(Yd) if ent["xb"] is not None else None
    bits = None
    if yb is not None:
        bits = _native.BITS_5 if max(ent["max"], int(yb.max())) <= 31 else _native.BITS_8
        distances = None
        for seg in _segments(X.shape[1], bits):
            distances = _native.hamming_dense(_x_planes(ent, bits, seg), _native.pack(yb[:, seg[0]:seg[1]], bits=bits),
                                              out_bytes=8, out=distances)
    else:
        distances = torch.sum(Xd != Yd[:, None, :], axis=2)
    if similarity:
        distances = 1 / (1 + distances)
    return distances.to(home)
