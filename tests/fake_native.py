"""
TEST-ONLY stand-in for prograph_amd._native so that the HOST logic (argument handling, result
containers, set logic around the kernels, sharding arithmetic) can be exercised on a box
without a GPU.  It answers every native call with the oracle (oracle/prograph_oracle.py) on
CPU tensors.  Nothing under prograph_amd/ imports this file; the product has no CPU path.
"""
import operator

import numpy as np
import torch

from oracle import prograph_oracle as O
from prograph_amd import _native

_OPS = {_native.CMP_LE: operator.le, _native.CMP_LT: operator.lt, _native.CMP_EQ: operator.eq,
        _native.CMP_GE: operator.ge, _native.CMP_GT: operator.gt}


class FakePlanes:
    def __init__(self, tok, bits):
        self.tok = np.ascontiguousarray(tok).astype(np.int64)
        self.n, self.l = self.tok.shape
        self.npad, self.g, self.q, self.bits = _native.npad(self.n), _native.ngroups(self.l), _native.nchunks(self.l, bits), bits
        self.buf = torch.zeros(1, dtype=torch.uint8)


def _pack(tokens, rows=None, bits=None, width=None):
    t = tokens.cpu().numpy() if isinstance(tokens, torch.Tensor) else np.asarray(tokens)
    if t.ndim != 2:
        raise ValueError("token matrix must be 2-D")
    if not np.issubdtype(t.dtype, np.integer):
        raise TypeError("integer tokens expected")
    if width is not None:
        t = _pad_to(t, width)
    if t.shape[1] > _native.MAX_L_5BIT:
        raise ValueError("L exceeds the native limit")
    if rows is not None:
        t = t[np.asarray(rows)]
    if t.size and (t.min() < 0 or t.max() > 255):
        raise ValueError("tokens outside 0..255")
    if t.shape[0] == 0:
        raise ValueError("empty token matrix")
    bits = bits or (8 if t.size and t.max() > 31 else 5)
    if t.size and t.max() >= (1 << bits):
        raise ValueError("tokens do not fit the bit planes")
    if t.shape[1] > (_native.MAX_L_5BIT if bits == 5 else _native.MAX_L):
        raise ValueError("L exceeds the native limit for these bit planes")
    return FakePlanes(t, bits)


def _pad_to(a, l):
    out = np.zeros((a.shape[0], l), dtype=a.dtype)
    out[:, :a.shape[1]] = a
    return out


def _dense(xp, yp, out_bytes=8, out=None):
    l = max(xp.l, yp.l)
    d = O.hamming(_pad_to(xp.tok, l), _pad_to(yp.tok, l))
    if out is not None:
        out += d.to(out.dtype)
        return out
    return d.to({1: torch.uint8, 2: torch.float16, 4: torch.int32, 8: torch.int64}[out_bytes])


def _window(rp, cp, row0, nrows):
    nrows = rp.n - row0 if nrows is None else nrows
    l = max(rp.l, cp.l)
    return _pad_to(rp.tok, l)[row0:row0 + nrows], _pad_to(cp.tok, l), nrows


def _eps_graph(rp, cp, cmp, eps, row0=0, nrows=None, cap=256):
    rows, cols, nrows = _window(rp, cp, row0, nrows)
    d = O.hamming(cols, rows).numpy()
    mask = _OPS[cmp](d, eps) & (d > 0)
    counts = mask.sum(1)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    r, c = np.nonzero(mask)
    return (torch.from_numpy(indptr), torch.from_numpy(c.astype(np.int32)), torch.from_numpy(d[r, c].astype(np.uint8)))


def _knn_graph(rp, cp, k, row0=0, nrows=None, out=None):
    if not 1 <= k <= _native.MAX_K_ROUNDS:
        raise RuntimeError("k out of range")
    rows, cols, nrows = _window(rp, cp, row0, nrows)
    d = O.hamming(cols, rows)
    s = torch.sort(d, dim=1, stable=True)
    idx = torch.full((nrows, k), -1, dtype=torch.int32)
    dist = torch.full((nrows, k), 255, dtype=torch.uint8)
    kk = min(k, cp.n - 1)
    idx[:, :kk] = s[1][:, 1:kk + 1].to(torch.int32)
    dist[:, :kk] = s[0][:, 1:kk + 1].to(torch.uint8)
    return idx, dist


def _index_flags(planes, ref, want=None, pos_mode=0, pos_mask=None, not_mask=None, want_dist_out=True,
                 want_hist=True, want_flags=True):
    tok = planes.tok
    mut = tok != tok[ref]
    d = mut.sum(1)
    ok = np.ones(planes.n, dtype=bool)
    if want is not None:
        ok &= np.isin(d, np.asarray(list(want)))
    if pos_mode:
        pm = np.zeros(planes.l, dtype=bool); pm[list(pos_mask)] = True
        nm = np.zeros(planes.l, dtype=bool); nm[list(not_mask)] = True
        anyp = (mut & pm).any(1)
        allp = (mut | ~pm).all(1)
        ok &= (anyp if pos_mode == 1 else allp) & ~(mut & nm).any(1)
    return (torch.from_numpy(d.astype(np.uint8)) if want_dist_out else None,
            torch.from_numpy(np.bincount(d, minlength=256).astype(np.int64)) if want_hist else None,
            torch.from_numpy(ok.astype(np.uint8)) if want_flags else None)


def _compact_flags(flags):
    return torch.nonzero(flags).reshape(-1).to(torch.int64)


def _csr_row_stats(indptr, indices, weights, f=None, want=("deg",), row0=0, ncols=None):
    ip = indptr.numpy(); ix = indices.numpy().astype(np.int64)
    w = np.ones(len(ix)) if weights is None else weights.numpy().astype(np.float64)
    rows = np.repeat(np.arange(len(ip) - 1), np.diff(ip))
    fv = None if f is None else f.numpy().astype(np.float64)
    out = {}
    for key in want:
        if key == "col_sum":
            acc = np.zeros(int(ncols))
            np.add.at(acc, ix, w)
        else:
            acc = np.zeros(len(ip) - 1)
            vals = {"deg": lambda: w, "sum_f": lambda: fv[ix], "sum_wf": lambda: w * fv[ix],
                    "self_w": lambda: np.where(ix == rows + row0, w, 0.0)}[key]()
            np.add.at(acc, rows, vals)
        out[key] = torch.from_numpy(acc)
    return out


def _f16_knn(block, k, first=1, descending=False):
    s = torch.sort(block.to(torch.float32), dim=1, descending=bool(descending), stable=True)
    return s[1][:, first:first + k].to(torch.int32), s[0][:, first:first + k].to(torch.float16)


def _f16_eps(block, cmp, eps, similarity=False):
    d = block.to(torch.float32)
    e = float(np.float16(eps))
    keep = (_OPS[cmp](e, d) & (d < 1)) if similarity else (_OPS[cmp](d, e) & (d > 0))
    counts = keep.sum(dim=1)
    indptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
    rows, cols = torch.where(keep)
    return indptr, cols.to(torch.int32), block[rows, cols]


def install(monkeypatch):
    monkeypatch.setattr(_native, "device", lambda: torch.device("cpu"))
    monkeypatch.setattr(_native, "pack", _pack)
    monkeypatch.setattr(_native, "hamming_dense", _dense)
    monkeypatch.setattr(_native, "eps_graph", _eps_graph)
    monkeypatch.setattr(_native, "knn_graph", _knn_graph)
    monkeypatch.setattr(_native, "index_flags", _index_flags)
    monkeypatch.setattr(_native, "compact_flags", _compact_flags)
    monkeypatch.setattr(_native, "csr_row_stats", _csr_row_stats)
    monkeypatch.setattr(_native, "f16_knn", _f16_knn)
    monkeypatch.setattr(_native, "f16_eps", _f16_eps)
