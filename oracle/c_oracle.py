"""ctypes wrapper of oracle/_build/liboracle.so — TEST INFRASTRUCTURE (see oracle/oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        # a bounded team: the GPU boxes show hundreds of CPUs but grant a share of them
        os.environ.setdefault("OMP_NUM_THREADS", str(min(32, os.cpu_count() or 1)))
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def hamming(X, Y):
    X = np.ascontiguousarray(X, dtype=np.uint8); Y = np.ascontiguousarray(Y, dtype=np.uint8)
    out = np.empty((Y.shape[0], X.shape[0]), dtype=np.int64)
    lib().orc_hamming(_p(X), ctypes.c_int64(X.shape[0]), _p(Y), ctypes.c_int64(Y.shape[0]), ctypes.c_int(X.shape[1]), _p(out))
    return out


def eps_csr(T, cmp, eps, row0=0, nrows=None, fast=False):
    """`fast`: the vectorised leg (orc_eps_fast, pinned to the scalar one by tests/test_oracle.py) for full-size checks."""
    T = np.ascontiguousarray(T, dtype=np.uint8)
    n, l = T.shape
    nrows = n - row0 if nrows is None else nrows
    counts = np.zeros(nrows, dtype=np.int64)
    args = (_p(T), ctypes.c_int64(n), ctypes.c_int(l), ctypes.c_int64(row0), ctypes.c_int64(nrows), ctypes.c_int(cmp),
            ctypes.c_double(eps))
    f = lib().orc_eps_fast if fast else lib().orc_eps
    f(*args, _p(counts), None, None, None)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    idx = np.empty(max(int(indptr[-1]), 1), dtype=np.int32); w = np.empty(max(int(indptr[-1]), 1), dtype=np.uint8)
    f(*args, _p(counts), _p(indptr), _p(idx), _p(w))
    return indptr, idx[:indptr[-1]], w[:indptr[-1]]


def knn(T, k, row0=0, nrows=None, fast=False):
    T = np.ascontiguousarray(T, dtype=np.uint8)
    n, l = T.shape
    nrows = n - row0 if nrows is None else nrows
    idx = np.empty((nrows, k), dtype=np.int32); d = np.empty((nrows, k), dtype=np.uint8)
    f = lib().orc_knn_fast if fast else lib().orc_knn
    f(_p(T), ctypes.c_int64(n), ctypes.c_int(l), ctypes.c_int64(row0), ctypes.c_int64(nrows), ctypes.c_int(k), _p(idx), _p(d))
    return idx, d


def synth(n, l, seed, members=256):
    out = np.empty((n, l), dtype=np.uint8)
    lib().orc_synth(ctypes.c_int64(n), ctypes.c_int(l), ctypes.c_uint64(seed), ctypes.c_int64(members), _p(out))
    return out


def lev_pair(a, b, band):
    a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
    la = int(np.argmin(np.append(a, 0) != 0)); lb = int(np.argmin(np.append(b, 0) != 0))
    f = lib().orc_lev_pair
    f.restype = ctypes.c_int
    return int(f(_p(a), ctypes.c_int(la), _p(b), ctypes.c_int(lb), ctypes.c_int(band)))


def lev_knn(T, k, band=8, row0=0, nrows=None):
    T = np.ascontiguousarray(T, dtype=np.uint8)
    n, l = T.shape
    nrows = n - row0 if nrows is None else nrows
    idx = np.empty((nrows, k), dtype=np.int32); d = np.empty((nrows, k), dtype=np.uint8)
    lib().orc_lev_knn(_p(T), ctypes.c_int64(n), ctypes.c_int(l), ctypes.c_int64(row0), ctypes.c_int64(nrows),
                      ctypes.c_int(band), ctypes.c_int(k), _p(idx), _p(d))
    return idx, d
