"""
ctypes binding of libprograph_hip.so (include/prograph_hip.h) — the only way the package
reaches the GPU for the hot path.  torch tensors are used as device storage and for the
stream handle only.

There is NO CPU or eager fallback for the native entry points: if the shared library is
missing, was built for another ABI, or no HIP device is visible, every call raises
`NativeUnavailable` (a RuntimeError) with the reason.
"""
import ctypes
import os
import threading

import numpy as np
import torch   # must be imported before the library is loaded: see _load()

_HERE = os.path.dirname(os.path.abspath(__file__))
# PROGRAPH_HIP_LIB: load another build of the same ABI (kernel A/B comparisons, tools/ab.py)
LIB_PATH = os.environ.get("PROGRAPH_HIP_LIB") or os.path.join(_HERE, "libprograph_hip.so")
ABI_VERSION = 3

BITS_5, BITS_8 = 5, 8
CMP_LE, CMP_LT, CMP_EQ, CMP_GE, CMP_GT = 0, 1, 2, 3, 4
MAX_L, MAX_L_5BIT, MAX_K, MAX_K_ROUNDS, MAX_N_KNN, LEV_MAX_BAND = 128, 255, 63, 1023, 1 << 24, 8

# every symbol include/prograph_hip.h declares (tests check the library exports them all)
SYMBOLS = [
    "pg_version", "pg_last_error", "pg_device_info", "pg_npad", "pg_ngroups", "pg_nchunks", "pg_planes_bytes", "pg_workspace_bytes",
    "pg_pack_planes", "pg_pack_bytes",
    "pg_hamming_dense", "pg_eps_slots", "pg_scan_scratch_bytes", "pg_exclusive_scan",
    "pg_eps_compact", "pg_eps_fill_rows", "pg_eps_slots_sym", "pg_eps_compact_sym", "pg_knn_hamming", "pg_knn_hamming_round", "pg_index_flags", "pg_compact_flags",
    "pg_lev_profile", "pg_lev_candidates", "pg_lev_candidates_sym", "pg_lev_knn", "pg_csr_row_stats",
    "pg_comm_available", "pg_comm_unique_id", "pg_comm_init", "pg_comm_destroy", "pg_allgather_tokens",
    "pg_f16_nchunks", "pg_pack_f16", "pg_minkowski_dense", "pg_f16_knn", "pg_f16_eps_count", "pg_f16_eps_fill",
]


class NativeUnavailable(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()

_i64, _i32, _vp, _dbl = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_double


def _load():
    """dlopen the library AFTER torch: hipcc stamps NEEDED libamdhip64.so.7 into it and torch
    ships a runtime with the same SONAME, so the loader binds us to the HIP runtime torch has
    already mapped.  One runtime per process is what makes torch-allocated pointers valid in
    our launches (SURVEY.md §7 hard part 1)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeUnavailable(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C prograph_amd/csrc` (hipcc, --offload-arch=gfx950)")
        try:
            lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:
            raise NativeUnavailable(f"cannot load {LIB_PATH}: {e}") from e
        missing = [s for s in SYMBOLS if not hasattr(lib, s)]
        if missing:
            raise NativeUnavailable(f"{LIB_PATH} does not export {missing}")
        lib.pg_version.restype = _i32
        if lib.pg_version() != ABI_VERSION:
            raise NativeUnavailable(f"ABI mismatch: library {lib.pg_version()}, binding {ABI_VERSION}")
        lib.pg_last_error.restype = ctypes.c_char_p
        lib.pg_npad.restype = _i64
        lib.pg_npad.argtypes = [_i64]
        lib.pg_ngroups.restype = _i32
        lib.pg_ngroups.argtypes = [_i32]
        lib.pg_nchunks.restype = _i32
        lib.pg_nchunks.argtypes = [_i32, _i32]
        lib.pg_planes_bytes.restype = _i64
        lib.pg_planes_bytes.argtypes = [_i64, _i32, _i32]
        lib.pg_scan_scratch_bytes.restype = _i64
        lib.pg_scan_scratch_bytes.argtypes = [_i64]
        lib.pg_workspace_bytes.restype = _i64
        lib.pg_workspace_bytes.argtypes = [_i64]
        lib.pg_device_info.argtypes = [ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.c_char_p, _i32]
        lib.pg_pack_planes.argtypes = [_vp, _i32, _i64, _i32, _i64, _vp, _i32, _vp, _i64, _vp, _vp]
        lib.pg_pack_bytes.argtypes = [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp]
        lib.pg_hamming_dense.argtypes = [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i32, _i64, _i32, _vp]
        lib.pg_eps_slots.argtypes = [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _dbl, _i32,
                                     _vp, _vp, _vp, _vp, _vp]
        lib.pg_exclusive_scan.argtypes = [_vp, _i64, _vp, _vp, _vp]
        lib.pg_eps_slots_sym.argtypes = [_vp, _i64, _i64, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _vp, _vp, _vp]
        lib.pg_eps_compact_sym.argtypes = [_vp, _i64, _i64, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _i32, _vp]
        lib.pg_eps_compact.argtypes = [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _dbl, _i32,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]
        lib.pg_eps_fill_rows.argtypes = [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _dbl,
                                         _vp, _vp, _vp, _vp, _vp, _vp]
        lib.pg_knn_hamming.argtypes = [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]
        lib.pg_knn_hamming_round.argtypes = [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp,
                                             _vp, _vp, _vp]
        lib.pg_index_flags.argtypes = [_vp, _i64, _i64, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]
        lib.pg_compact_flags.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp]
        lib.pg_csr_row_stats.argtypes = [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        lib.pg_lev_profile.argtypes = [_vp, _i64, _i32, _i64, _vp, _i64, _vp, _vp, _vp]
        lib.pg_lev_candidates.argtypes = [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp]
        lib.pg_lev_candidates_sym.argtypes = [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]
        lib.pg_lev_knn.argtypes = [_vp, _i64, _i32, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _vp]
        lib.pg_f16_nchunks.argtypes = [_i32]
        lib.pg_pack_f16.argtypes = [_vp, _i64, _i32, _i64, _vp, _vp, _i64, _vp]
        lib.pg_minkowski_dense.argtypes = [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp]
        lib.pg_f16_knn.argtypes = [_vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp]
        lib.pg_f16_eps_count.argtypes = [_vp, _i64, _i64, _i64, _i32, ctypes.c_float, _i32, _vp, _vp]
        lib.pg_f16_eps_fill.argtypes = [_vp, _i64, _i64, _i64, _i32, ctypes.c_float, _i32, _vp, _vp, _vp, _vp]
        lib.pg_comm_unique_id.argtypes = [_vp]
        lib.pg_comm_init.argtypes = [ctypes.POINTER(_vp), _i32, _i32, _vp]
        lib.pg_comm_destroy.argtypes = [_vp]
        lib.pg_allgather_tokens.argtypes = [_vp, _vp, _i64, _i32, _vp, _vp]
        for name in SYMBOLS:
            fn = getattr(lib, name)
            if fn.restype is ctypes.c_int and name not in ("pg_version",):
                fn.restype = _i32
        _lib = lib
        return lib


def lib():
    return _load()


def device():
    """The device the hot path runs on: the current CUDA(HIP) device of this process."""
    if not torch.cuda.is_available():
        raise NativeUnavailable("no HIP device visible to torch; the Hamming/graph path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _check(rc, what):
    if rc != 0:
        msg = lib().pg_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(0) if t is None else ctypes.c_void_p(t.data_ptr())


def workspace(nrows, dev):
    """Launch-private device state of one all-pairs call (pg_workspace_bytes): a fresh block per launch - torch's
    caching allocator hands a freed block out again only to later work on the same stream, which is exactly the
    reuse the ABI allows."""
    return torch.empty(int(lib().pg_workspace_bytes(int(nrows))), dtype=torch.uint8, device=dev)


def npad(n):
    return ((max(int(n), 1) + 255) // 256) * 256


def ngroups(l):
    return (max(int(l), 1) + 31) // 32


def nchunks(l, bits):
    return (ngroups(l) * int(bits) + 3) // 4


def planes_bytes(n, l, bits):
    """Chunk arrays + 64 bytes per sequence: signature section (MFMA column operand) + fold section."""
    return (nchunks(l, bits) * 16 + 64) * npad(n)


class Planes:
    """Device-resident token matrix as bit-sliced records (see include/prograph_hip.h).
    `flags`: the device word pg_pack_planes sets when a token does not fit the bit planes; `pack(check=False)`
    leaves reading it (one host sync) to the caller, `ensure_valid()` does it."""
    __slots__ = ("buf", "n", "l", "npad", "g", "q", "bits", "flags")

    def __init__(self, buf, n, l, bits, flags=None):
        self.buf, self.n, self.l, self.bits, self.flags = buf, int(n), int(l), int(bits), flags
        self.npad, self.g, self.q = npad(n), ngroups(l), nchunks(l, bits)

    def ensure_valid(self):
        if self.flags is not None and int(self.flags.item()):
            raise ValueError(f"tokens outside 0..{(1 << self.bits) - 1} cannot be packed with {self.bits} bit planes")
        self.flags = None

    @property
    def nbytes(self):
        return self.buf.numel()


def pack(tokens, rows=None, bits=None, width=None, check=True):
    """
    (N, L) integer tokens (torch tensor on the GPU, or anything np.asarray takes) -> Planes.
    `rows`: optional index list (the reference's `idxs`, prograph/prograph.py:726).
    `bits`: 5 or 8 bit planes per token; None picks 5 when every token is <= 31, else 8.
    `width`: pack as if the rows were zero right-padded to this length (clean_input's padding).
    Raises ValueError when a token does not fit a byte: such data is not "tokenized" and the
    caller must take the generic torch path.  `check=False` skips the host sync that reads the
    device-side validity word (callers in a pipeline read `Planes.flags` together with something
    else they wait for anyway, or call `ensure_valid()` later).
    """
    L = lib()
    dev = device()
    if not isinstance(tokens, torch.Tensor):
        tokens = torch.from_numpy(np.ascontiguousarray(np.asarray(tokens)))
    if tokens.dim() != 2:
        raise ValueError("token matrix must be 2-D")
    if tokens.dtype not in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
        raise TypeError(f"integer tokens expected, got {tokens.dtype}")
    if tokens.dtype == torch.int8:
        tokens = tokens.to(torch.int16)
    tokens = tokens.to(dev).contiguous()
    n_src, l = tokens.shape
    lw = l if width is None else int(width)
    if lw < l:
        raise ValueError("width smaller than the token matrix")
    if lw > MAX_L_5BIT:
        raise ValueError(f"L={lw} exceeds the native limit of {MAX_L_5BIT}")
    ridx = None
    n = n_src
    if rows is not None:
        ridx = torch.as_tensor(np.asarray(rows), dtype=torch.int64).reshape(-1)
        if ridx.numel() and (int(ridx.min()) < -n_src or int(ridx.max()) >= n_src):
            raise IndexError("row index out of range")
        ridx = torch.where(ridx < 0, ridx + n_src, ridx).to(dev)
        n = int(ridx.numel())
    if n == 0 or l == 0:
        raise ValueError("empty token matrix")
    if bits is None:
        lo, hi = int(tokens.min()), int(tokens.max())
        if lo < 0 or hi > 255:
            raise ValueError("tokens outside 0..255 cannot use the byte-token Hamming path")
        bits = BITS_5 if hi <= 31 else BITS_8
    if lw > (MAX_L_5BIT if bits == BITS_5 else MAX_L):
        raise ValueError(f"L={lw} exceeds the native limit for {bits} bit planes")
    np_ = npad(n)
    if lw != l:
        wide = torch.zeros((n_src, lw), dtype=tokens.dtype, device=dev)   # clean_input's zero right-padding
        wide[:, :l] = tokens
        tokens = wide
    buf = torch.empty(planes_bytes(n, lw, bits), dtype=torch.uint8, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    _check(L.pg_pack_planes(_ptr(tokens), tokens.element_size(), n, lw, tokens.stride(0), _ptr(ridx), int(bits),
                            _ptr(buf), np_, _ptr(flags), _stream()), "pg_pack_planes")
    planes = Planes(buf, n, lw, bits, flags)
    if check:
        planes.ensure_valid()
    return planes


def pack_bytes(raw, lut, bits=BITS_5, want_tokens=True, check=True):
    """
    Tokenise and pack on the device (pg_pack_bytes): `raw` = the fixed-width byte view of the sequence strings,
    (N, width) uint8 (host array or device tensor), `lut` = 256 table entries (letter -> token, else 0).
    Returns (Planes, tokens) with tokens = the (N, width) uint8 token matrix on the device, or None.
    """
    L = lib()
    dev = device()
    if not isinstance(raw, torch.Tensor):
        raw = torch.from_numpy(np.ascontiguousarray(np.asarray(raw, dtype=np.uint8)))
    if raw.dim() != 2 or raw.dtype != torch.uint8:
        raise TypeError("pack_bytes expects a 2-D uint8 byte matrix")
    raw = raw.to(dev).contiguous()
    n, width = raw.shape
    if n == 0 or width == 0:
        raise ValueError("empty byte matrix")
    if width > (MAX_L_5BIT if bits == BITS_5 else MAX_L):
        raise ValueError(f"L={width} exceeds the native limit for {bits} bit planes")
    lut_t = torch.as_tensor(np.asarray(lut, dtype=np.uint8).reshape(256)).to(dev)
    np_ = npad(n)
    buf = torch.empty(planes_bytes(n, width, bits), dtype=torch.uint8, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    tokens = torch.empty((n, width), dtype=torch.uint8, device=dev) if want_tokens else None
    _check(L.pg_pack_bytes(_ptr(raw), n, width, raw.stride(0), None, _ptr(lut_t), int(bits), _ptr(buf), np_, _ptr(tokens),
                           _ptr(flags), _stream()), "pg_pack_bytes")
    planes = Planes(buf, n, width, bits, flags)
    if check:
        planes.ensure_valid()
    return planes, tokens


_TORCH_OUT = {1: torch.uint8, 2: torch.float16, 4: torch.int32, 8: torch.int64}


def hamming_dense(xp, yp, out_bytes=8, out=None):
    """(M, N) distance matrix of every row of `yp` against every row of `xp` (hamming.py:34).
    With `out` given the distances are ADDED to it (segment-wise sums for long sequences)."""
    if xp.g != yp.g or xp.bits != yp.bits:
        raise ValueError("operands must be packed with the same width and bit planes")
    accumulate = out is not None
    if out is None:
        out = torch.empty((yp.n, xp.n), dtype=_TORCH_OUT[out_bytes], device=xp.buf.device)
    out_bytes = out.element_size()
    _check(lib().pg_hamming_dense(_ptr(xp.buf), xp.n, xp.npad, _ptr(yp.buf), yp.n, yp.npad, xp.g * 32, xp.bits,
                                  _ptr(out), out_bytes, out.stride(0), 1 if accumulate else 0, _stream()),
           "pg_hamming_dense")
    return out


def _bits2(rp, cp):
    if rp.bits != cp.bits or rp.g != cp.g:
        raise ValueError("row and column operands must be packed with the same width and bit planes")
    return rp.bits


def eps_graph(rp, cp, cmp, eps, row0=0, nrows=None, cap=256):
    """
    Epsilon-neighbourhood CSR of rows [row0, row0+nrows) of `rp` against all of `cp`.
    Returns device tensors (indptr int64 [nrows+1], indices int32 [nnz], weights uint8 [nnz]).
    ONE host sync (nnz and the number of rows that outgrew their slot) sits between the N^2 pass and
    the compaction: the output size is data dependent.  Rows with more matches than `cap` stay exact:
    a handful is recomputed inside the compaction kernel; more than that (dense graphs - most rows of
    a mutant library at eps >= 2) are listed on the device and the engine runs once more over just those
    rows, writing straight into the CSR (`pg_eps_fill_rows`).
    """
    L = lib()
    nrows = rp.n - row0 if nrows is None else int(nrows)
    dev = rp.buf.device
    bits = _bits2(rp, cp)
    cap = int(cap)
    symenv = os.environ.get("PG_EPS_SYM", "auto")            # 0 = never, 1 = whenever possible, auto = from 32k rows
    sym = (rp is cp and row0 == 0 and nrows == rp.n and rp.n < (1 << 27) and cap >= 2 and symenv != "0"
           and (symenv == "1" or rp.n >= 32768))          # whole square graph: every unordered pair once
    counts = torch.empty(nrows, dtype=torch.int32, device=dev)
    counts_lo = torch.empty(nrows, dtype=torch.int32, device=dev) if sym else None
    indptr = torch.empty(nrows + 1, dtype=torch.int64, device=dev)
    scratch = torch.empty(int(L.pg_scan_scratch_bytes(nrows)), dtype=torch.uint8, device=dev)
    slot_idx = torch.empty(nrows * cap, dtype=torch.int32, device=dev)
    slot_w = torch.empty(nrows * cap, dtype=torch.uint8, device=dev)
    if sym:
        args = (_ptr(rp.buf), rp.npad, rp.n, rp.g * 32, bits, cmp, float(eps), cap, _ptr(slot_idx), _ptr(slot_w),
                _ptr(counts), _ptr(counts_lo))
        _check(L.pg_eps_slots_sym(*args, _ptr(workspace(nrows, dev)), _stream()), "pg_eps_slots_sym")
        total = counts + counts_lo
        over = (total > cap) | (counts_lo > 512)              # more than PG_SORT_MAX entries from below: not rank-sorted in LDS
    else:
        args = (_ptr(rp.buf), rp.npad, row0, nrows, _ptr(cp.buf), cp.npad, cp.n, cp.g * 32, bits, cmp, float(eps),
                cap, _ptr(slot_idx), _ptr(slot_w), _ptr(counts))
        _check(L.pg_eps_slots(*args, _ptr(workspace(nrows, dev)), _stream()), "pg_eps_slots")
        total = counts
        over = total > cap
    _check(L.pg_exclusive_scan(_ptr(total), nrows, _ptr(indptr), _ptr(scratch), _stream()), "pg_exclusive_scan")
    nnz, n_over = (int(v) for v in torch.stack([indptr[-1], over.sum()]).cpu())     # the one sync
    indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
    weights = torch.empty(max(nnz, 1), dtype=torch.uint8, device=dev)[:nnz]
    if nnz:
        fill = n_over > int(os.environ.get("PG_FILL_MIN_ROWS", "8"))
        if sym:
            _check(L.pg_eps_compact_sym(*args, _ptr(indptr), _ptr(indices), _ptr(weights), 1 if fill else 0, _stream()),
                   "pg_eps_compact_sym")
        else:
            _check(L.pg_eps_compact(*args, _ptr(indptr), _ptr(indices), _ptr(weights), 1 if fill else 0, _stream()),
                   "pg_eps_compact")
        if fill:
            rows = compact_flags(over.to(torch.uint8), count=n_over)
            again = torch.empty(n_over, dtype=torch.int32, device=dev)
            _check(L.pg_eps_fill_rows(_ptr(rp.buf), rp.npad, row0, _ptr(rows), n_over, _ptr(cp.buf), cp.npad, cp.n,
                                      cp.g * 32, bits, cmp, float(eps), _ptr(indptr), _ptr(indices), _ptr(weights),
                                      _ptr(again), _ptr(workspace(n_over, dev)), _stream()), "pg_eps_fill_rows")
    return indptr, indices, weights


def eps_slots_only(rp, cp, cmp, eps, row0, nrows, cap, slot_idx, slot_w, counts):
    """Just the N^2 launch on preallocated buffers (bench.py times this)."""
    args = (_ptr(rp.buf), rp.npad, row0, nrows, _ptr(cp.buf), cp.npad, cp.n, cp.g * 32, _bits2(rp, cp), cmp,
            float(eps), int(cap))
    _check(lib().pg_eps_slots(*args, _ptr(slot_idx), _ptr(slot_w), _ptr(counts), _ptr(workspace(nrows, counts.device)), _stream()),
           "pg_eps_slots")


def knn_graph_rounds(rp, cp, k, row0=0, nrows=None):
    """k > 63: the canonical order is produced 63 + 64 + 64 ... ranks per all-pairs round, every
    round continuing after the previous round's last (distance, column) key."""
    L = lib()
    nrows = rp.n - row0 if nrows is None else int(nrows)
    dev = rp.buf.device
    bits = _bits2(rp, cp)
    idx = torch.empty((nrows, k), dtype=torch.int32, device=dev)
    dist = torch.empty((nrows, k), dtype=torch.uint8, device=dev)
    keys_a = torch.empty(nrows, dtype=torch.int32, device=dev)
    keys_b = torch.empty(nrows, dtype=torch.int32, device=dev)
    done, first = 0, True
    while done < k:
        kk = min(63 if first else 64, k - done)
        ri = torch.empty((nrows, kk), dtype=torch.int32, device=dev)
        rd = torch.empty((nrows, kk), dtype=torch.uint8, device=dev)
        _check(L.pg_knn_hamming_round(_ptr(rp.buf), rp.npad, row0, nrows, _ptr(cp.buf), cp.npad, cp.n, cp.g * 32, bits,
                                      kk, 1 if first else 0, _ptr(keys_a), _ptr(keys_b), _ptr(ri), _ptr(rd),
                                      _ptr(workspace(nrows, dev)), _stream()),
               "pg_knn_hamming_round")
        idx[:, done:done + kk] = ri
        dist[:, done:done + kk] = rd
        keys_a, keys_b = keys_b, keys_a
        done += kk
        first = False
    return idx, dist


def knn_graph(rp, cp, k, row0=0, nrows=None, out=None):
    """(nrows, k) int32 indices and uint8 distances, ranks 1..k of the canonical order."""
    if k > MAX_K and out is None:
        return knn_graph_rounds(rp, cp, k, row0=row0, nrows=nrows)
    nrows = rp.n - row0 if nrows is None else int(nrows)
    dev = rp.buf.device
    if out is None:
        idx = torch.empty((nrows, k), dtype=torch.int32, device=dev)
        dist = torch.empty((nrows, k), dtype=torch.uint8, device=dev)
    else:
        idx, dist = out
    _check(lib().pg_knn_hamming(_ptr(rp.buf), rp.npad, row0, nrows, _ptr(cp.buf), cp.npad, cp.n, cp.g * 32,
                                _bits2(rp, cp), int(k), _ptr(idx), _ptr(dist), _ptr(workspace(nrows, dev)), _stream()),
           "pg_knn_hamming")
    return idx, dist


def index_flags(planes, ref, want=None, pos_mode=0, pos_mask=None, not_mask=None, want_dist_out=True,
                want_hist=True, want_flags=True):
    """Fused 1xN pass of Prograph.indexing; returns (dist uint8[n] | None, hist int64[256] | None, flags | None).
    `pos_mask` / `not_mask` are iterables of positions (selected / must-not-differ)."""
    dev = planes.buf.device
    n = planes.n
    dist = torch.empty(n, dtype=torch.uint8, device=dev) if want_dist_out else None
    hist = torch.zeros(256, dtype=torch.int64, device=dev) if want_hist else None
    flags = torch.empty(n, dtype=torch.uint8, device=dev) if want_flags else None
    wt = None
    if want is not None:
        bits = np.zeros(8, dtype=np.uint32)
        for d in want:
            d = int(d)
            if 0 <= d < 256:
                bits[d >> 5] |= np.uint32(1 << (d & 31))
        wt = torch.from_numpy(bits.view(np.int32)).to(dev)
    pm = nm = None
    if pos_mode:
        pm = torch.from_numpy(position_bitmask(pos_mask, planes.g).view(np.int32)).to(dev)
        nm = torch.from_numpy(position_bitmask(not_mask, planes.g).view(np.int32)).to(dev)
    _check(lib().pg_index_flags(_ptr(planes.buf), n, planes.npad, planes.g * 32, planes.bits, int(ref), _ptr(wt),
                                int(pos_mode), _ptr(pm), _ptr(nm), _ptr(dist), _ptr(hist), _ptr(flags), _stream()),
           "pg_index_flags")
    return dist, hist, flags


def position_bitmask(positions, g):
    """uint32[g] with bit j of word w set for every position 32w+j in `positions`."""
    out = np.zeros(g, dtype=np.uint32)
    for p in positions:
        p = int(p)
        if not 0 <= p < 32 * g:
            raise IndexError(f"position {p} outside the packed width {32 * g}")
        out[p >> 5] |= np.uint32(1 << (p & 31))
    return out


def compact_flags(flags, count=None):
    """Ascending int64 indices of the non-zero entries of a uint8 device vector.  `count`: the number
    of non-zero entries when the caller already knows it (saves the host sync that reads it)."""
    L = lib()
    n = flags.numel()
    dev = flags.device
    out = torch.empty(n, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    scratch = torch.empty(int(L.pg_scan_scratch_bytes(n)), dtype=torch.uint8, device=dev)
    _check(L.pg_compact_flags(_ptr(flags), n, _ptr(out), _ptr(cnt), _ptr(scratch), _stream()), "pg_compact_flags")
    return out[: int(cnt.item()) if count is None else int(count)]


def levenshtein_knn(tokens, k, band=8, row0=0, nrows=None, cap=512, return_stats=False):
    """
    Banded (capped) Levenshtein kNN — build defined, BASELINE.json configs[4] (no reference
    counterpart).  `tokens`: (N, L<=128) uint8, tokens 1..31, zero right-padded.
    Returns (idx int32 (nrows,k), dist uint8 (nrows,k)): ranks 1..k of the (d, column) order with
    d = min(edit distance, band+1).
    Invalid input (a token above 31, an interior zero) raises ValueError - after the candidate pass: the validity
    word of the profile kernel is read together with the largest candidate count (the step's ONE host sync), so a
    bad matrix costs one filter sweep before it is rejected (the kernels mask symbols, nothing goes out of bounds).
    """
    L = lib()
    dev = device()
    if not isinstance(tokens, torch.Tensor):
        tokens = torch.from_numpy(np.ascontiguousarray(np.asarray(tokens)))
    if tokens.dtype != torch.uint8 or tokens.dim() != 2:
        raise TypeError("levenshtein_knn expects a 2-D uint8 token matrix")
    tokens = tokens.to(dev).contiguous()
    n, l = tokens.shape
    nrows = n - row0 if nrows is None else int(nrows)
    np_ = npad(n)
    prof = torch.empty(3 * np_ * 16, dtype=torch.uint8, device=dev)
    lens = torch.empty(n, dtype=torch.int32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    _check(L.pg_lev_profile(_ptr(tokens), n, l, tokens.stride(0), _ptr(prof), np_, _ptr(lens), _ptr(flags), _stream()),
           "pg_lev_profile")
    if l > 128:
        raise ValueError("levenshtein_knn: at most 128 tokens per sequence")
    planes = pack(tokens, bits=BITS_5, width=128, check=False)   # chunk p of a record = bit plane p (128 bits)
    counts = torch.empty(nrows, dtype=torch.int32, device=dev)
    symenv = os.environ.get("PG_EPS_SYM", "auto")            # the filter is symmetric like the eps graph
    sym = row0 == 0 and nrows == n and n < (1 << 24) and symenv != "0" and (symenv == "1" or n >= 32768)
    counts_lo = torch.empty(nrows, dtype=torch.int32, device=dev) if sym else None
    passes = 0
    while True:
        passes += 1
        slot_idx = torch.empty(nrows * cap, dtype=torch.int32, device=dev)
        slot_w = torch.empty(nrows * cap, dtype=torch.uint8, device=dev)
        slot_aux = torch.empty(nrows * cap, dtype=torch.int32, device=dev) if sym else None
        if sym:
            _check(L.pg_lev_candidates_sym(_ptr(prof), np_, n, int(band), int(cap), _ptr(slot_idx), _ptr(slot_w),
                                           _ptr(slot_aux), _ptr(counts), _ptr(counts_lo), _stream()), "pg_lev_candidates_sym")
            tot = counts + counts_lo
        else:
            _check(L.pg_lev_candidates(_ptr(prof), np_, n, row0, nrows, int(band), int(cap), _ptr(slot_idx),
                                       _ptr(slot_w), _ptr(counts), _stream()), "pg_lev_candidates")
            tot = counts
        # the ONE host sync of a step: the token check of the profile pass and the largest candidate count
        bad, mx = (int(v) for v in torch.stack([(flags[0] | planes.flags[0]).to(torch.int64), tot.max().to(torch.int64)]).cpu())
        if bad:
            raise ValueError("levenshtein_knn: tokens must be 1..31 with zeros only as right padding")
        if mx <= cap:
            break
        cap = ((mx + 63) // 64) * 64          # some row has more candidates than slots: redo with room
    idx = torch.empty((nrows, k), dtype=torch.int32, device=dev)
    dist = torch.empty((nrows, k), dtype=torch.uint8, device=dev)
    _check(L.pg_lev_knn(_ptr(tokens), n, l, tokens.stride(0), _ptr(planes.buf), planes.npad, _ptr(lens), row0, nrows,
                        int(band), int(k), int(cap),
                        _ptr(slot_idx), _ptr(slot_w), _ptr(slot_aux), _ptr(counts), _ptr(counts_lo), _ptr(idx), _ptr(dist),
                        _stream()), "pg_lev_knn")
    if return_stats:
        ncand = int(counts.to(torch.int64).sum().item()) + (int(counts_lo.to(torch.int64).sum().item()) if sym else 0)
        return idx, dist, {"candidates": ncand, "cap": cap, "filter_passes": passes, "symmetric": bool(sym)}
    return idx, dist


def csr_row_stats(indptr, indices, weights, f=None, want=("deg",), row0=0, ncols=None):
    """Per-row reductions over a device CSR (see pg_csr_row_stats).  `weights`: uint8 or float32
    device tensor or None (boolean).  `want` from deg / sum_f / sum_wf / self_w (per row) and col_sum
    (per column, needs `ncols`).  Returns a dict of float64 device tensors."""
    nrows = indptr.numel() - 1
    dev = indptr.device
    out = {k: torch.empty(nrows, dtype=torch.float64, device=dev) for k in want if k != "col_sum"}
    if "col_sum" in want:
        out["col_sum"] = torch.zeros(int(ncols), dtype=torch.float64, device=dev)
    w8 = weights if (weights is not None and weights.dtype == torch.uint8) else None
    wf = weights if (weights is not None and weights.dtype == torch.float32) else None
    if weights is not None and w8 is None and wf is None:
        raise TypeError("weights must be uint8 or float32")
    fd = None if f is None else f.to(device=dev, dtype=torch.float64).contiguous()
    _check(lib().pg_csr_row_stats(_ptr(indptr), _ptr(indices), _ptr(w8), _ptr(wf), nrows, int(row0), _ptr(fd),
                                  _ptr(out.get("deg")), _ptr(out.get("sum_f")), _ptr(out.get("sum_wf")),
                                  _ptr(out.get("self_w")), _ptr(out.get("col_sum")), _stream()), "pg_csr_row_stats")
    return out


class PackedF16:
    """Device-resident fp16 vectors in chunk-major order (pg_pack_f16)."""
    __slots__ = ("buf", "n", "d", "npad")

    def __init__(self, buf, n, d):
        self.buf, self.n, self.d, self.npad = buf, int(n), int(d), npad(n)


def pack_f16(x):
    """(N, D) fp16 device tensor -> PackedF16."""
    if x.dtype != torch.float16 or x.dim() != 2 or not x.is_cuda or x.shape[0] == 0 or x.shape[1] == 0:
        raise TypeError("pack_f16 expects a non-empty 2-D fp16 device tensor")
    x = x.contiguous()
    n, d = x.shape
    np_ = npad(n)
    buf = torch.empty(((d + 7) // 8) * np_ * 16, dtype=torch.uint8, device=x.device)
    _check(lib().pg_pack_f16(_ptr(x), n, d, x.stride(0), None, _ptr(buf), np_, _stream()), "pg_pack_f16")
    return PackedF16(buf, n, d)


def minkowski_dense(xp, yp, similarity=False):
    """(M, N) fp16 block: Minkowski p=2 distance (or 1/(1+d)) of every Y vector against every X vector,
    rounded step by step like the reference's fp16 tensor expression (minkowski.py:36-40)."""
    if xp.d != yp.d:
        raise ValueError("operands must have the same dimension")
    out = torch.empty((yp.n, xp.n), dtype=torch.float16, device=xp.buf.device)
    _check(lib().pg_minkowski_dense(_ptr(xp.buf), xp.n, xp.npad, _ptr(yp.buf), yp.n, yp.npad, xp.d, 1 if similarity else 0,
                                    _ptr(out), out.stride(0), _stream()), "pg_minkowski_dense")
    return out


def f16_knn(block, k, first=1, descending=False):
    """Ranks first..first+k-1 of every row of an fp16 block in (value, column) order -> (idx int32, w fp16)."""
    m, n = block.shape
    idx = torch.empty((m, k), dtype=torch.int32, device=block.device)
    w = torch.empty((m, k), dtype=torch.float16, device=block.device)
    _check(lib().pg_f16_knn(_ptr(block), m, n, block.stride(0), int(k), int(first), 1 if descending else 0, _ptr(idx), _ptr(w),
                            _stream()), "pg_f16_knn")
    return idx, w


def f16_eps(block, cmp, eps, similarity=False):
    """CSR of the entries of an fp16 block that satisfy comp(d, eps) & (d > 0)  [comp(eps, s) & (s < 1)].
    `eps` is rounded to fp16 first, as torch does when an fp16 tensor meets a Python number."""
    L = lib()
    m, n = block.shape
    dev = block.device
    e16 = float(np.float16(eps))
    counts = torch.empty(m, dtype=torch.int32, device=dev)
    _check(L.pg_f16_eps_count(_ptr(block), m, n, block.stride(0), int(cmp), e16, 1 if similarity else 0, _ptr(counts), _stream()),
           "pg_f16_eps_count")
    indptr = torch.empty(m + 1, dtype=torch.int64, device=dev)
    scratch = torch.empty(int(L.pg_scan_scratch_bytes(m)), dtype=torch.uint8, device=dev)
    _check(L.pg_exclusive_scan(_ptr(counts), m, _ptr(indptr), _ptr(scratch), _stream()), "pg_exclusive_scan")
    nnz = int(indptr[-1].item())
    indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
    weights = torch.empty(max(nnz, 1), dtype=torch.float16, device=dev)[:nnz]
    if nnz:
        _check(L.pg_f16_eps_fill(_ptr(block), m, n, block.stride(0), int(cmp), e16, 1 if similarity else 0, _ptr(indptr),
                                 _ptr(indices), _ptr(weights), _stream()), "pg_f16_eps_fill")
    return indptr, indices, weights


COMM_ID_BYTES = 128


def comm_available():
    """Can this process bind RCCL (local check, not a collective)?"""
    try:
        return bool(lib().pg_comm_available())
    except NativeUnavailable:
        return False


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it, the host carries it to the other ranks)."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(lib().pg_comm_unique_id(buf), "pg_comm_unique_id")
    return buf.raw


def comm_init(nranks, rank, id_bytes):
    """RCCL communicator of this rank on the current HIP device; returns an opaque handle."""
    device()
    h = _vp(0)
    _check(lib().pg_comm_init(ctypes.byref(h), int(nranks), int(rank), ctypes.c_char_p(bytes(id_bytes))), "pg_comm_init")
    return h


def comm_destroy(comm):
    _check(lib().pg_comm_destroy(comm), "pg_comm_destroy")


def allgather_tokens(comm, shard, nranks):
    """(rows_per_rank, L) uint8 device shard of every rank -> (nranks*rows_per_rank, L) on every rank."""
    if shard.dtype != torch.uint8 or shard.dim() != 2 or not shard.is_cuda:
        raise TypeError("allgather_tokens expects a 2-D uint8 device tensor")
    shard = shard.contiguous()
    full = torch.empty((shard.shape[0] * int(nranks), shard.shape[1]), dtype=torch.uint8, device=shard.device)
    _check(lib().pg_allgather_tokens(comm, _ptr(shard), shard.shape[0], shard.shape[1], _ptr(full), _stream()), "pg_allgather_tokens")
    return full


def device_info():
    L = lib()
    device()
    cus, wave = _i32(0), _i32(0)
    arch = ctypes.create_string_buffer(64)
    _check(L.pg_device_info(ctypes.byref(cus), ctypes.byref(wave), arch, 64), "pg_device_info")
    return {"cus": cus.value, "wave": wave.value, "arch": arch.value.decode()}
