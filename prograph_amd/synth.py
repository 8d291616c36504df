"""
Deterministic synthetic token matrices for tests and bench.

The generator is counter based (splitmix64 finaliser), so it is independent of
the numpy version and can be restated bit for bit in C (oracle/oracle.c does).

    mix64(z):  z = (z ^ z>>30) * 0xBF58476D1CE4E5B9
               z = (z ^ z>>27) * 0x94D049BB133111EB
               z =  z ^ z>>31
    h(seed, stream, i) = mix64(seed + 0x9E3779B97F4A7C15 * (4*i + stream + 1))

Rows are clustered (the reference's graphs are built on mutant libraries around
a seed sequence, `prograph/prograph.py:119-136`): `n_centres = max(1, N // 256)`
random centres over tokens 1..20, row i copies centre `i % n_centres` and takes
1..3 substitutions.  Uniform random rows would make every eps<=2 graph empty
(mean distance 0.95*L).  Duplicates are kept on purpose: they exercise the
reference's `d > 0` exclusion (`prograph/prograph.py:736`) and the kNN rank-0
rule (`:761-762`).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)

DEFAULT_SEED = 20260104


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def h(seed, stream, i):
    i = np.asarray(i, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed) + _GOLD * (np.uint64(4) * i + np.uint64(stream + 1)))


def clustered_tokens(N, L, seed=DEFAULT_SEED, members=256, row0=0, nrows=None):
    """(N, L) uint8 tokens in 1..20, clustered as described in the module docstring.
    `row0`/`nrows` return only rows [row0, row0+nrows) of the same N-row matrix (every row is a
    pure function of its index, so ranks can generate their own shard)."""
    N, L = int(N), int(L)
    nrows = N - row0 if nrows is None else int(nrows)
    n_centres = max(1, N // members)
    cj = np.arange(n_centres * L, dtype=np.uint64)
    centres = (1 + h(seed, 0, cj) % np.uint64(20)).astype(np.uint8).reshape(n_centres, L)
    i = np.arange(row0, row0 + nrows, dtype=np.uint64)
    tok = centres[(i % np.uint64(n_centres)).astype(np.int64)].copy()
    m = (1 + h(seed, 1, i) % np.uint64(3)).astype(np.int64)
    rows = np.arange(nrows)
    for t in range(3):
        act = m > t
        pos = (h(seed, 2, np.uint64(8) * i + np.uint64(t)) % np.uint64(L)).astype(np.int64)
        step = (h(seed, 3, np.uint64(8) * i + np.uint64(t)) % np.uint64(19)).astype(np.int64)
        cur = tok[rows, pos].astype(np.int64)
        new = 1 + ((cur - 1 + 1 + step) % 20)
        tok[rows[act], pos[act]] = new[act].astype(np.uint8)
    return tok


def clustered_varlen_tokens(N, Lmax=128, Lmin=96, seed=DEFAULT_SEED, members=256, row0=0, nrows=None):
    """
    Variable-length clustered rows for the banded Levenshtein configuration
    (BASELINE.json configs[4]; build defined): cluster
    centres of length Lmin..Lmax, each member takes 1..3 edits drawn from
    {substitute, insert, delete} applied in order; rows are right padded with 0 to Lmax.
    Edit t of row i: kind = h(5, 8i+t) % 3, position = h(2, 8i+t) % current_length,
    token = 1 + h(3, 8i+t) % 20; an insert into a full row / a delete from a 1-token row is a no-op.
    Returns (tokens (nrows, Lmax) uint8, lengths (nrows,) int32); `row0`/`nrows` select a shard.
    """
    N = int(N)
    nrows = N - row0 if nrows is None else int(nrows)
    n_centres = max(1, N // members)
    cj = np.arange(n_centres * Lmax, dtype=np.uint64)
    centres = (1 + h(seed, 0, cj) % np.uint64(20)).astype(np.uint8).reshape(n_centres, Lmax)
    clen = (Lmin + h(seed, 4, np.arange(n_centres, dtype=np.uint64)) % np.uint64(Lmax - Lmin + 1)).astype(np.int64)
    i = np.arange(row0, row0 + nrows, dtype=np.uint64)
    c = (i % np.uint64(n_centres)).astype(np.int64)
    lens = clen[c].copy()
    j = np.arange(Lmax)[None, :]
    arr = np.where(j < lens[:, None], centres[c], 0).astype(np.uint8)
    m = (1 + h(seed, 1, i) % np.uint64(3)).astype(np.int64)
    rows = np.arange(nrows)
    for t in range(3):
        act = m > t
        e = np.uint64(8) * i + np.uint64(t)
        kind = (h(seed, 5, e) % np.uint64(3)).astype(np.int64)
        pos = (h(seed, 2, e) % lens.astype(np.uint64)).astype(np.int64)
        tokv = (1 + h(seed, 3, e) % np.uint64(20)).astype(np.uint8)
        sub = act & (kind == 0)
        ins = act & (kind == 1) & (lens < Lmax)
        dele = act & (kind == 2) & (lens > 1)
        arr[rows[sub], pos[sub]] = tokv[sub]
        src = np.broadcast_to(j, arr.shape).copy()
        src[ins] -= (j > pos[ins, None])
        src[dele] += (j >= pos[dele, None])
        padded = np.concatenate([arr, np.zeros((nrows, 1), np.uint8)], axis=1)
        arr = np.take_along_axis(padded, np.minimum(src, Lmax), axis=1)
        arr[rows[ins], pos[ins]] = tokv[ins]
        lens = lens + ins.astype(np.int64) - dele.astype(np.int64)
        arr[j >= lens[:, None]] = 0
    return arr, lens.astype(np.int32)


AMINO = "ACDEFGHIKLMNPQRSTVWY"


def tokens_to_strings(tok):
    """Inverse of `Prograph.tokenize` for tokens 1..20 (0 = right padding is dropped)."""
    lut = np.array([""] + list(AMINO))
    return ["".join(lut[r[r > 0]]) for r in np.asarray(tok)]
