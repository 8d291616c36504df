"""
ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's graph-construction hot path (acmater/prograph,
citations are into /root/reference).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module, and only as the checker or
as the reported CPU baseline.  Nothing under `prograph_amd/` imports it.

Parity status: PINNED.  The functions below are checked bit for bit against golden
vectors produced by importing the real reference in the build container
(`oracle/gen_golden.py` -> `tests/golden/*.npz`; `tests/test_oracle.py`), and against
the known answers the reference's own `tests/tests.py` holds.
The one function with no reference counterpart is `levenshtein_banded` (BASELINE
config 5): PARITY UNPINNED, see its docstring.

The arithmetic deliberately uses the same torch CPU operators the reference calls on
its device tensor (`!=`, `sum`, `where`, `sort`), batch by batch, so that timing this
file is timing the reference's algorithm ("port" in bench.py's `cpu_baseline`).
"""
import operator
from functools import reduce

import numpy as np
import torch

AMINO_ACIDS = "ACDEFGHIKLMNPQRSTVWY"


# --------------------------------------------------------------------------------------
# a1  Prograph.tokenize                                  prograph/prograph.py:454-474
# --------------------------------------------------------------------------------------
def tokenize(sequences, amino_acids=AMINO_ACIDS):
    """bytes view 'S1' of the string array, one np.where pass per alphabet letter;
    token table `prograph/prograph.py:127` (letter j -> j+1, pad / unknown -> 0)."""
    seqs = np.array(sequences, dtype="bytes").reshape(-1, 1).view("S1")
    out = np.zeros(seqs.shape, dtype=int)
    for tok, ch in enumerate(amino_acids, start=1):
        out[np.where(seqs == ch.encode("utf-8"))] = tok
    return out


# --------------------------------------------------------------------------------------
# a2  clean_input                                        prograph/distance/utils.py:7-39
# --------------------------------------------------------------------------------------
def clean_input(X, Y):
    if X.shape[0] == 0 or Y.shape[0] == 0:                       # utils.py:29-30
        raise ValueError("You cannot pass an empty tensor.")
    X = torch.atleast_2d(torch.as_tensor(X))                     # utils.py:31
    Y = torch.atleast_2d(torch.as_tensor(Y))
    if X.shape[1] != Y.shape[1]:                                 # utils.py:32-38
        if Y.shape[1] > X.shape[1]:
            X = torch.nn.functional.pad(X, (0, Y.shape[1] - X.shape[1]))
        else:
            Y = torch.nn.functional.pad(Y, (0, X.shape[1] - Y.shape[1]))
    return X, Y


# --------------------------------------------------------------------------------------
# a3  hamming                                            prograph/distance/hamming.py:8-39
# --------------------------------------------------------------------------------------
def hamming(X, Y, similarity=False):
    """(M, N) int64: row m = Y[m] against every row of X (hamming.py:34; the docstring's
    N x M is wrong, pinned by tests/tests.py:175-186)."""
    X, Y = clean_input(X, Y)
    d = torch.sum(X != Y[:, None, :], axis=2)                    # hamming.py:34
    if similarity:
        d = 1 / (1 + d)                                          # hamming.py:38
    return d


# --------------------------------------------------------------------------------------
# a4  get_every_n                                        prograph/prograph.py:617-624
# --------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------
# f2  minkowski                                          prograph/distance/minkowski.py:8-41
# --------------------------------------------------------------------------------------
def minkowski(X, Y, p=2, similarity=False):
    """The reference's tensor expression verbatim in meaning: with fp16 operands (build_graph stages the
    embedding as fp16, prograph/prograph.py:726) every elementwise step rounds to fp16."""
    X, Y = clean_input(X, Y)
    distances = torch.pow(torch.sum(torch.pow(X - Y[:, None, :], exponent=p), axis=2), exponent=1 / p)   # :36
    if similarity:
        distances = 1 / (1 + distances)                                                             # :40
    return distances


def get_every_n(a, n=2):
    for i in range((a.shape[0] // n) + (a.shape[0] % n > 0)):
        yield a[n * i:n * (i + 1)]


# --------------------------------------------------------------------------------------
# a6  prod_neighbours                                    prograph/prograph.py:626-654
# --------------------------------------------------------------------------------------
def prod_neighbours(index, out, batch_size, weights=None):
    results = {}
    row, col = out
    row = row + (index * batch_size)                             # :648
    if weights is None:
        weights = np.ones(col.shape)
    for idx in np.unique(row):                                   # :652-653
        sel = np.where(row == idx)
        results[idx] = (col[sel], weights[sel])
    return results


def _validate(eps, k):
    if operator.xor(bool(eps), bool(k)) is False:                # :714-715
        raise ValueError("Epsilon or K must be provided, but both cannot be.")
    if k is not None and not isinstance(k, int):                 # :716-718
        raise TypeError("K must be provided as an integer.")


# --------------------------------------------------------------------------------------
# a5 / a7  build_graph                                   prograph/prograph.py:656-765
# --------------------------------------------------------------------------------------
def build_graph(tokens, idxs=None, batch_size=8, eps=None, k=None, similarity=False,
                distance=hamming, comp=operator.le, stable=True, dtype=torch.float16, row_limit=None):
    """
    Restates both branches of `build_graph`.  `tokens` is the (N, L) representation the
    reference fetches with `self(representation)`; it is cast to fp16 exactly as
    `prograph/prograph.py:726` does (lossless for tokens 0..2048).

    `stable=True` is the build's canonical kNN tie rule (SURVEY.md §7 hard part 2): the
    reference's `torch.sort` at :758-760 is unstable, so with integer distances its
    *indices* are implementation defined; weights are not.

    `row_limit=R` walks only the batches of the first R rows (against all columns): the bounded
    sample bench.py times as the CPU baseline; the per-batch work is unchanged.
    """
    _validate(eps, k)
    if similarity and eps:
        eps = 1 / (1 + eps)                                      # :720-721
    X = torch.as_tensor(np.asarray(tokens), dtype=dtype)         # :726 (device = CPU here)
    if idxs is not None:
        X = X[idxs, :]
    weights, edge_idxs = [], []
    R = X if row_limit is None else X[:row_limit]
    if eps:
        for batch in list(get_every_n(R, n=batch_size)):         # :731
            d = distance(X, batch, similarity=similarity)        # :732
            if similarity:
                loc = torch.where(comp(eps, d) & (d < 1))        # :734
            else:
                loc = torch.where(comp(d, eps) & (d > 0))        # :736
            weights.append(d[loc].numpy())                       # :738
            edge_idxs.append([x.numpy() for x in loc])           # :739
        final = [prod_neighbours(i, r, batch_size, weights=weights[i])
                 for i, r in enumerate(edge_idxs)]               # :743-745
        nd = {kk: v for d_ in final for kk, v in d_.items()}     # :748
        return [nd.get(i, (np.array([], dtype=int), np.array([], dtype=int)))
                for i in range(len(R))]                          # :751-753
    for batch in list(get_every_n(R, n=batch_size)):             # :756
        d = distance(X, batch, similarity=similarity)
        s = torch.sort(d, dim=1, descending=bool(similarity), stable=stable)   # :758-760
        weights.append([x.numpy() for x in s[0][:, 1:k + 1]])    # :761
        edge_idxs.append([x.numpy() for x in s[1][:, 1:k + 1]])  # :762
    flat = lambda l: [it for sub in l for it in sub]             # utils/general.py:55-59
    return list(zip(flat(edge_idxs), flat(weights)))             # :764


def neighbours_to_csr(neigh):
    """list of N (idx, w) tuples -> (indptr int64, indices int64, weights)."""
    counts = np.array([len(a[0]) for a in neigh], dtype=np.int64)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    if indptr[-1] == 0:
        return indptr, np.zeros(0, np.int64), np.zeros(0, np.int64)
    return (indptr,
            np.concatenate([np.asarray(a[0], dtype=np.int64) for a in neigh]),
            np.concatenate([np.asarray(a[1]) for a in neigh]))


def neighbours_to_knn(neigh):
    """list of N (idx(k,), w(k,)) -> (N,k) idx int64, (N,k) weights."""
    return (np.stack([np.asarray(a[0], dtype=np.int64) for a in neigh]),
            np.stack([np.asarray(a[1]) for a in neigh]))


# --------------------------------------------------------------------------------------
# a8  indexing (+ boolean_mutant_array, calc_neighbours, __str__ numbers)
#                                                        prograph/prograph.py:254-343
# --------------------------------------------------------------------------------------
def boolean_mutant_array(tokenized, ref_idx):                    # :488-492
    return tokenized != tokenized[ref_idx]


def indexing(tokenized, ref_idx, seq_len, distances=None, positions=None, percentage=None,
             Bool="or", complement=False, rng=None):
    """`seq_len` = len of the *reference string* (`len(self[reference_seq]["Sequence"])`,
    :316), which is what bounds `not_positions`."""
    idxs = []
    assert Bool == "or" or Bool == "and", "Not a valid boolean value."          # :293
    tokenized = np.asarray(tokenized)
    d_data = hamming(tokenized, tokenized[ref_idx].reshape(1, -1))               # :298
    if distances is not None:
        if type(distances) == int:
            distances = [distances]
        assert type(distances) == list, "Distances must be provided as integer or list"
        for d in distances:
            assert d in np.unique(d_data), f"{d} is not a valid distance"        # :305
        idxs.append(reduce(np.union1d, [np.where(d_data == d)[1] for d in distances]))
    if positions is not None:
        not_positions = [x for x in range(seq_len) if x not in positions]        # :316
        mut = boolean_mutant_array(tokenized, ref_idx)
        op = np.logical_or if Bool == "or" else np.logical_and
        working = reduce(op, [mut[:, p] for p in positions])                     # :318-321
        for p in not_positions:                                                  # :322-324
            temp = np.logical_xor(working, mut[:, p])
            working = np.logical_and(temp, np.logical_not(mut[:, p]))
        idxs.append(np.where(working)[0])
    if len(idxs) > 0:
        idxs = reduce(np.intersect1d, idxs)                                      # :328
    else:
        idxs = np.array(range(len(tokenized)))
    if percentage is not None:
        assert 0 <= percentage <= 1, "Percentage must be between 0 and 1"
        rng = np.random if rng is None else rng
        sel = np.zeros(len(idxs), dtype=bool)
        sel[rng.choice(np.arange(len(idxs)), size=int(len(idxs) * percentage), replace=False)] = 1
        return idxs[sel]
    assert len(idxs) != 0, "No possible valid indices have been provided."      # :338
    if complement:
        return idxs, np.setdiff1d(np.arange(len(tokenized)), idxs)               # :341
    return idxs


def calc_neighbours(tokenized, ref_idx, eps=1, comp=operator.eq):                # :526-544
    tokenized = np.asarray(tokenized)
    return np.where(comp(hamming(tokenized, tokenized[ref_idx].reshape(1, -1)), eps))[1]


def summary_numbers(tokenized, seed_idx):
    """The two distance-derived numbers `__str__` prints (:147-154)."""
    d = hamming(np.asarray(tokenized), np.asarray(tokenized)[seed_idx].reshape(1, -1))
    return int(torch.max(d)), int(len(np.unique(d)))


# --------------------------------------------------------------------------------------
# f1  CSR consumers                                      prograph/prograph.py:797-872
# --------------------------------------------------------------------------------------
def degree(neigh, boolean_weights=False):                                        # :797-822
    deg = np.zeros((len(neigh),), dtype=np.float32)
    for i, e in enumerate(neigh):
        deg[i] = len(e[0]) if boolean_weights else np.sum(np.asarray(e[1]).astype(np.float32))
    return deg


def neighbour_coords(neigh, boolean_weights=False):                              # :824-857
    nb, w = zip(*neigh)
    I = np.concatenate([np.zeros(len(J), dtype=int) + i for i, J in enumerate(nb)])
    J = np.concatenate(nb)
    if boolean_weights:
        return I, J, np.ones(I.shape)
    return I, J, np.concatenate(w).astype(np.float32)


# --------------------------------------------------------------------------------------
# a9  banded Levenshtein — NOT IN THE REFERENCE.  PARITY UNPINNED.
# --------------------------------------------------------------------------------------
def levenshtein_banded(a, la, b, lb, band):
    """
    Build-defined (BASELINE.json configs[4]).  Plain Wagner–Fischer
    restricted to the diagonal band |i - j| <= band; the result is capped:
    returns min(true_distance_within_band, band + 1), and band + 1 whenever
    |la - lb| > band.  There is no reference function, test or fixture for this, so
    parity is unpinned: the GPU kernel is checked against this definition only
    (plus an unbanded Wagner–Fischer cross-check for pairs whose true distance <= band,
    where the banded value is provably exact).
    """
    la, lb = int(la), int(lb)
    cap = band + 1
    if abs(la - lb) > band:
        return cap
    INF = 1 << 20
    prev = [j if j <= band else INF for j in range(lb + 1)]
    for i in range(1, la + 1):
        cur = [INF] * (lb + 1)
        lo, hi = max(0, i - band), min(lb, i + band)
        if lo == 0:
            cur[0] = i
        for j in range(max(1, lo), hi + 1):
            cost = 0 if a[i - 1] == b[j - 1] else 1
            v = prev[j - 1] + cost
            if prev[j] + 1 < v:
                v = prev[j] + 1
            if cur[j - 1] + 1 < v:
                v = cur[j - 1] + 1
            cur[j] = v
        prev = cur
    return min(prev[lb], cap)


def levenshtein_full(a, la, b, lb):
    la, lb = int(la), int(lb)
    prev = list(range(lb + 1))
    for i in range(1, la + 1):
        cur = [i] + [0] * lb
        for j in range(1, lb + 1):
            cur[j] = min(prev[j - 1] + (a[i - 1] != b[j - 1]), prev[j] + 1, cur[j - 1] + 1)
        prev = cur
    return prev[lb]


def knn_from_distance_matrix(D, k):
    """Canonical (distance, index) order, drop rank 0, ranks 1..k (same rule as kNN Hamming)."""
    D = torch.as_tensor(np.asarray(D))
    s = torch.sort(D, dim=1, stable=True)
    return s[1][:, 1:k + 1].numpy(), s[0][:, 1:k + 1].numpy()
