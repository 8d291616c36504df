"""
Result containers of the graph-construction path.

The reference hands back a Python list of N `(indices, weights)` tuples per graph
(prograph/prograph.py:751-753, :764) and stores it in a DataFrame column.  The HIP path
produces CSR / (N,k) arrays on the device; these classes keep that form (device resident,
no per-row Python objects) and materialise the reference's tuples only on request, as
zero-copy views into two host arrays.
"""
import numpy as np
import torch

from . import _native


class CSRGraph:
    """epsilon-neighbourhood graph: indptr int64 [n+1], indices int32 [nnz], weights uint8|int16|float32 [nnz]."""

    def __init__(self, indptr, indices, weights, ncols, similarity=False, row0=0):
        self.indptr, self.indices, self.weights = indptr, indices, weights
        self.ncols, self.similarity, self.row0 = int(ncols), bool(similarity), int(row0)

    @property
    def nrows(self):
        return int(self.indptr.numel() - 1)

    @property
    def nnz(self):
        return int(self.indices.numel())

    def host(self):
        """(indptr, indices int64, weights) as numpy, weights in the reference's dtype:
        int64 Hamming distances, or float32 similarities 1/(1+d) (hamming.py:38)."""
        indptr = self.indptr.cpu().numpy()
        idx = self.indices.to(torch.int64).cpu().numpy()
        if self.similarity:
            w = (1 / (1 + self.weights.to(torch.int64))).cpu().numpy()
        else:
            w = self.weights.to(torch.int64).cpu().numpy()
        return indptr, idx, w

    def to_tuples(self):
        """list of N (np.int64 indices ascending, weights) — the reference's `Neighbours` format.
        Rows without neighbours get `(array([], int), array([], int))` (prograph.py:753); all such rows
        share ONE pair of zero-length arrays (nothing can be written into them), the others are
        views into the two host arrays."""
        indptr, idx, w = self.host()
        nothing = (np.array([], dtype=int), np.array([], dtype=int))
        a, b = indptr[:-1].tolist(), indptr[1:].tolist()
        return [(idx[i:j], w[i:j]) if j > i else nothing for i, j in zip(a, b)]

    def _w(self, boolean_weights):
        if boolean_weights:
            return None
        if self.similarity:
            return (1 / (1 + self.weights.to(torch.int64))).to(torch.float32)
        if self.weights.dtype not in (torch.uint8, torch.float32):
            return self.weights.to(torch.float32)          # int16 distances of sequences beyond one record: exact
        return self.weights

    def row_stats(self, f=None, boolean_weights=False, want=("deg",)):
        """Device reductions (pg_csr_row_stats): per row deg = sum w, sum_f = sum f[col], sum_wf = sum w f[col],
        self_w = weight of the row's own node in its list; per column col_sum = sum of the column's weights."""
        return _native.csr_row_stats(self.indptr, self.indices, self._w(boolean_weights), f=f, want=want,
                                     row0=self.row0, ncols=self.ncols)

    def degree(self, boolean_weights=False):
        """Out-degree per row as float32 (prograph.py:797-822) without touching Python tuples."""
        if boolean_weights:
            return (self.indptr[1:] - self.indptr[:-1]).to(torch.float32).cpu().numpy()
        return self.row_stats(want=("deg",))["deg"].to(torch.float32).cpu().numpy()

    def _degree_vector(self, st, mode):
        """The Laplacian's diagonal D as the reference builds it (prograph.py:887-896): out-degree = row
        sums kept in float32 (`degree()`), in-degree = column sums of the float32 adjacency."""
        if mode == "outdegree":
            return st["deg"].to(torch.float32).to(torch.float64)
        if mode == "indegree":
            return st["col_sum"][self.row0:self.row0 + self.nrows].to(torch.float32).to(torch.float64)
        raise ValueError("Not a valid degree mode.")

    def laplacian_diagonal(self, boolean_weights=False, mode="outdegree"):
        want = ("deg",) if mode == "outdegree" else ("col_sum",)
        if mode not in ("outdegree", "indegree"):
            raise ValueError("Not a valid degree mode.")
        return self._degree_vector(self.row_stats(boolean_weights=boolean_weights, want=want), mode).cpu().numpy()

    def dirichlet(self, f, boolean_weights=False, mode="outdegree"):
        """f^T L f with L = -A, diagonal OVERWRITTEN by D (`L.setdiag(D)`, prograph.py:887-896): a row's own
        node in its neighbour list (kNN with duplicated sequences) does not enter the off-diagonal sum.
        f = per-node values of the ROWS' nodes (square graph: nrows == ncols)."""
        if mode not in ("outdegree", "indegree"):
            raise ValueError("Not a valid degree mode.")
        fd = torch.as_tensor(np.asarray(f, dtype=np.float64).reshape(-1), device=self.indptr.device)
        want = ("deg", "sum_wf", "self_w") + (("col_sum",) if mode == "indegree" else ())
        st = self.row_stats(f=fd, boolean_weights=boolean_weights, want=want)
        fr = fd[self.row0:self.row0 + self.nrows]
        D = self._degree_vector(st, mode)
        off = st["sum_wf"] - st["self_w"] * fr                # sum over j != r of A_rj f_j
        return float((fr * (D * fr - off)).sum().item())

    def local_variance(self, f):
        """mean_j (f_r - f_j) over the row's neighbours (prograph.py:924-946); NaN for empty rows."""
        fd = torch.as_tensor(np.asarray(f, dtype=np.float64).reshape(-1), device=self.indptr.device)
        st = self.row_stats(f=fd, boolean_weights=True, want=("sum_f",))
        cnt = (self.indptr[1:] - self.indptr[:-1]).to(torch.float64)
        fr = fd[self.row0:self.row0 + self.nrows]
        out = torch.where(cnt > 0, fr - st["sum_f"] / cnt.clamp(min=1), torch.full_like(fr, float("nan")))
        return out.cpu().numpy()

    def coords(self, boolean_weights=False):
        """(I, J, V) of prograph.py:824-857 straight from CSR."""
        indptr, idx, w = self.host()
        I = np.repeat(np.arange(self.nrows, dtype=int) + self.row0, np.diff(indptr))
        if boolean_weights:
            return I, idx, np.ones(I.shape)
        return I, idx, w.astype(np.float32)


class KNNGraph:
    """k nearest neighbours: idx int32 (n,k), dist uint8 (n,k); canonical (distance, index) order."""

    def __init__(self, idx, dist, ncols, similarity=False, row0=0):
        self.idx, self.dist = idx, dist
        self.ncols, self.similarity, self.row0 = int(ncols), bool(similarity), int(row0)

    @property
    def nrows(self):
        return int(self.idx.shape[0])

    def host(self):
        kk = min(self.idx.shape[1], max(self.ncols - 1, 0))    # ranks beyond N-1 do not exist ([:,1:k+1])
        idx = self.idx[:, :kk].to(torch.int64).cpu().numpy()
        d = self.dist[:, :kk].to(torch.int64)
        w = (1 / (1 + d)).cpu().numpy() if self.similarity else d.cpu().numpy()
        return idx, w

    def to_tuples(self):
        idx, w = self.host()
        return list(zip(list(idx), list(w)))

    def as_csr(self):
        """The same graph as a CSRGraph view (k entries per row; ranks beyond N-1 do not exist)."""
        kk = min(self.idx.shape[1], max(self.ncols - 1, 0))
        n = self.nrows
        indptr = torch.arange(0, n * kk + 1, kk, dtype=torch.int64, device=self.idx.device) if kk else \
            torch.zeros(n + 1, dtype=torch.int64, device=self.idx.device)
        return CSRGraph(indptr, self.idx[:, :kk].reshape(-1).contiguous(), self.dist[:, :kk].reshape(-1).contiguous(),
                        self.ncols, similarity=self.similarity, row0=self.row0)


# ----------------------------------------------------------------------------------------------
# Persistence of device graphs without per-row Python objects.  The reference
# pickles the DataFrame, `Neighbours` column of N tuples included (prograph/utils/save.py:5-39);
# at N = 1M that is a million small arrays.  A graph is three flat arrays: they go into one .npz.
# ----------------------------------------------------------------------------------------------
def fingerprint(tokens):
    """64-bit fingerprint of a token matrix (shape + an order-sensitive checksum): what ties a graph side-car to the
    data it was built from."""
    t = np.ascontiguousarray(np.asarray(tokens)).astype(np.uint8, copy=False)
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        w = (np.arange(1, t.shape[1] + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))[None, :] if t.ndim == 2 and t.size else np.zeros((1, 0), np.uint64)
        rows = (t.astype(np.uint64) * w).sum(axis=1) if t.size else np.zeros(0, np.uint64)
        idx = np.arange(1, len(rows) + 1, dtype=np.uint64) * np.uint64(0xBF58476D1CE4E5B9)
        h = h ^ np.uint64((rows * idx).sum()) ^ (np.uint64(t.shape[0]) << np.uint64(32)) ^ np.uint64(t.shape[1] if t.ndim == 2 else 0)
    return int(h)


def save_graphs(path, graphs, tokens_fingerprint=None):
    """{name: CSRGraph | KNNGraph} -> one .npz (arrays `<name>/indptr|indices|weights` or `<name>/idx|dist`
    plus a small meta vector [kind, ncols, similarity, row0]); `__fingerprint__` = fingerprint of the token matrix."""
    out = {}
    if tokens_fingerprint is not None:
        out["__fingerprint__"] = np.array([tokens_fingerprint], dtype=np.uint64)
    for name, g in graphs.items():
        if isinstance(g, KNNGraph):
            out[f"{name}/idx"] = g.idx.cpu().numpy()
            out[f"{name}/dist"] = g.dist.cpu().numpy()
            out[f"{name}/meta"] = np.array([1, g.ncols, int(g.similarity), g.row0], dtype=np.int64)
        else:
            out[f"{name}/indptr"] = g.indptr.cpu().numpy()
            out[f"{name}/indices"] = g.indices.cpu().numpy()
            out[f"{name}/weights"] = g.weights.cpu().numpy()
            out[f"{name}/meta"] = np.array([0, g.ncols, int(g.similarity), g.row0], dtype=np.int64)
    np.savez(path, **out)


def load_graphs(path, device=None, tokens_fingerprint=None):
    """Inverse of save_graphs (no pickled objects inside: plain arrays, `allow_pickle=False`).  With
    `tokens_fingerprint` a side-car written for other data (or by a version without fingerprints) yields nothing;
    graphs whose arrays are not a well-formed CSR / kNN table over `ncols` columns are skipped (device kernels index
    by these columns)."""
    device = _native.device() if device is None else device
    z = np.load(path, allow_pickle=False)
    graphs = {}
    if tokens_fingerprint is not None:
        if "__fingerprint__" not in z.files or int(z["__fingerprint__"][0]) != int(tokens_fingerprint):
            return graphs
    for key in z.files:
        if not key.endswith("/meta"):
            continue
        name = key[:-5]
        kind, ncols, sim, row0 = (int(v) for v in z[key])
        if kind == 1:
            idx = z[f"{name}/idx"]
            if idx.ndim != 2 or z[f"{name}/dist"].shape != idx.shape or (idx.size and (idx.min() < -1 or idx.max() >= ncols)):
                continue
        else:
            ip, ix = z[f"{name}/indptr"], z[f"{name}/indices"]
            if (ip.ndim != 1 or len(ip) < 1 or ip[0] != 0 or np.any(np.diff(ip) < 0) or int(ip[-1]) != len(ix)
                    or len(z[f"{name}/weights"]) != len(ix) or (len(ix) and (ix.min() < 0 or ix.max() >= ncols))):
                continue
        t = lambda a: torch.from_numpy(np.ascontiguousarray(z[f"{name}/{a}"])).to(device)
        if kind == 1:
            graphs[name] = KNNGraph(t("idx"), t("dist"), ncols, similarity=bool(sim), row0=row0)
        else:
            graphs[name] = CSRGraph(t("indptr"), t("indices"), t("weights"), ncols, similarity=bool(sim), row0=row0)
    return graphs
