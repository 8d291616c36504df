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
    """epsilon-neighbourhood graph: indptr int64 [n+1], indices int32 [nnz], weights uint8|float32 [nnz]."""

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
        return self.weights

    def row_stats(self, f=None, boolean_weights=False, want=("deg",)):
        """Device reductions per row (pg_csr_row_stats): deg = sum w, sum_f = sum f[col], sum_wf = sum w f[col]."""
        return _native.csr_row_stats(self.indptr, self.indices, self._w(boolean_weights), f=f, want=want)

    def degree(self, boolean_weights=False):
        """Out-degree per row as float32 (prograph.py:797-822) without touching Python tuples."""
        if boolean_weights:
            return (self.indptr[1:] - self.indptr[:-1]).to(torch.float32).cpu().numpy()
        return self.row_stats(want=("deg",))["deg"].to(torch.float32).cpu().numpy()

    def dirichlet(self, f, boolean_weights=False):
        """f^T L f with L = D_out - A (prograph.py:874-922), f = per-node values of the ROWS' nodes
        (square graph: nrows == ncols)."""
        fd = torch.as_tensor(np.asarray(f, dtype=np.float64).reshape(-1), device=self.indptr.device)
        st = self.row_stats(f=fd, boolean_weights=boolean_weights, want=("deg", "sum_wf"))
        fr = fd[self.row0:self.row0 + self.nrows]
        # the reference keeps the degree in float32 (prograph.py:815) before it enters the Laplacian
        deg = st["deg"].to(torch.float32).to(torch.float64)
        return float((fr * (deg * fr - st["sum_wf"])).sum().item())

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
