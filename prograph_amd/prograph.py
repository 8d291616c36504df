"""
`Prograph` — host-side mirror of the reference's class (prograph/prograph.py of
acmater/prograph) with the graph-construction hot path running on hand-written HIP kernels.

Same constructor, same methods, same return conventions, so code written against the
reference keeps working; what changed is *where the work happens*:

  reference (prograph/prograph.py)                     here
  -----------------------------------------------      ------------------------------------
  :726  list-of-arrays -> fp16 tensor on cuda:0        int8 plane layout, packed on device once
  :731-739 batches of 8: broadcast !=, sum, where,     one `pg_eps_slots` launch + scan +
           3 D->H copies and a sync per batch          `pg_eps_compact`; one sync in total
  :756-762 full torch.sort per row                     `pg_knn_hamming` (in-register top-(k+1))
  :298-325 1xN hamming + numpy set logic               `pg_index_flags` + `pg_compact_flags`

There is no CPU fallback for that path: without the HIP library or a GPU the calls raise
`prograph_amd._native.NativeUnavailable`.  A custom `distance` callable, a `comp` outside the
five orderings, non-byte representations (e.g. fp16 embeddings with `minkowski`), k > 63 or
L > 128 go through `_build_graph_generic`, which is the reference's batch loop kept on torch
ops on the GPU — the distance-operator protocol stays pluggable.

kNN tie rule: the reference's `torch.sort` (:758-760) is unstable, so with integer distances
its neighbour *indices* are implementation defined.  This implementation fixes the canonical
order (distance, index) == `torch.sort(stable=True)`, drops rank 0 like the reference does
and returns ranks 1..k.  Weights are bit-identical to the reference either way.
"""
import copy
import operator
import os

import numpy as np
import pandas as pd
import torch

from . import _native
from .distance import hamming, minkowski
from .graph import CSRGraph, KNNGraph
from .protein import Protein
from .utils import Dataset, flatten

_CMP_CODE = {operator.le: _native.CMP_LE, operator.lt: _native.CMP_LT, operator.eq: _native.CMP_EQ,
             operator.ge: _native.CMP_GE, operator.gt: _native.CMP_GT}
_DEFAULT_SCALER = object()      # "use sklearn's MinMaxScaler" without importing sklearn at module import


class Prograph:
    def __init__(self, file, seed_seq=None, seqs_col="Sequence", columns=["Fitness"], index_col=0,
                 amino_acids="ACDEFGHIKLMNPQRSTVWY"):
        try:
            ext = file.split(".")[-1]
            if ext == "csv":
                self.graph = self.csvDataLoader(file, seqs_col=seqs_col, columns=columns, index_col=index_col)
            elif ext == "pkl":
                self.graph = pd.read_pickle(file)
            else:
                raise ValueError(ext)
        except Exception:
            # the reference turns every load problem (missing file, file=None, file=2, ...) into this
            raise FileNotFoundError("File could not be opened")

        self.file, self.seed_seq, self.seqs_col = file, seed_seq, seqs_col
        self.columns, self.index_col, self.amino_acids = columns, index_col, amino_acids

        self.seed = Protein(seed_seq) if seed_seq else Protein(**self.graph.loc[0])
        self.seq_len = len(self.seed)
        self.len = len(self)

        self.tokens = {aa.encode("utf-8"): i for i, aa in enumerate(self.amino_acids, start=1)}
        self._planes = {}            # device-resident plane layouts, keyed by representation
        self.tokenized = self._ingest_tokens(self.graph[seqs_col])
        self._token_dict = None
        self.seq_idxs = dict(zip(self.graph[seqs_col], range(len(self.graph))))   # last duplicate wins

        self.mutated_positions = self.calc_mutated_positions()
        self.sequence_mutation_locations = self.boolean_mutant_array(self.seed.Sequence)
        self.mutation_arrays = self.gen_mutation_arrays()
        self.csr_graphs = {}         # name -> CSRGraph / KNNGraph kept on the device
        self._csr_rows = {}          # name -> row objects of the column the device graph answers for

        if "Tokenized" not in self.graph:
            self.graph["Tokenized"] = list(self.tokenized)
        if isinstance(file, str) and ext == "pkl":
            self._restore_graphs(os.path.splitext(file)[0] + ".graphs.npz")
        if "Neighbours" not in self.graph:
            self.graph["Neighbours"] = self.build_graph(eps=1, _keep="Neighbours")

        self.learners = {}
        print(self)

    # ------------------------------------------------------------------ dunder / query surface
    def __str__(self):
        hist = self._distance_histogram(self.seed.Sequence)
        present = np.nonzero(hist)[0]
        longest = max((len(s) for s in self("Sequence")), default=0)
        return f"""
            Prograph
            Number of Sequences : {len(self)}
            Max Distance        : {int(present[-1])}
            Longest Sequence    : {longest}
            Number of Distances : {len(present)}
            Seed Sequence       : {self.coloured_seed_string()}
                Modified positions are shown in green"""

    def __repr__(self):
        return (f"Prograph(file={self.file},\n seed_seq='{self.seed.Sequence}',\n seqs_col='{self.seqs_col}',\n"
                f" columns={self.columns},\n index_col={self.index_col},\n amino_acids='{self.amino_acids}')")

    def __len__(self):
        return len(self.graph)

    def __getitem__(self, idx):
        return self.graph.iloc[self.query(idx)]

    def __call__(self, label=None, **kwargs):
        return self.label_iter(label, **kwargs)

    def label_iter(self, label, **kwargs):
        """`pgraph("Sequence")`, `pgraph("sklearn", ...)`, `pgraph("pytorch", ...)`, `pgraph()`; copies."""
        if label == "pytorch":
            return self.pytorch_dataloaders(**kwargs)
        if label == "sklearn":
            return self.sklearn_data(**kwargs)
        if label is None:
            return self.graph.copy()
        return self.graph[label].copy()

    @property
    def token_dict(self):
        """{tuple(tokens): index}; built on first use (O(N*L) Python objects, unused by the hot path)."""
        if self._token_dict is None:
            self._token_dict = {tuple(seq): i for i, seq in enumerate(self.tokenized)}
        return self._token_dict

    def query(self, sequence):
        """int / str / token tuple / list or array of those -> positional index (reference :204-240)."""
        missing = "This sequence is not in the dataset."
        if isinstance(sequence, (int, np.integer)):
            assert sequence <= self.len, "Index exceeds bounds of dataset"
            return sequence
        if isinstance(sequence, (np.ndarray, list)):
            first = sequence[0]
            if isinstance(first, (int, np.integer, np.bool_)):
                return sequence
            if isinstance(first, str):
                return [self.seq_idxs.get(s, missing) for s in sequence]
            print("Wrong data format in numpy array or list iterable.")
            return None
        if isinstance(sequence, str):
            return self.seq_idxs.get(sequence, missing)
        if isinstance(sequence, tuple):
            assert len(sequence) == self.seq_len, "Tuple not valid length for dataset."
            hits = np.where(np.all(np.asarray(sequence) == self.tokenized, axis=1))[0]
            assert len(hits) > 0, "Not a valid tuple representation of a protein in this dataset."
            return int(hits[0]) if len(hits) == 1 else int(hits)
        raise ValueError("Input format not understood.")

    # ------------------------------------------------------------------ ingest / tokenisation
    @staticmethod
    def csvDataLoader(csvfile, seqs_col, columns="all", index_col=None):
        data = pd.read_csv(csvfile, index_col=index_col)
        if columns == "all":
            columns = [c for c in data.keys() if c != seqs_col]
        wanted = [seqs_col] + list(columns)
        if "Neighbours" in data:
            wanted.append("Neighbours")
        return data[wanted]

    def _byte_view(self, sequences):
        """The fixed-width byte view of the strings ((N, Lmax) uint8, NUL padded: numpy 'S' storage) and the
        256-entry letter table (letter j of `amino_acids` -> j+1, everything else -> 0; reference :127)."""
        arr = np.array(sequences, dtype="bytes").reshape(-1)
        width = max(arr.dtype.itemsize, 1)
        table = np.zeros(256, dtype=int)
        for ch, tok in self.tokens.items():
            if len(ch) == 1:
                table[ch[0]] = tok
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(len(arr), width) if (len(arr) and arr.dtype.itemsize) else \
            np.zeros((len(arr), width), dtype=np.uint8)
        return raw, table

    def tokenize(self, sequences):
        """
        Strings -> (N, Lmax) int tokens: letter j of `amino_acids` -> j+1, padding / unknown -> 0
        (reference :454-474).  One table lookup over the fixed-width byte view instead of one
        masked pass per letter.
        """
        raw, table = self._byte_view(sequences)
        return table[raw]

    def _ingest_tokens(self, sequences):
        """
        The constructor's tokenisation.  With a GPU the byte view goes to the device once and ONE
        kernel (`pg_pack_bytes`) applies the letter table, bit-slices the tokens into the plane layout the graph
        kernels read and writes the token matrix: the planes are cached for `build_graph` / `indexing`, the host's
        `self.tokenized` is that matrix copied back (uint8) and widened - the table pass over N x L bytes on the
        host is gone.  Without a device (the host-logic tests' fake backend), or beyond the kernels' widths, the host
        table lookup of `tokenize` is used.
        """
        raw, table = self._byte_view(sequences)
        bits = _native.BITS_5 if len(self.amino_acids) <= 31 else _native.BITS_8
        limit = _native.MAX_L_5BIT if bits == _native.BITS_5 else _native.MAX_L
        try:
            on_gpu = _native.device().type == "cuda" and hasattr(_native, "pack_bytes")
        except _native.NativeUnavailable:
            on_gpu = False
        if not on_gpu or raw.shape[0] == 0 or raw.shape[1] > limit or len(self.amino_acids) > 255:
            return table[raw]
        planes, tok = _native.pack_bytes(raw, table.astype(np.uint8), bits=bits, want_tokens=True)
        self._planes["Tokenized"] = planes
        self._tok_u8_dev = tok
        return tok.cpu().numpy().astype(int)

    def custom_tokenize(self, seq, tokenizer=None):
        if tokenizer is None:
            return np.array([self.tokens[aa.encode("utf-8")] for aa in seq])
        return "This feature is not ready yet"

    def embedding(self, embedded, name):
        self.graph[f"{name}_embedded"] = embedded

    def boolean_mutant_array(self, seq=None):
        return self.tokenized != self.tokenized[self.query(seq)]

    def calc_mutated_positions(self):
        varies = ~np.all(self.tokenized == self.tokenize(self.seed.Sequence), axis=0)
        return np.nonzero(varies[: len(self.seed)])[0]

    def coloured_seed_string(self):
        try:
            from colorama import Fore, Style
            on, off = Fore.GREEN, Style.RESET_ALL
        except ImportError:
            on = off = ""
        marked = set(int(i) for i in self.mutated_positions)
        return "".join(f"{on}{c}{off}" if i in marked else c for i, c in enumerate(self.seed.Sequence))

    def gen_mutation_arrays(self):
        n_aa = len(self.amino_acids)
        xs = np.arange(self.seq_len * n_aa)
        ys = np.repeat(np.arange(self.seq_len), n_aa)
        modifiers = np.tile(np.arange(n_aa), self.seq_len)
        return xs, ys, modifiers

    def generate_mutations(self, seq):
        seq = self.tokenized[self.query(seq)]
        xs, ys, mutations = self.mutation_arrays
        variants = np.tile(np.asarray(seq, dtype=float), (len(xs), 1))
        variants[xs, ys] = mutations
        return variants[~np.all(variants == seq, axis=1)]

    def get_mutated_positions(self, positions):
        for pos in positions:
            assert pos in self.mutated_positions, "{} is not a position that was mutated in this dataset".format(pos)
        constants = np.setdiff1d(self.mutated_positions, positions)
        return np.all(~self.sequence_mutation_locations[:, constants], axis=1)

    def get_data(self, tokenized=False):
        if tokenized:
            return np.array([x[["Sequence", "Fitness"]] for x in self.graph])
        return copy.copy(self.tokenized)

    # ------------------------------------------------------------------ device residency
    def _byte_planes(self, representation="Tokenized", idxs=None):
        """Plane layout of a byte-token representation on the GPU.  Only the token matrix of the
        constructor is cached (`self.tokenized` never changes); any other representation is a
        DataFrame column the user may reassign (`embedding()`, `pg.graph[name] = ...`) and is read and
        packed on every call, as the reference re-reads `self(representation)` (prograph.py:726)."""
        key = representation
        if idxs is None and key == "Tokenized" and key in self._planes:
            return self._planes[key]
        if representation == "Tokenized":
            mat = self.tokenized
        else:
            mat = np.vstack(self(representation))
        mat = np.asarray(mat)
        if not np.issubdtype(mat.dtype, np.integer):
            raise ValueError("not an integer representation")
        if mat.size and mat.min() >= 0 and mat.max() <= 255:
            mat = mat.astype(np.uint8)               # 8x less PCIe traffic than the int64 token matrix
        # tokens of the built-in tokeniser are 0..len(amino_acids): 5 bit planes cover 31 letters
        bits = _native.BITS_5 if (representation == "Tokenized" and len(self.amino_acids) <= 31) else None
        planes = _native.pack(torch.from_numpy(np.ascontiguousarray(mat)), rows=idxs, bits=bits)
        if idxs is None and key == "Tokenized":
            self._planes[key] = planes
        return planes

    def _planes_or_none(self):
        """The cached plane layout, or None when the token matrix is beyond the fused kernels' limits
        (more than 255 positions, 128 for alphabets above 31 symbols): the queries below then run on
        the native dense operator (`hamming`, any length) plus torch ops on the GPU."""
        try:
            return self._byte_planes()
        except (ValueError, TypeError):
            return None

    def _tokens_dev(self):
        if getattr(self, "_tok_dev", None) is None:
            self._tok_dev = torch.as_tensor(np.ascontiguousarray(self.tokenized), device=_native.device())
        return self._tok_dev

    def _row_distances_long(self, ref):
        """(N,) int64 Hamming distances of every sequence to row `ref`, long-sequence path."""
        T = self._tokens_dev()
        return hamming(T, T[ref:ref + 1]).reshape(-1)

    def _distance_histogram(self, reference_seq):
        planes = self._planes_or_none()
        ref = int(self.query(reference_seq))
        if planes is None:
            return torch.bincount(self._row_distances_long(ref)).cpu().numpy()
        _, hist, _ = _native.index_flags(planes, ref, want_dist_out=False, want_flags=False)
        return hist.cpu().numpy()

    # ------------------------------------------------------------------ indexing
    def positions(self, positions):
        return self.indexing(positions=positions)

    def distances(self, distances):
        return self.indexing(distances=distances)

    def indexing(self, reference_seq=None, distances=None, positions=None, percentage=None, Bool="or",
                 complement=False):
        """
        Distance-k / mutated-position index queries (reference :254-343), evaluated in one fused
        1xN HIP pass: Hamming distance of every sequence to the reference row, membership of that
        distance in `distances`, the position logic, then a device stream compaction to the
        ascending index array.  `percentage` / `complement` are the reference's numpy post-steps.
        """
        assert Bool == "or" or Bool == "and", "Not a valid boolean value."
        if reference_seq is None:
            reference_seq = self.seed.Sequence
        ref = int(self.query(reference_seq))
        planes = self._planes_or_none()

        want = None
        if distances is not None:
            if type(distances) == int:
                distances = [distances]
            assert type(distances) == list, "Distances must be provided as integer or list"
            hist = self._distance_histogram(reference_seq)
            for d in distances:
                assert isinstance(d, (int, np.integer)) and 0 <= d < len(hist) and hist[d] > 0, f"{d} is not a valid distance"
            want = distances

        pos_mode, pos_mask, not_mask = 0, None, None
        if positions is not None:
            ref_len = len(self[reference_seq]["Sequence"])
            ncol = self.tokenized.shape[1]
            for p in positions:
                if not -ncol <= p < ncol:
                    raise IndexError(f"index {p} is out of bounds for axis 1 with size {ncol}")
            pos_mask = [p % ncol for p in positions]
            not_mask = [p for p in range(ref_len) if p not in positions]
            if len(positions) == 0:
                raise TypeError("reduce() of empty sequence with no initial value")
            pos_mode = 1 if Bool == "or" else 2

        if want is None and pos_mode == 0:
            idxs = np.array(range(len(self)))
        elif planes is None:
            # long sequences: the same logic (reference :298-325) with torch ops on the GPU
            T = self._tokens_dev()
            keep = torch.ones(len(self), dtype=torch.bool, device=T.device)
            if want is not None:
                keep &= torch.isin(self._row_distances_long(ref), torch.as_tensor(want, device=T.device))
            if pos_mode:
                mut = T != T[ref:ref + 1]
                sel = mut[:, pos_mask]
                working = sel.any(dim=1) if pos_mode == 1 else sel.all(dim=1)
                if not_mask:
                    working &= ~mut[:, not_mask].any(dim=1)
                keep &= working
            idxs = torch.nonzero(keep).reshape(-1).cpu().numpy()
        else:
            _, _, flags = _native.index_flags(planes, ref, want=want, pos_mode=pos_mode, pos_mask=pos_mask,
                                              not_mask=not_mask, want_dist_out=False, want_hist=False)
            idxs = _native.compact_flags(flags).cpu().numpy()

        if percentage is not None:
            assert 0 <= percentage <= 1, "Percentage must be between 0 and 1"
            keep = np.zeros(len(idxs), dtype=bool)
            keep[np.random.choice(np.arange(len(idxs)), size=int(len(idxs) * percentage), replace=False)] = 1
            return idxs[keep]

        assert len(idxs) != 0, "No possible valid indices have been provided."
        if complement:
            return idxs, np.setdiff1d(np.arange(self.len), idxs)
        return idxs

    def calc_neighbours(self, seq, eps=1, distance=hamming, comp=operator.eq, weights=False):
        """Column indices with comp(distance to `seq`, eps) (reference :526-544)."""
        if distance is hamming and comp in _CMP_CODE:
            planes = self._planes_or_none()
            if planes is None:
                d = self._row_distances_long(int(self.query(seq)))
                return torch.nonzero(comp(d, eps)).reshape(-1).cpu().numpy()
            want = [d for d in range(256) if comp(d, eps)]
            _, _, flags = _native.index_flags(planes, int(self.query(seq)), want=want,
                                              want_dist_out=False, want_hist=False)
            return _native.compact_flags(flags).cpu().numpy()
        d = distance(self.tokenized, self.tokenized[self.query(seq)].reshape(1, -1))
        return np.where(comp(d, eps))[1]

    def neighbourhood(self, seq, eps, distance=hamming):
        """All rows within `eps` of `seq`, the row itself included (reference :571-588)."""
        planes = self._planes_or_none()
        if planes is None:
            dist = self._row_distances_long(int(self.query(seq)))
        else:
            dist, _, _ = _native.index_flags(planes, int(self.query(seq)), want_hist=False, want_flags=False)
        return self[(dist <= eps).cpu().numpy().flatten()]

    def neighbourhood_clustering(self, eps, distance=hamming):
        clusters, seen = {}, set()
        for i, seq in enumerate(self("Sequence")):
            if i not in seen:
                members = self.neighbourhood(seq, eps, distance)
                clusters[i] = members
                seen |= set(members.index)
        return clusters

    @staticmethod
    def get_every_n(a, n=2):
        for start in range(0, a.shape[0], n):
            yield a[start:start + n]

    @staticmethod
    def prod_neighbours(index, out, batch_size, weights=None):
        """COO of one batch -> {row: (cols, weights)} (reference :626-654); only the generic path needs it."""
        row, col = out
        row = row + index * batch_size
        if weights is None:
            weights = np.ones(col.shape)
        result = {}
        if len(row):
            cuts = np.nonzero(np.diff(row))[0] + 1          # rows arrive grouped and ascending from where()
            for r, c, w in zip(row[np.r_[0, cuts]], np.split(col, cuts), np.split(weights, cuts)):
                result[r] = (c, w)
        return result

    # ------------------------------------------------------------------ graph construction
    def build_graph(self, idxs=None, batch_size=8, eps=None, k=None, weighted=False, similarity=False,
                    representation="Tokenized", distance=hamming, comp=operator.le, output="tuples", cap=256,
                    store=None, _keep=None):
        """
        epsilon-neighbourhood (`eps`) or kNN (`k`) graph over all pairwise distances
        (reference :656-765).  Returns the reference's list of N `(indices, weights)` tuples;
        `output="csr"` returns the device-resident `CSRGraph` / `KNNGraph` instead (no per-row
        Python objects — what large N wants).  `batch_size` is accepted for compatibility: the
        HIP path tiles the pair space itself.  `cap` = slot capacity per row of the fused pass
        (rows with more matches are recomputed exactly, so it only affects speed).  `store="Name"`
        also assigns the result to `self.graph["Name"]` and keeps the device CSR, so that `degree`,
        `dirichlet`, `local_variance`, `adjacency` on that name run from the CSR on the GPU.
        """
        if operator.xor(bool(eps), bool(k)) is False:
            raise ValueError("Epsilon or K must be provided, but both cannot be as they are different methods of graph construction.")
        if k is not None and not isinstance(k, int):
            raise TypeError("K must be provided as an integer.")

        if idxs is not None:
            # the reference indexes a tensor with `idxs` (prograph.py:726): integer lists, boolean
            # masks and slices all select rows; normalise to integer positions
            if isinstance(idxs, slice):
                idxs = np.arange(len(self))[idxs]
            else:
                idxs = np.asarray(idxs)
                if idxs.dtype == bool:
                    idxs = np.nonzero(idxs)[0]
        if distance is minkowski and comp in _CMP_CODE and (k is None or k <= _native.MAX_K):
            if output == "csr":
                raise NotImplementedError("output='csr' (device-resident graphs) exists for the Hamming path only; "
                                          "Minkowski graphs are returned as the reference's tuples")
            out = self._build_graph_minkowski(idxs, eps, k, similarity, representation, comp)
            if out is not None:
                if store is not None and idxs is None:
                    self.graph[store] = out       # (no device graph is kept: the consumers take the column path)
                    self.csr_graphs.pop(store, None)
                return out
        native = distance is hamming and (comp in _CMP_CODE) and (k is None or k <= _native.MAX_K_ROUNDS)
        planes = None
        if native:
            try:
                planes = self._byte_planes(representation, idxs)
            except (ValueError, TypeError):
                native = False                      # not byte tokens / L > 128: generic torch path
        if native and k and planes.n > _native.MAX_N_KNN:
            native = False
        g = None
        if not native and distance is hamming and comp in _CMP_CODE and (k is None or k <= _native.MAX_K):
            g = self._build_graph_long(idxs, eps, k, similarity, representation, comp)
        if not native and g is None:
            return self._build_graph_generic(idxs, batch_size, eps, k, similarity, representation, distance, comp)

        if g is not None:
            pass                                    # sequences beyond one record of the fused engines: built above
        elif eps:
            # similarity: comp(1/(1+eps), 1/(1+d)) & (s < 1) is the mirrored integer test on d
            # (:720-721, :734); both sides are the same correctly rounded float32 quotient when d == eps
            indptr, indices, wts = _native.eps_graph(planes, planes, _CMP_CODE[comp], eps, cap=cap)
            g = CSRGraph(indptr, indices, wts, planes.n, similarity=similarity)
        else:
            idx, dist = _native.knn_graph(planes, planes, k)
            g = KNNGraph(idx, dist, planes.n, similarity=similarity)
        tuples = None
        if store is not None and idxs is None:
            tuples = g.to_tuples()
            self.graph[store] = tuples
            _keep = store
        if _keep is not None and idxs is None:
            if tuples is None:
                tuples = g.to_tuples()
            self.csr_graphs[_keep] = g
            # the device CSR answers for the column only while the column still holds THESE row objects
            self._csr_rows[_keep] = self._row_ids(tuples)
        if output == "csr":
            return g
        return tuples if tuples is not None else g.to_tuples()

    _LONG_MAX_L = 2048                               # integers up to here are exact in fp16

    def _build_graph_long(self, idxs, eps, k, similarity, representation, comp):
        """
        Graphs of byte-token sequences LONGER than one record of the fused engines (more than 255 positions,
        128 for alphabets above 31 symbols; the reference has no limit: hamming.py:32-34 is a broadcast).
        Per block of rows the distance matrix comes from the dense kernel - the sequence cut into column
        segments of whole records, their distances accumulated in place (`pg_hamming_dense`, fp16 output:
        integers up to 2048 are exact) - and the selection runs on the device as for fp16 embeddings: the
        canonical (distance, column) ranks 1..k (`pg_f16_knn`; :756-763) or the thresholded CSR
        (`pg_f16_eps_*`; :734-739).  Returns a KNNGraph / CSRGraph with int16 weights, or None when the
        representation is not byte tokens or longer than 2048 positions (the generic batch loop then).
        """
        try:
            mat = self.tokenized if representation == "Tokenized" else np.vstack(self(representation))
            mat = np.asarray(mat)
        except (ValueError, TypeError):
            return None
        if mat.ndim != 2 or not np.issubdtype(mat.dtype, np.integer) or mat.shape[0] == 0 or mat.shape[1] == 0:
            return None
        if mat.shape[1] > self._LONG_MAX_L or mat.min() < 0 or mat.max() > 255:
            return None
        if idxs is not None:
            mat = mat[np.asarray(idxs)]
        dev = _native.device()
        T = torch.as_tensor(np.ascontiguousarray(mat.astype(np.uint8)), device=dev)
        n, l = T.shape
        bits = _native.BITS_5 if int(mat.max()) <= 31 else _native.BITS_8
        w = (_native.MAX_L_5BIT if bits == _native.BITS_5 else _native.MAX_L) // 32 * 32   # whole 32-token groups
        segs = [(a, min(l, a + w)) for a in range(0, l, w)]
        xs = [_native.pack(T[:, a:b], bits=bits) for a, b in segs]
        # comp(d, eps) on integer distances as an integer threshold (exact in fp16 whatever eps is);
        # similarity graphs: comp(1/(1+eps), 1/(1+d)) & (s < 1) is the same test on d (:720-721, :734)
        cmp = _CMP_CODE[comp]
        if eps:
            e = float(eps)
            lo, hi = int(np.floor(e)), int(np.ceil(e))
            thr = {_native.CMP_LE: lo, _native.CMP_LT: hi, _native.CMP_GE: hi, _native.CMP_GT: lo}.get(cmp, lo if lo == hi else -1)
            thr = float(min(max(thr, -1), 4096))
        rows_per_block = max(64, min(n, (1 << 27) // n))                  # <= 256 MB of fp16 distances at a time
        kk = min(k, n - 1) if k else 0
        parts = []
        for r0 in range(0, n, rows_per_block):
            r1 = min(n, r0 + rows_per_block)
            block = None
            for (a, b), xp in zip(segs, xs):
                block = _native.hamming_dense(xp, _native.pack(T[r0:r1, a:b], bits=bits), out_bytes=2, out=block)
            if k:
                if kk:
                    idx, wt = _native.f16_knn(block, kk, first=1, descending=False)
                    parts.append((idx, wt.to(torch.int16)))
            else:
                indptr, indices, wts = _native.f16_eps(block, cmp, thr, similarity=False)
                parts.append((indptr, indices, wts.to(torch.int16)))
            del block
        if k:
            if not kk:
                return KNNGraph(torch.zeros((n, 0), dtype=torch.int32, device=dev), torch.zeros((n, 0), dtype=torch.int16, device=dev),
                                n, similarity=similarity)
            return KNNGraph(torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts]), n, similarity=similarity)
        base, ptrs = 0, [torch.zeros(1, dtype=torch.int64, device=dev)]
        for indptr, _, _ in parts:
            ptrs.append(indptr[1:] + base)
            base += int(indptr[-1].item())
        return CSRGraph(torch.cat(ptrs), torch.cat([p_[1] for p_ in parts]), torch.cat([p_[2] for p_ in parts]), n,
                        similarity=similarity)

    def _build_graph_minkowski(self, idxs, eps, k, similarity, representation, comp):
        """
        `build_graph(representation=<embedding>, distance=minkowski)` on the HIP kernels:
        the fp16 staging of the reference (:726), then per block of rows the fp16 distance matrix
        (`pg_minkowski_dense`, rounding like the reference's fp16 tensor ops) and the selection on the
        device - the canonical (value, column) ranks 1..k (`pg_f16_knn`; the reference drops sorted rank
        0, :761-763) or the thresholded CSR (`pg_f16_eps_*`; :734-739).  Returns None when the
        representation is not a numeric matrix (the generic path then reports as the reference would).
        """
        try:
            mat = np.vstack(self(representation))
            X = torch.as_tensor(mat, dtype=torch.float16, device=_native.device())
        except (ValueError, TypeError):
            return None
        if X.dim() != 2 or X.shape[0] == 0 or X.shape[1] == 0 or not X.is_cuda:
            return None
        if idxs is not None:
            X = X[torch.as_tensor(np.asarray(idxs), device=X.device)]
        n = X.shape[0]
        if similarity and eps:
            eps = 1 / (1 + eps)                                          # :720-721
        xp = _native.pack_f16(X)
        rows_per_block = max(64, min(n, (1 << 27) // max(n, 1)))          # <= 256 MB of fp16 distances at a time
        out = []
        empty = (np.array([], dtype=int), np.array([], dtype=int))
        for r0 in range(0, n, rows_per_block):
            yp = _native.pack_f16(X[r0:r0 + rows_per_block])
            block = _native.minkowski_dense(xp, yp, similarity=similarity)
            if k:
                kk = min(k, n - 1)
                idx, w = _native.f16_knn(block, kk, first=1, descending=similarity) if kk else (None, None)
                if kk:
                    idx, w = idx.to(torch.int64).cpu().numpy(), w.cpu().numpy()
                    out.extend(zip(list(idx), list(w)))
                else:
                    out.extend((np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float16)) for _ in range(block.shape[0]))
            else:
                indptr, indices, wts = _native.f16_eps(block, _CMP_CODE[comp], eps, similarity=similarity)
                ip, ix, ww = indptr.cpu().numpy(), indices.to(torch.int64).cpu().numpy(), wts.cpu().numpy()
                out.extend((ix[a:b], ww[a:b]) if b > a else empty for a, b in zip(ip[:-1].tolist(), ip[1:].tolist()))
        return out

    def _build_graph_generic(self, idxs, batch_size, eps, k, similarity, representation, distance, comp):
        """
        The distance-operator protocol for everything outside the byte-token Hamming path:
        any callable `distance(X, Y, similarity=...) -> (M,N)` (README.md:48 of the reference).
        Same batch loop and fp16 staging as the reference (:726-764), on the GPU, with a stable
        sort for the canonical tie order.
        """
        if similarity and eps:
            eps = 1 / (1 + eps)
        dev = _native.device()
        X = torch.as_tensor(np.vstack(self(representation)), dtype=torch.float16, device=dev)
        if idxs is not None:
            X = X[idxs, :]
        if distance is hamming and len(X):
            # our own operator (sequences beyond the fused engine's limits): batching does not change
            # its result, so take row blocks of up to 2^26 distances instead of the reference's 8 rows
            batch_size = max(batch_size, min(4096, (1 << 26) // len(X)))
        weights, edges = [], []
        if eps:
            for batch in self.get_every_n(X, n=batch_size):
                d = distance(X, batch, similarity=similarity)
                loc = torch.where(comp(eps, d) & (d < 1)) if similarity else torch.where(comp(d, eps) & (d > 0))
                weights.append(d[loc].cpu().numpy())
                edges.append([x.cpu().numpy() for x in loc])
            merged = {}
            for i, coo in enumerate(edges):
                merged.update(self.prod_neighbours(i, coo, batch_size, weights=weights[i]))
            empty = (np.array([], dtype=int), np.array([], dtype=int))
            return [merged.get(i, empty) for i in range(len(X))]
        for batch in self.get_every_n(X, n=batch_size):
            s = torch.sort(distance(X, batch, similarity=similarity), dim=1, descending=bool(similarity), stable=True)
            weights.append([x.cpu().numpy() for x in s[0][:, 1:k + 1]])
            edges.append([x.cpu().numpy() for x in s[1][:, 1:k + 1]])
        return list(zip(flatten(edges), flatten(weights)))

    def _restore_graphs(self, sidecar):
        """Graphs saved as flat arrays by `utils.save(..., graphs="csr")`: back onto the device, and their
        columns (the reference's tuple format, views into two host arrays per graph) into the frame."""
        if not os.path.exists(sidecar):
            return
        from .graph import load_graphs, fingerprint
        for name, g in load_graphs(sidecar, tokens_fingerprint=fingerprint(self.tokenized)).items():
            if name in self.graph or g.nrows != len(self.graph):
                continue
            tuples = g.to_tuples()
            self.graph[name] = tuples
            self.csr_graphs[name] = g
            self._csr_rows[name] = self._row_ids(tuples)

    def _device_graph_any(self, graph):
        """The device-resident form (CSRGraph or KNNGraph as built) of a column that still matches it."""
        return self.csr_graphs.get(graph) if self._device_graph(graph) is not None else None

    # ------------------------------------------------------------------ consumers of the graph column
    def _device_graph(self, graph):
        """The device-resident CSR of a graph column built by this object, if it still matches the
        column (a user may have overwritten `self.graph[name]` with something else)."""
        g = self.csr_graphs.get(graph)
        if g is None or graph not in self.graph:
            return None
        g = g.as_csr() if isinstance(g, KNNGraph) else g
        col = self.graph[graph]
        if len(col) != g.nrows or g.nrows != g.ncols:
            return None
        # identity, not shape: a user who overwrote the column (even with a graph of the same structure,
        # e.g. similarity weights instead of distances) stored other row objects
        # (every row up to 262 144 rows - an in-place replacement of ANY row's tuple is seen; 1024 evenly spaced rows beyond,
        #  where walking a million Python objects would cost more than the device analytics save)
        keep, ids = self._csr_rows.get(graph, (None, None))
        if ids is None or len(ids) != len(col):
            return None
        now = self._row_ids(col.values)[1]
        pick = slice(None) if len(ids) <= 262144 else np.linspace(0, len(ids) - 1, 1024).astype(np.int64)
        if not np.array_equal(now[pick], ids[pick]):
            return None
        return g

    @staticmethod
    def _row_ids(rows):
        """(the row objects - kept alive, so their ids stay theirs -, their ids): which tuples a graph column holds.
        Beyond 262 144 rows only 1024 evenly spaced ones are looked at."""
        n = len(rows)
        if n <= 262144:
            return list(rows), np.fromiter(map(id, rows), dtype=np.int64, count=n)
        ids = np.zeros(n, dtype=np.int64)
        pick = np.linspace(0, n - 1, 1024).astype(np.int64)
        held = [rows[i] for i in pick]
        ids[pick] = [id(o) for o in held]
        return held, ids

    def _column_csr(self, graph):
        """(indptr, indices, weights) numpy arrays of a Neighbours-style column."""
        col = self(graph)
        counts = np.fromiter((len(e[0]) for e in col), dtype=np.int64, count=len(col))
        indptr = np.concatenate([[0], np.cumsum(counts)])
        if indptr[-1] == 0:
            return indptr, np.zeros(0, dtype=int), np.zeros(0)
        return indptr, np.concatenate([e[0] for e in col]), np.concatenate([e[1] for e in col])

    def degree(self, graph="Neighbours", boolean_weights=False):
        g = self._device_graph(graph)
        if g is not None:
            return g.degree(boolean_weights)
        indptr, _, w = self._column_csr(graph)
        if boolean_weights:
            return np.diff(indptr).astype(np.float32)
        deg = np.zeros(len(self), dtype=np.float32)
        rows = np.repeat(np.arange(len(self)), np.diff(indptr))
        np.add.at(deg, rows, w.astype(np.float32))
        return deg

    def get_neighbour_coords(self, graph="Neighbours", boolean_weights=False):
        g = self._device_graph(graph)
        if g is not None:
            return g.coords(boolean_weights)
        indptr, J, w = self._column_csr(graph)
        I = np.repeat(np.arange(len(self), dtype=int), np.diff(indptr))
        if boolean_weights:
            return I, J, np.ones(I.shape)
        return I, J, w.astype(np.float32)

    def adjacency(self, graph="Neighbours", boolean_weights=False):
        from scipy import sparse
        I, J, V = self.get_neighbour_coords(graph=graph, boolean_weights=boolean_weights)
        return sparse.coo_matrix((V, (I, J)), shape=(len(self), len(self)))

    def laplacian(self, graph="Neighbours", boolean_weights=False, mode="outdegree"):
        L = (-1) * self.adjacency(graph, boolean_weights)
        if mode == "outdegree":
            D = self.degree(graph, boolean_weights)
        elif mode == "indegree":
            D = (-1) * np.array(L.sum(0)).reshape(-1,)
        else:
            raise ValueError("Not a valid degree mode.")
        L.setdiag(D)
        return L

    def dirichlet(self, graph="Neighbours", boolean_weights=False, scaler=_DEFAULT_SCALER, mode="outdegree"):
        if scaler is _DEFAULT_SCALER:
            from sklearn.preprocessing import MinMaxScaler as scaler
        fitness = self("Fitness").to_numpy().reshape(-1, 1)
        if scaler is not None:
            fitness = scaler().fit_transform(fitness)
        g = self._device_graph(graph) if mode in ("outdegree", "indegree") else None
        if g is not None:
            return np.array([[g.dirichlet(fitness, boolean_weights, mode=mode)]])
        L = self.laplacian(graph=graph, boolean_weights=boolean_weights, mode=mode)
        return fitness.T @ L @ fitness

    def local_variance(self, graph="Neighbours", boolean_weights=False, scaler=_DEFAULT_SCALER):
        if scaler is _DEFAULT_SCALER:
            from sklearn.preprocessing import MinMaxScaler as scaler
        f = scaler().fit_transform(self("Fitness").to_numpy().reshape(-1, 1)).reshape(-1)
        g = self._device_graph(graph)
        if g is not None:
            return g.local_variance(f)
        indptr, J, _ = self._column_csr(graph)
        out = np.full(len(self), np.nan)
        rows = np.repeat(np.arange(len(self)), np.diff(indptr))
        sums = np.zeros(len(self))
        np.add.at(sums, rows, f[rows] - f[J])
        has = np.diff(indptr) > 0
        out[has] = sums[has] / np.diff(indptr)[has]
        return out

    def graph_to_networkx(self, graph="Neighbours", labels=None, update_self=False, iterable="Sequence"):
        import networkx as nx
        names = list(self(iterable))
        label_cols = [self(l) for l in labels] if labels is not None else []
        g = nx.Graph()
        for i, name in enumerate(names):
            g.add_node(name, **{labels[j]: label_cols[j][i] for j in range(len(label_cols))})
        for i, (nbrs, _) in enumerate(self(graph)):
            g.add_edges_from((names[i], names[j]) for j in nbrs)
        if update_self:
            self.networkx_graph = g
            return None
        return g

    # ------------------------------------------------------------------ data access for ML frameworks
    def _resolve_idxs(self, idxs, distance, positions):
        """README.md:36-40 of the reference documents `distance=` / `positions=` on the data
        accessors but never wires them; here they are forwarded to `indexing`."""
        if idxs is None and (distance not in (None, False) or positions is not None):
            idxs = self.indexing(distances=None if distance in (None, False) else distance, positions=positions)
        return idxs

    def _xy(self, representation, labels, idxs):
        reps = self(representation)
        if idxs is not None:
            X = np.vstack(reps[idxs])
            y = np.vstack([self(l)[idxs] for l in labels]).T
        else:
            X = np.vstack(reps)
            y = np.vstack([self(l) for l in labels]).T
        return X, y

    def sklearn_data(self, data=None, idxs=None, representation="Tokenized", labels=["Fitness"],
                     split=[0.8, 0, 0.2], scaler=False, shuffle=True, random_state=0, distance=None, positions=None):
        import sklearn.utils as skutils
        if isinstance(split, int):
            split = [split, 0, 1 - split]
        assert sum(split) <= 1, "The sum of the split terms must be between 0 and 1"
        idxs = self._resolve_idxs(idxs, distance, positions)
        tokenized, labels = self._xy(representation, labels, idxs)
        if shuffle:
            tokenized, labels = skutils.shuffle(tokenized, labels, random_state=random_state)
        if scaler:
            labels = scaler.fit_transform(labels.reshape(-1, 1)).reshape(-1)
        labels = labels.ravel()
        a, b = int(len(tokenized) * split[0]), int(len(tokenized) * sum(split[:2]))
        parts = [tokenized[:a], labels[:a], tokenized[a:b], labels[a:b], tokenized[b:], labels[b:]]
        return tuple(p.astype("float") for p in parts)

    def gen_dataloaders(self, labels, keys, params, split_points):
        a, b = split_points
        out = {}
        for name, part in (("train", keys[:a]), ("val", keys[a:b]), ("test", keys[b:])):
            if len(part):
                out[name] = torch.utils.data.DataLoader(Dataset(part, labels), **params)
        return out

    def pytorch_dataloaders(self, split=[0.8, 0, 0.2], idxs=None, representation="Tokenized", labels=["Fitness"],
                            distance=False, positions=None,
                            params={"batch_size": 500, "shuffle": True, "num_workers": 8},
                            unsupervised=False, real_label=0):
        idxs = self._resolve_idxs(idxs, distance, positions)
        tokenized, labels = self._xy(representation, labels, idxs)
        keys = [torch.Tensor(t.astype("float32")).long() for t in tokenized]
        if unsupervised:
            data_labels = {key: real_label for key in keys}
        else:
            data_labels = {key: lab for key, lab in zip(keys, labels)}
        cuts = [int(len(tokenized) * split[0]), int(len(tokenized) * sum(split[:2]))]
        return self.gen_dataloaders(labels=data_labels, keys=list(data_labels.keys()), params=params, split_points=cuts)

    def fit(self, model, model_args, save_model=False, **kwargs):
        x_train, y_train, _, _, x_test, y_test = self("sklearn", **kwargs)
        model = model(**model_args)
        if model.__class__.__name__ == "NeuralNetRegressor":
            y_train, y_test = y_train.reshape(-1, 1), y_test.reshape(-1, 1)
        print(f"Training model {model}")
        model.fit(x_train, y_train)
        train_score = model.score(x_train, y_train)
        print(f"Model score on training data: {train_score}")
        test_score = model.score(x_test, y_test)
        print(f"Score of {model} on testing data is {test_score}")
        if save_model:
            self.learners[f"{model}"] = model
        return train_score, test_score
