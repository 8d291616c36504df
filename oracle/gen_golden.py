#!/usr/bin/env python3
"""
Golden-vector generator — TEST INFRASTRUCTURE.  Runs ONLY in the build container, where
the read-only reference checkout exists at /root/reference; its outputs (small .npz
files under tests/golden/) are committed, the reference itself never travels.

It imports the real acmater/prograph with two in-process shims (ordinary Python errors,
nothing was permission-denied — SURVEY.md §8c):
  * `colorama` is not installed -> a 2-attribute stand-in module is registered in
    sys.modules before the import (only used to colour one printed string,
    prograph/prograph.py:15,516);
  * `build_graph` hard-codes `torch.device("cuda:0")` (prograph/prograph.py:726) and this
    container has no GPU -> the module-level name `torch` inside prograph.prograph is
    replaced by a proxy whose `.device(...)` returns the CPU device and which forwards
    everything else to torch.  For canonical kNN *indices* the proxy additionally routes
    `.sort` to `torch.sort(..., stable=True)`; kNN *weights* are taken from the unpatched
    run and asserted equal.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import io
import contextlib
import operator
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

import numpy as np
import torch

col = types.ModuleType("colorama")
col.Fore = types.SimpleNamespace(GREEN="", RED="")
col.Style = types.SimpleNamespace(RESET_ALL="")
sys.modules["colorama"] = col
sys.path.insert(0, REF)
import prograph.prograph as refmod          # noqa: E402
from prograph import Prograph               # noqa: E402
from prograph.distance import hamming as ref_hamming   # noqa: E402


class TorchProxy:
    stable = False

    def __getattr__(self, n):
        return getattr(torch, n)

    @staticmethod
    def device(*a, **k):
        return torch.device("cpu")

    def sort(self, *a, **k):
        if self.stable:
            k["stable"] = True
        return torch.sort(*a, **k)


proxy = TorchProxy()
refmod.torch = proxy

from prograph_amd import synth              # noqa: E402


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
        return fn(*a, **k)


def csr(neigh):
    counts = np.array([len(a[0]) for a in neigh], dtype=np.int64)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    idx = np.concatenate([np.asarray(a[0], dtype=np.int64) for a in neigh]) if indptr[-1] else np.zeros(0, np.int64)
    w = np.concatenate([np.asarray(a[1]) for a in neigh]) if indptr[-1] else np.zeros(0, np.int64)
    return indptr, idx, w


def store_eps(d, name, neigh):
    indptr, idx, w = csr(neigh)
    assert all(a[0].dtype == np.int64 for a in neigh)
    d[name + "_indptr"] = indptr
    d[name + "_indices"] = idx.astype(np.int32)
    if np.issubdtype(w.dtype, np.integer):
        d[name + "_weights"] = w.astype(np.uint8)
        assert np.array_equal(d[name + "_weights"].astype(np.int64), w)
    else:
        d[name + "_weights"] = w


def knn_both(pg, **kw):
    proxy.stable = False
    a = quiet(pg.build_graph, **kw)
    proxy.stable = True
    b = quiet(pg.build_graph, **kw)
    proxy.stable = False
    wa = np.stack([x[1] for x in a]); wb = np.stack([x[1] for x in b])
    assert np.array_equal(wa, wb), "kNN weights must not depend on sort stability"
    return np.stack([x[0] for x in b]), wb


def store_knn(d, name, pg, **kw):
    idx, w = knn_both(pg, **kw)
    d[name + "_idx"] = idx.astype(np.int32)
    d[name + "_w"] = w.astype(np.uint8) if np.issubdtype(w.dtype, np.integer) else w


def make_csv(tmp, name, tokens, seed):
    import pandas as pd
    seqs = synth.tokens_to_strings(tokens)
    fit = (synth.h(seed, 7, np.arange(len(seqs))) % np.uint64(10 ** 6)).astype(np.float64) / 1e6
    p = os.path.join(tmp, name + ".csv")
    pd.DataFrame({"Sequence": seqs, "Fitness": fit}).to_csv(p)
    return p


def store_analytics(d, pg, graph, prefix):
    """Graph-signal quantities of the REAL reference for one stored graph column (prograph/prograph.py:
    797-946): degree, Laplacian diagonal (both degree modes), Dirichlet energy, local variance -
    weighted and boolean.  The device path (pg_csr_row_stats) is checked against these."""
    import warnings
    for bw, tag in ((False, "w"), (True, "b")):
        d[f"{prefix}_deg_{tag}"] = np.asarray(pg.degree(graph, boolean_weights=bw))
        for mode in ("outdegree", "indegree"):
            d[f"{prefix}_lapdiag_{tag}_{mode[:3]}"] = np.asarray(pg.laplacian(graph, boolean_weights=bw, mode=mode).diagonal(), dtype=np.float64)
            d[f"{prefix}_dirichlet_{tag}_{mode[:3]}"] = np.asarray(pg.dirichlet(graph, boolean_weights=bw, mode=mode), dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d[f"{prefix}_locvar"] = np.asarray(pg.local_variance(graph), dtype=np.float64)


def gen_reference_csv():
    os.chdir(REF)
    pg = quiet(Prograph, file="data/synthetic_data.csv")
    d = {}
    d["tokens"] = pg.tokenized.astype(np.uint8)
    d["fitness"] = pg("Fitness").to_numpy()
    store_eps(d, "eps1", list(pg.graph["Neighbours"]))
    store_eps(d, "eps2", quiet(pg.build_graph, eps=2))
    store_eps(d, "eps3", quiet(pg.build_graph, eps=3))
    for nm, op in [("eq", operator.eq), ("lt", operator.lt), ("ge", operator.ge), ("gt", operator.gt)]:
        store_eps(d, "eps2_" + nm, quiet(pg.build_graph, eps=2, comp=op))
    store_eps(d, "eps1_b5", quiet(pg.build_graph, eps=1, batch_size=5))
    sub = np.arange(100, 400)
    d["sub_idxs"] = sub
    store_eps(d, "eps1_sub", quiet(pg.build_graph, eps=1, idxs=sub))
    store_eps(d, "eps1_sim", quiet(pg.build_graph, eps=1, similarity=True))
    for k in (1, 2, 16):
        store_knn(d, f"knn{k}", pg, k=k)
    store_knn(d, "knn4_sim", pg, k=4, similarity=True)
    store_knn(d, "knn3_sub", pg, k=3, idxs=sub)
    # indexing known answers (tests/tests.py:42-53, 92-98)
    d["ix_pos12"] = pg.indexing(positions=[1, 2])
    d["ix_pos12_and"] = pg.indexing(positions=[1, 2], Bool="and")
    d["ix_d3"] = pg.indexing(distances=3)
    d["ix_d2"] = pg.indexing(distances=2)
    d["ix_d13"] = pg.indexing(distances=[1, 3])
    d["ix_pos12_d2"] = pg.indexing(positions=[1, 2], distances=2)
    a, b = pg.indexing(positions=[1, 2], distances=2, complement=True)
    d["ix_pos12_d2_c0"], d["ix_pos12_d2_c1"] = a, b
    d["ix_LDC_pos1"] = pg.indexing(reference_seq="LDC", positions=[1])
    d["ix_LDC_d1"] = pg.indexing(reference_seq="LDC", distances=1)
    d["LDC_idx"] = np.int64(pg.query("LDC"))
    d["ACL_idx"] = np.int64(pg.query("ACL"))
    d["calc_neigh_ACL"] = pg.calc_neighbours(seq="ACL")
    d["calc_neigh_ACL_le2"] = pg.calc_neighbours(seq="ACL", eps=2, comp=operator.le)
    d["degree"] = pg.degree()
    d["adj33"] = np.asarray(pg.adjacency().todense()[:3, :3])
    store_analytics(d, pg, "Neighbours", "ana_eps1")
    proxy.stable = True
    pg.graph["K5"] = quiet(pg.build_graph, k=5)
    pg.graph["K4s"] = quiet(pg.build_graph, k=4, similarity=True)
    proxy.stable = False
    pg.graph["E2s"] = quiet(pg.build_graph, eps=2, similarity=True)
    store_analytics(d, pg, "K5", "ana_knn5")
    store_analytics(d, pg, "K4s", "ana_knn4sim")
    store_analytics(d, pg, "E2s", "ana_eps2sim")
    dd = ref_hamming(pg.tokenized, pg.tokenized[pg.query(pg.seed.Sequence)].reshape(1, -1))
    d["str_maxdist"] = np.int64(int(torch.max(dd)))
    d["str_ndist"] = np.int64(len(np.unique(dd)))
    d["dist_to_seed"] = dd.numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "ref_synthetic_csv.npz"), **d)
    print("ref_synthetic_csv", {k: v.shape for k, v in list(d.items())[:6]})


def gen_set(tmp, name, tokens, seed, eps_list, k_list, extra=None):
    p = make_csv(tmp, name, tokens, seed)
    pg = quiet(Prograph, file=p)
    d = {"tokens": pg.tokenized.astype(np.uint8)}
    assert np.array_equal(d["tokens"], tokens), "tokenize(strings) must reproduce the tokens"
    store_eps(d, "eps1", list(pg.graph["Neighbours"]))
    for e in eps_list:
        store_eps(d, f"eps{e}", quiet(pg.build_graph, eps=e))
    for k in k_list:
        store_knn(d, f"knn{k}", pg, k=k)
    ref_row = len(tokens) // 3
    dd = ref_hamming(pg.tokenized, pg.tokenized[ref_row].reshape(1, -1))
    d["ref_row"] = np.int64(ref_row)
    d["dist_to_ref"] = dd.numpy().astype(np.uint8)
    if extra:
        extra(pg, d)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, tokens.shape, "eps nnz", {e: int(d[f'eps{e}_indptr'][-1]) for e in eps_list})


def gen_hamming_kats():
    d = {}
    rng = np.random.RandomState(7)
    # literal known answers of tests/tests.py:175-186
    X = torch.Tensor([[1, 2, 3], [4, 5, 6]]); Y = torch.Tensor([[1, 2, 3], [7, 8, 9]])
    d["kat_2d2d"] = ref_hamming(X, Y).numpy()
    d["kat_2d1d"] = ref_hamming(X, torch.Tensor([1, 2, 3])).numpy()
    d["kat_1d1d"] = ref_hamming(torch.Tensor([4, 5, 6]), torch.Tensor([1, 2, 3])).numpy()
    # random integer blocks, incl. unequal D (zero padding, distance/utils.py:32-38)
    for i, (n, m, dx, dy) in enumerate([(37, 5, 11, 11), (64, 9, 20, 13), (10, 70, 7, 16), (129, 3, 64, 64),
                                        (50, 8, 130, 130), (33, 33, 200, 150)]):
        A = rng.randint(0, 21, size=(n, dx)).astype(np.int64)
        Bm = rng.randint(0, 21, size=(m, dy)).astype(np.int64)
        d[f"r{i}_X"] = A.astype(np.uint8); d[f"r{i}_Y"] = Bm.astype(np.uint8)
        d[f"r{i}_out"] = ref_hamming(A, Bm).numpy().astype(np.int64)
    A = rng.randint(0, 256, size=(40, 24)).astype(np.int64)       # full byte range
    Bm = rng.randint(0, 256, size=(6, 24)).astype(np.int64)
    Bm[0] = A[3]
    d["wide_X"] = A.astype(np.uint8); d["wide_Y"] = Bm.astype(np.uint8)
    d["wide_out"] = ref_hamming(A, Bm).numpy()
    A = rng.randint(-5, 70000, size=(30, 9)).astype(np.int64)     # needs 32-bit elements
    Bm = A[rng.randint(0, 30, size=7)].copy(); Bm[:, ::3] += 1
    d["i32_X"] = A; d["i32_Y"] = Bm
    d["i32_out"] = ref_hamming(A, Bm).numpy()
    d["sim_out"] = ref_hamming(d["r0_X"].astype(np.int64), d["r0_Y"].astype(np.int64), similarity=True).numpy()
    np.savez_compressed(os.path.join(OUT, "hamming_kats.npz"), **d)
    print("hamming_kats", len(d))


KNNTEST_EMBEDDING = np.array([[0, 0], [0.5, 0], [3.75, 0], [2.75, 0], [4.0, 1.4], [4.75, 1.15]], dtype=np.float64)


def gen_minkowski():
    """`build_graph(representation="Embedded", distance=minkowski)` of the real reference
    (prograph/distance/minkowski.py:8-41 through prograph/prograph.py:726-764: fp16 staging, fp16
    elementwise arithmetic).  data/knntest_pgraph.pkl is a pickle and is not loaded; the 2-D embedding
    below was constructed to satisfy every known answer tests/tests.py:139-167 pins on that file
    (asserted here against the reference itself), the sequences / fitness come from data/knntest.csv."""
    from prograph.distance import minkowski as ref_mink
    d = {}
    os.chdir(REF)
    pg = quiet(Prograph, file="data/knntest.csv")
    pg.graph["Embedded"] = list(KNNTEST_EMBEDDING)
    proxy.stable = True
    L1 = quiet(pg.build_graph, representation="Embedded", k=1, distance=ref_mink)
    L2 = quiet(pg.build_graph, representation="Embedded", k=2, distance=ref_mink)
    assert np.all(np.array([x[0] for x in L1]).reshape(-1) == np.array([1, 0, 3, 2, 5, 4]))               # tests.py:141-144
    assert np.all(np.array([x[0] for x in L2]) == np.array([[1, 3], [0, 3], [3, 4], [2, 4], [5, 2], [4, 2]]))   # :145-148
    pg.graph["Weighted"] = quiet(pg.build_graph, eps=2, representation="Embedded", distance=ref_mink)
    assert np.allclose(pg.degree(graph="Weighted", boolean_weights=True), [1, 1, 3, 2, 3, 2])            # :159-162
    pg.graph["W1"] = L1
    assert np.allclose(pg.degree(graph="W1"), [0.5, 0.5, 1., 1., 0.79052734, 0.79052734], atol=1e-7)       # :164-167
    d["knntest_emb"] = KNNTEST_EMBEDDING
    d["knntest_fitness"] = pg("Fitness").to_numpy()
    for k in (1, 2, 5):
        L = quiet(pg.build_graph, representation="Embedded", k=k, distance=ref_mink)
        d[f"knntest_knn{k}_idx"] = np.stack([x[0] for x in L]).astype(np.int32)
        d[f"knntest_knn{k}_w"] = np.stack([x[1] for x in L])
    store_eps(d, "knntest_eps2", list(pg.graph["Weighted"]))
    d["knntest_deg_w1"] = pg.degree(graph="W1")
    # seeded fp16 sets through the same reference path (D = 2, 64, 1280)
    with tempfile.TemporaryDirectory() as tmp:
        for name, n, dim, seed in (("d2", 300, 2, 1), ("d64", 1000, 64, 2), ("d1280", 512, 1280, 3)):
            rng = np.random.RandomState(seed)
            centres = rng.randn(max(1, n // 20), dim)
            emb = (centres[rng.randint(0, len(centres), size=n)] + 0.35 * rng.randn(n, dim)).astype(np.float16)
            emb[n // 2] = emb[3]                                   # an exact duplicate (d == 0 is excluded, rank 0 rule)
            tok = synth.clustered_tokens(n, 8, seed=synth.DEFAULT_SEED + 40 + seed)
            pgs = quiet(Prograph, file=make_csv(tmp, "mink_" + name, tok, 20 + seed))
            pgs.graph["Embedded"] = list(emb)
            d[f"{name}_emb"] = emb
            dist = ref_mink(torch.as_tensor(emb, dtype=torch.float16), torch.as_tensor(emb[:64], dtype=torch.float16))
            d[f"{name}_dist64"] = dist.numpy()                     # (64, n) fp16 block of the operator itself
            eps = float(np.float16(np.quantile(dist.numpy().astype(np.float64), 0.03)))
            d[f"{name}_eps"] = np.float64(eps)
            store_eps(d, f"{name}_eps", quiet(pgs.build_graph, eps=eps, representation="Embedded", distance=ref_mink))
            store_eps(d, f"{name}_eps_sim", quiet(pgs.build_graph, eps=eps, similarity=True, representation="Embedded", distance=ref_mink))
            for k in (1, 5, 16):
                L = quiet(pgs.build_graph, representation="Embedded", k=k, distance=ref_mink)
                d[f"{name}_knn{k}_idx"] = np.stack([x[0] for x in L]).astype(np.int32)
                d[f"{name}_knn{k}_w"] = np.stack([x[1] for x in L])
            L = quiet(pgs.build_graph, representation="Embedded", k=4, similarity=True, distance=ref_mink)
            d[f"{name}_knn4_sim_idx"] = np.stack([x[0] for x in L]).astype(np.int32)
            d[f"{name}_knn4_sim_w"] = np.stack([x[1] for x in L])
    proxy.stable = False
    np.savez_compressed(os.path.join(OUT, "minkowski_f16.npz"), **d)
    print("minkowski_f16", {k: (v.shape, str(v.dtype)) for k, v in d.items() if "knn5" in k or "eps_w" in k})


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_minkowski()
    gen_reference_csv()
    gen_hamming_kats()
    with tempfile.TemporaryDirectory() as tmp:
        t1 = synth.clustered_tokens(1000, 32, seed=synth.DEFAULT_SEED)
        gen_set(tmp, "synth_n1000_l32", t1, 11, [2], [16, 5])

        t2 = synth.clustered_tokens(2085, 64, seed=synth.DEFAULT_SEED + 1)
        gen_set(tmp, "synth_n2085_l64", t2, 12, [2, 4], [16])

        t3 = synth.clustered_tokens(515, 20, seed=synth.DEFAULT_SEED + 2, members=64)
        for r in (6, 100, 514):
            t3[r] = t3[5]                     # exact duplicates before/after their twin
        t3[200] = t3[0]

        def extra3(pg, d):
            sub = np.array([5, 6, 7, 100, 101, 300, 514, 0, 200, 33], dtype=np.int64)
            d["sub_idxs"] = sub
            store_eps(d, "eps2_sub", quiet(pg.build_graph, eps=2, idxs=sub))
            store_knn(d, "knn3_sub", pg, k=3, idxs=sub)
            # duplicates + kNN + similarity: rows whose list contains the row itself (Laplacian setdiag case)
            d["fitness"] = pg("Fitness").to_numpy()
            store_analytics(d, pg, "Neighbours", "ana_eps1")
            proxy.stable = True
            pg.graph["K3s"] = quiet(pg.build_graph, k=3, similarity=True)
            pg.graph["K16"] = quiet(pg.build_graph, k=16)
            proxy.stable = False
            store_knn(d, "knn3_sim", pg, k=3, similarity=True)
            store_analytics(d, pg, "K3s", "ana_knn3sim")
            store_analytics(d, pg, "K16", "ana_knn16")
        gen_set(tmp, "synth_n515_l20_dups", t3, 13, [2, 3], [1, 16, 40], extra=extra3)

        t4, l4 = synth.clustered_varlen_tokens(300, Lmax=24, Lmin=12, seed=synth.DEFAULT_SEED + 3, members=50)
        # the reference's seed row fixes seq_len; make row 0 full length so tokenize pads to 24
        t4[0] = 1 + (np.arange(24) % 20); l4[0] = 24

        def extra4(pg, d):
            d["lengths"] = l4
        gen_set(tmp, "synth_n300_varlen24", t4, 14, [2, 6], [8], extra=extra4)
    made = [x for d_, _, fs in os.walk(REF) for x in fs if x.endswith(".pyc")]
    assert not made, "bytecode was written into the reference checkout"


if __name__ == "__main__":
    main()
