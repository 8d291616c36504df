"""
The reference's own test-suite (tests/tests.py of acmater/prograph, 50 methods) restated
against `prograph_amd.Prograph`, plus golden-vector checks of the drop-in surface.

Every test runs twice:
  * backend "hip"  (marked gpu): the real HIP kernels through the C ABI;
  * backend "fake" (CPU):        tests/fake_native.py answers the native calls with the oracle,
                                 so only the HOST logic is under test there.
The input file is `data/synthetic_data.csv` of the reference, rebuilt from the committed
golden vector (tokens + fitness) because /root/reference does not exist on the GPU box.
"""
import operator
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden
from prograph_amd import synth

NW0 = {"batch_size": 500, "shuffle": True, "num_workers": 0}


@pytest.fixture(params=[pytest.param("fake"), pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    if request.param == "fake":
        import fake_native
        fake_native.install(monkeypatch)
    else:
        from prograph_amd import _native
        _native.lib(); _native.device()
    return request.param


@pytest.fixture
def csv_path(tmp_path):
    g = load_golden("ref_synthetic_csv")
    p = tmp_path / "data"
    p.mkdir()
    f = p / "synthetic_data.csv"
    pd.DataFrame({"Sequence": synth.tokens_to_strings(g["tokens"]), "Fitness": g["fitness"]}).to_csv(f)
    return str(f)


@pytest.fixture
def pgraph(backend, csv_path, capsys):
    from prograph_amd import Prograph
    pg = Prograph(file=csv_path)
    out = capsys.readouterr().out
    assert "Number of Sequences : 1000" in out and "Max Distance        : 3" in out
    assert "Number of Distances : 4" in out and "Longest Sequence    : 3" in out
    return pg


# ---------------------------------------------------------------- tests/tests.py:15-23, 104-114
def test_gen_pgraph_errors(backend):
    from prograph_amd import Prograph
    with pytest.raises(TypeError):
        Prograph()
    with pytest.raises(FileNotFoundError):
        Prograph(file="None.pkl")
    with pytest.raises(FileNotFoundError):
        Prograph(file=2)
    with pytest.raises(FileNotFoundError):
        Prograph(file=None)


# ---------------------------------------------------------------- tests/tests.py:27-39
def test_query_forms(pgraph):
    assert pgraph["AAC"]["Sequence"] == "AAC"
    assert pgraph("Sequence")[26] == "ADH"
    assert pgraph[(1, 2, 2)]["Sequence"] == "ACC"
    assert len(pgraph) == 1000
    assert pgraph[[1, 2, 4]]["Sequence"][2] == "AAD"
    assert pgraph[np.array([63, 87])]["Sequence"][87] == "AKI"
    with pytest.raises(KeyError):
        pgraph.label_iter("PLK")                                   # tests/tests.py:87-90


# ---------------------------------------------------------------- tests/tests.py:41-53, 92-98
def test_indexing_known_answers(pgraph):
    g = load_golden("ref_synthetic_csv")
    assert len(pgraph.indexing(positions=[1, 2])) == 99
    assert len(pgraph.indexing(distances=3)) == 729
    assert len(pgraph.indexing(positions=[1, 2], distances=2)) == 81 and len(pgraph.indexing(distances=2)) == 243
    assert len(pgraph.indexing(percentage=0.7)) == 700
    assert len(pgraph.indexing(positions=[1, 2], distances=2, percentage=0.3)) == 24
    assert pgraph.indexing(positions=[1, 2], distances=2, complement=True)[1][12] == 30
    with pytest.raises(AssertionError):
        pgraph.indexing(distances=[1, 2, 4])
    assert len(pgraph.indexing(distances=[1, 3])) == 756
    assert pgraph[pgraph.indexing(reference_seq="LDC", positions=[1])]["Sequence"][901] == "LAC"
    # bit-exact against the reference's outputs
    for key, kw in [("ix_pos12", dict(positions=[1, 2])), ("ix_pos12_and", dict(positions=[1, 2], Bool="and")),
                    ("ix_d3", dict(distances=3)), ("ix_d2", dict(distances=2)), ("ix_d13", dict(distances=[1, 3])),
                    ("ix_pos12_d2", dict(positions=[1, 2], distances=2)),
                    ("ix_LDC_pos1", dict(reference_seq="LDC", positions=[1])),
                    ("ix_LDC_d1", dict(reference_seq="LDC", distances=1))]:
        got = pgraph.indexing(**kw)
        assert got.dtype == np.int64 and np.array_equal(got, g[key]), key
    a, b = pgraph.indexing(positions=[1, 2], distances=2, complement=True)
    assert np.array_equal(a, g["ix_pos12_d2_c0"]) and np.array_equal(b, g["ix_pos12_d2_c1"])
    assert np.array_equal(pgraph.positions([1, 2]), g["ix_pos12"]) and np.array_equal(pgraph.distances(3), g["ix_d3"])
    with pytest.raises(AssertionError):
        pgraph.indexing(Bool="xor")
    with pytest.raises(AssertionError):
        pgraph.indexing(distances="3")
    with pytest.raises(AssertionError):
        pgraph.indexing(percentage=1.5)
    assert np.array_equal(pgraph.indexing(), np.arange(1000))


# ---------------------------------------------------------------- tests/tests.py:63-64
def test_calc_neighbours(pgraph):
    g = load_golden("ref_synthetic_csv")
    assert np.all(pgraph.calc_neighbours(seq="ACL") == pgraph["ACL"]["Neighbours"][0])
    assert np.array_equal(pgraph.calc_neighbours(seq="ACL"), g["calc_neigh_ACL"])
    assert np.array_equal(pgraph.calc_neighbours(seq="ACL", eps=2, comp=operator.le), g["calc_neigh_ACL_le2"])
    nb = pgraph.neighbourhood("ACL", 1)
    assert len(nb) == 28 and "ACL" in set(nb["Sequence"])


# ---------------------------------------------------------------- tests/tests.py:66-85
def test_pytorch_dataloaders(pgraph):
    train, test = pgraph.pytorch_dataloaders(params=NW0).values()
    assert len(next(iter(test))[0]) == 200
    train, test = pgraph.pytorch_dataloaders(idxs=np.arange(10), params=NW0).values()
    assert len(next(iter(test))[0]) == 2
    train, test = pgraph.pytorch_dataloaders(idxs=pgraph.indexing(distances=1), params=NW0).values()
    assert len(next(iter(test))[0]) == 6
    train, test = pgraph.pytorch_dataloaders(idxs=pgraph.indexing(distances=[1, 2]), params=NW0).values()
    assert len(next(iter(test))[0]) == 54
    train, test = pgraph.pytorch_dataloaders(idxs=pgraph.indexing(positions=[1, 2]), params=NW0).values()
    assert len(next(iter(test))[0]) == 20
    train, test = pgraph.pytorch_dataloaders(unsupervised=True, params=NW0).values()
    batch = next(iter(test))
    assert torch.all(0 == batch[1]) and len(batch[1]) == 200
    # README.md:36-40 of the reference: distance= / positions= on the accessors
    train, test = pgraph("pytorch", positions=[1, 2], params=NW0).values()
    assert len(next(iter(test))[0]) == 20
    xtr, ytr, _, _, xte, yte = pgraph("sklearn", distance=2, positions=[1, 2])
    assert len(xtr) + len(xte) == 81 and xtr.dtype == np.float64
    xtr, ytr, xv, yv, xte, yte = pgraph("sklearn")
    assert (len(xtr), len(xv), len(xte)) == (800, 0, 200)


# ---------------------------------------------------------------- tests/tests.py:100-122
def test_networkx_and_save_roundtrip(pgraph, tmp_path, capsys):
    from prograph_amd import Prograph
    from prograph_amd.utils import save
    pgraph.graph_to_networkx(labels=["Fitness", "Tokenized"], update_self=True)
    assert "Fitness" in pgraph.networkx_graph.nodes["AAA"].keys()
    assert save(pgraph, name="test", directory=str(tmp_path) + "/")
    again = Prograph(file=str(tmp_path / "test.pkl"))
    assert again[0]["Sequence"] == "AAA"
    assert np.array_equal(again["ACL"]["Neighbours"][0], pgraph["ACL"]["Neighbours"][0])
    assert again[0]["Fitness"] == 0.660972597708149               # tests/tests.py:108


def test_csr_persistence_without_tuples(pgraph, tmp_path, capsys):
    """`save(graphs="csr")` writes the graphs as flat arrays (no N tuples in the pickle),
    the constructor restores them from the side-car and - like the reference with a pickled `Neighbours`
    column (prograph/prograph.py:140-141) - does not run the N^2 build again."""
    from prograph_amd import Prograph
    from prograph_amd.utils import save
    pgraph.build_graph(k=5, store="K5")
    pgraph.build_graph(eps=2, similarity=True, store="E2s")
    assert save(pgraph, name="flat", directory=str(tmp_path) + "/", graphs="csr")
    frame = pd.read_pickle(tmp_path / "flat.pkl")
    assert "Neighbours" not in frame and "K5" not in frame and "E2s" not in frame and "Sequence" in frame
    z = np.load(tmp_path / "flat.graphs.npz", allow_pickle=False)        # plain arrays only
    assert {"Neighbours/indptr", "Neighbours/indices", "Neighbours/weights", "K5/idx", "K5/dist", "E2s/weights"} <= set(z.files)
    assert z["Neighbours/indptr"][-1] == 27000 and z["K5/idx"].shape == (1000, 5)

    def no_build(self, *a, **k):
        raise AssertionError("the constructor rebuilt a graph that was saved")
    real_build = Prograph.build_graph
    Prograph.build_graph = no_build
    try:
        again = Prograph(file=str(tmp_path / "flat.pkl"))
    finally:
        Prograph.build_graph = real_build
    for name in ("Neighbours", "K5", "E2s"):
        a, b = list(again.graph[name]), list(pgraph.graph[name])
        assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[1].dtype == y[1].dtype for x, y in zip(a, b))
        assert again._device_graph(name) is not None
        assert np.array_equal(again.degree(name), pgraph.degree(name))
    assert np.allclose(again.dirichlet("K5"), pgraph.dirichlet("K5"), rtol=1e-12)


# ---------------------------------------------------------------- tests/tests.py:124-133
def test_tokenization(pgraph):
    assert np.all(pgraph.tokenize("ACA") == np.array([1, 2, 1]))
    assert np.all(pgraph.tokenize(["ACA", "ACC"]) == np.array([[1, 2, 1], [1, 2, 2]]))
    tokens = pgraph.tokenize(["ACCCACAAA", "ACAA"])
    assert np.all(tokens == np.array([[1, 2, 2, 2, 1, 2, 1, 1, 1], [1, 2, 1, 1, 0, 0, 0, 0, 0]]))
    assert len(pgraph.tokenize([])) == 0
    from oracle import prograph_oracle as O
    seqs = ["ACDEFGHIKLMNPQRSTVWY", "XYZ-", "", "WWWWWWWWWWWWWWWWWWWWWWWW"]
    assert np.array_equal(pgraph.tokenize(seqs), O.tokenize(seqs)) and pgraph.tokenize(seqs).dtype == O.tokenize(seqs).dtype
    assert np.array_equal(pgraph.tokenized, load_golden("ref_synthetic_csv")["tokens"])
    assert pgraph.token_dict[(1, 2, 2)] == pgraph.query("ACC")


# ---------------------------------------------------------------- tests/tests.py:135-137, 157-158
def test_matrix_and_degree(pgraph):
    g = load_golden("ref_synthetic_csv")
    assert np.all(pgraph.adjacency().todense()[:3, :3] == np.array([[0, 1, 1], [1, 0, 1], [1, 1, 0]]))
    assert np.all(pgraph.degree() == np.array([27 for _ in range(1000)]))
    assert pgraph.degree().dtype == np.float32 and np.array_equal(pgraph.degree(), g["degree"])
    assert np.array_equal(pgraph.degree(boolean_weights=True), g["degree"])
    assert np.array_equal(pgraph.csr_graphs["Neighbours"].degree(), g["degree"])
    L = pgraph.laplacian()
    assert L.shape == (1000, 1000) and abs(L.sum()) < 1e-3
    assert np.isfinite(pgraph.dirichlet()).all() and pgraph.local_variance().shape == (1000,)
    # device-CSR consumers (pg_csr_row_stats) against the column-based host path of the same graph
    assert pgraph._device_graph("Neighbours") is not None
    dev = (pgraph.degree(), pgraph.dirichlet(), pgraph.local_variance(), pgraph.dirichlet(boolean_weights=True))
    saved, pgraph.csr_graphs = pgraph.csr_graphs, {}
    host = (pgraph.degree(), pgraph.dirichlet(), pgraph.local_variance(), pgraph.dirichlet(boolean_weights=True))
    pgraph.csr_graphs = saved
    assert np.array_equal(dev[0], host[0])
    assert dev[1].shape == (1, 1) and np.allclose(dev[1], host[1], rtol=1e-9) and np.allclose(dev[3], host[3], rtol=1e-9)
    assert np.allclose(dev[2], host[2], rtol=1e-9, atol=1e-12)
    # a kNN graph and an eps graph stored under a name use the device CSR as well
    pgraph.build_graph(k=5, store="K5")
    pgraph.build_graph(eps=2, store="E2")
    for name in ("K5", "E2"):
        d1, v1, q1 = pgraph.degree(name), pgraph.local_variance(name), pgraph.dirichlet(name)
        saved, pgraph.csr_graphs = pgraph.csr_graphs, {}
        d0, v0, q0 = pgraph.degree(name), pgraph.local_variance(name), pgraph.dirichlet(name)
        pgraph.csr_graphs = saved
        assert np.array_equal(d1, d0) and np.allclose(v1, v0, rtol=1e-9, atol=1e-12) and np.allclose(q1, q0, rtol=1e-9)
        A = pgraph.adjacency(name)
        assert A.shape == (1000, 1000) and A.nnz == len(pgraph.get_neighbour_coords(name)[0])
    # ONE row's tuple replaced in place - any row, not just the first / middle / last: the device CSR is stale
    assert pgraph._device_graph("K5") is not None
    col = pgraph.graph["K5"].values
    col[337] = (col[337][0], col[337][1] * 3)
    assert pgraph._device_graph("K5") is None
    assert pgraph.degree("K5")[337] == col[337][1].sum()       # (the host path reads the column as it is now)
    pgraph.graph["E2"] = pgraph.build_graph(eps=1)           # user overwrites the column: cache must not be used
    assert pgraph._device_graph("E2") is None and np.all(pgraph.degree("E2") == 27)
    # ... also when the new graph has the SAME structure and only other weights (the old check
    # compared neighbour counts of two rows and kept answering with the stale integer weights)
    assert np.all(pgraph.degree() == 27)
    pgraph.graph["Neighbours"] = pgraph.build_graph(eps=1, similarity=True)
    assert pgraph._device_graph("Neighbours") is None
    assert np.allclose(pgraph.degree(), 13.5) and np.all(pgraph.degree(boolean_weights=True) == 27)


def _analytics_vs_reference(pg, name, g, prefix, exact_degree):
    """degree / Laplacian diagonal / Dirichlet energy / local variance of a stored graph against the
    values the REAL reference produced for the same graph (oracle/gen_golden.py::store_analytics,
    prograph/prograph.py:797-946).  Tolerance: the reference keeps degrees and weights in float32 and
    sums them in float32 (numpy pairwise for degree(), scipy's sequential column sums for the in-degree:
    243 similarity weights of 1/3 and 1/2 drift by 1.4e-6 there), this path sums in float64 and rounds
    once - integer weights agree exactly, similarity weights 1/(1+d) within 1e-5 relative."""
    assert pg._device_graph(name) is not None
    rt = 0.0 if exact_degree else 1e-5
    for bw, tag in ((False, "w"), (True, "b")):
        deg = pg.degree(name, boolean_weights=bw)
        assert deg.dtype == np.float32
        assert np.allclose(deg, g[f"{prefix}_deg_{tag}"], rtol=0.0 if bw else rt, atol=0)
        for mode in ("outdegree", "indegree"):
            want = g[f"{prefix}_lapdiag_{tag}_{mode[:3]}"]
            assert np.allclose(pg.laplacian(name, boolean_weights=bw, mode=mode).diagonal(), want, rtol=max(rt, 1e-12))
            assert np.allclose(pg.csr_graphs[name].as_csr().laplacian_diagonal(bw, mode) if hasattr(pg.csr_graphs[name], "as_csr")
                               else pg.csr_graphs[name].laplacian_diagonal(bw, mode), want, rtol=max(rt, 1e-12))
            got = pg.dirichlet(name, boolean_weights=bw, mode=mode)
            assert got.shape == (1, 1)
            assert np.allclose(got, g[f"{prefix}_dirichlet_{tag}_{mode[:3]}"], rtol=1e-9 if (exact_degree or bw) else 1e-4)
    lv, want = pg.local_variance(name), g[f"{prefix}_locvar"]
    assert np.array_equal(np.isnan(lv), np.isnan(want)) and np.allclose(lv[~np.isnan(lv)], want[~np.isnan(want)], rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------- prograph/prograph.py:797-946
def test_graph_analytics_match_the_reference(pgraph):
    g = load_golden("ref_synthetic_csv")
    _analytics_vs_reference(pgraph, "Neighbours", g, "ana_eps1", exact_degree=True)
    pgraph.build_graph(k=5, store="K5")
    _analytics_vs_reference(pgraph, "K5", g, "ana_knn5", exact_degree=True)
    pgraph.build_graph(k=4, similarity=True, store="K4s")
    _analytics_vs_reference(pgraph, "K4s", g, "ana_knn4sim", exact_degree=False)
    pgraph.build_graph(eps=2, similarity=True, store="E2s")
    _analytics_vs_reference(pgraph, "E2s", g, "ana_eps2sim", exact_degree=False)


def test_graph_analytics_with_duplicates_match_the_reference(backend, tmp_path, capsys):
    """Exact duplicates + kNN + similarity weights: eight rows of the k=3 graph list the row itself,
    which the reference's `L.setdiag(D)` removes from the Laplacian (prograph.py:894-896)."""
    from prograph_amd import Prograph
    g = load_golden("synth_n515_l20_dups")
    f = tmp_path / "dups.csv"
    pd.DataFrame({"Sequence": synth.tokens_to_strings(g["tokens"]), "Fitness": g["fitness"]}).to_csv(f)
    pg = Prograph(file=str(f))
    capsys.readouterr()
    _analytics_vs_reference(pg, "Neighbours", g, "ana_eps1", exact_degree=True)
    neigh = pg.build_graph(k=3, similarity=True, store="K3s")
    idx = np.stack([a[0] for a in neigh])
    assert np.array_equal(idx, g["knn3_sim_idx"]) and (idx == np.arange(len(idx))[:, None]).any(1).sum() == 8
    _analytics_vs_reference(pg, "K3s", g, "ana_knn3sim", exact_degree=False)
    pg.build_graph(k=16, store="K16")
    _analytics_vs_reference(pg, "K16", g, "ana_knn16", exact_degree=True)


def test_hamming_operator_cache_does_not_outlive_its_operand(backend):
    """the operand cache was keyed on the storage address; a freed operand's
    distances were returned for the next tensor allocated in its place."""
    from prograph_amd.distance import hamming
    from prograph_amd import _native
    from oracle import prograph_oracle as O
    dev = _native.device()
    rng = np.random.RandomState(5)
    Y = torch.from_numpy(rng.randint(0, 21, size=(3, 40)).astype(np.float32)).to(torch.float16).to(dev)
    ptrs = set()
    for it in range(6):
        host = rng.randint(0, 21, size=(2048, 40)).astype(np.float32)
        X = torch.from_numpy(host).to(torch.float16).to(dev)          # same shape, usually the block just freed
        ptrs.add(X.data_ptr())
        d = hamming(X, Y)
        assert np.array_equal(d.cpu().numpy(), O.hamming(host.astype(np.int64), Y.cpu().numpy().astype(np.int64)).numpy()), it
        del X, d
    assert len(ptrs) < 6 or backend == "fake"      # the allocator did hand a block out again (what the bug needed)


# ---------------------------------------------------------------- build_graph: reference outputs, bit-exact
def _check_tuples(neigh, g, name, knn=False):
    assert isinstance(neigh, list)
    if knn:
        idx = np.stack([a[0] for a in neigh]); w = np.stack([a[1] for a in neigh])
        assert idx.dtype == np.int64 and np.array_equal(idx, g[name + "_idx"])
        gw = g[name + "_w"]
        assert np.array_equal(w, gw) if not np.issubdtype(gw.dtype, np.integer) else (w.dtype == np.int64 and np.array_equal(w, gw))
        return
    counts = np.array([len(a[0]) for a in neigh])
    assert np.array_equal(np.concatenate([[0], np.cumsum(counts)]), g[name + "_indptr"])
    idx = np.concatenate([a[0] for a in neigh]); w = np.concatenate([a[1] for a in neigh])
    assert all(a[0].dtype == np.int64 for a in neigh)
    assert np.array_equal(idx, g[name + "_indices"])
    gw = g[name + "_weights"]
    if np.issubdtype(gw.dtype, np.integer):
        assert all(a[1].dtype == np.int64 for a in neigh) and np.array_equal(w, gw)
    else:
        assert w.dtype == gw.dtype and np.array_equal(w, gw)


def test_build_graph_matches_reference(pgraph):
    g = load_golden("ref_synthetic_csv")
    _check_tuples(list(pgraph.graph["Neighbours"]), g, "eps1")
    _check_tuples(pgraph.build_graph(eps=2), g, "eps2")
    _check_tuples(pgraph.build_graph(eps=3, cap=8), g, "eps3")
    for nm, op in [("eq", operator.eq), ("lt", operator.lt), ("ge", operator.ge), ("gt", operator.gt)]:
        _check_tuples(pgraph.build_graph(eps=2, comp=op), g, "eps2_" + nm)
    _check_tuples(pgraph.build_graph(eps=1, batch_size=5), g, "eps1_b5")
    _check_tuples(pgraph.build_graph(eps=1, idxs=g["sub_idxs"]), g, "eps1_sub")
    _check_tuples(pgraph.build_graph(eps=1, similarity=True), g, "eps1_sim")
    for k in (1, 2, 16):
        _check_tuples(pgraph.build_graph(k=k), g, f"knn{k}", knn=True)
    _check_tuples(pgraph.build_graph(k=4, similarity=True), g, "knn4_sim", knn=True)
    _check_tuples(pgraph.build_graph(k=3, idxs=g["sub_idxs"]), g, "knn3_sub", knn=True)
    csr = pgraph.build_graph(eps=2, output="csr")
    assert csr.nnz == int(g["eps2_indptr"][-1]) and csr.nrows == 1000
    # idxs as a boolean mask / a slice select the same rows as the integer list
    mask = np.zeros(1000, dtype=bool); mask[g["sub_idxs"]] = True
    _check_tuples(pgraph.build_graph(eps=1, idxs=mask), g, "eps1_sub")
    _check_tuples(pgraph.build_graph(eps=1, idxs=slice(100, 400)), g, "eps1_sub")
    _check_tuples(pgraph.build_graph(k=3, idxs=list(g["sub_idxs"])), g, "knn3_sub", knn=True)


def test_build_graph_argument_errors(pgraph):
    with pytest.raises(ValueError):
        pgraph.build_graph()
    with pytest.raises(ValueError):
        pgraph.build_graph(eps=1, k=1)
    with pytest.raises(ValueError):
        pgraph.build_graph(k=0)                                    # tests/tests.py:149-151
    with pytest.raises(ValueError):
        pgraph.build_graph(eps=0)
    with pytest.raises(TypeError):
        pgraph.build_graph(k=0.5)                                  # tests/tests.py:152-154


# ---------------------------------------------------------------- tests/tests.py:173-191: the distance operator
def test_hamming_operator(backend):
    from prograph_amd.distance import hamming
    X = torch.Tensor([[1, 2, 3], [4, 5, 6]]); Y = torch.Tensor([[1, 2, 3], [7, 8, 9]])
    assert torch.all(hamming(X, Y) == torch.Tensor([[0, 3], [3, 3]]))
    assert torch.all(hamming(X, torch.Tensor([1, 2, 3])) == torch.Tensor([[0, 3]]))
    assert torch.all(hamming(torch.Tensor([4, 5, 6]), torch.Tensor([1, 2, 3])) == torch.Tensor([[3]]))
    with pytest.raises(ValueError):
        hamming(torch.Tensor([4, 5, 6]), torch.Tensor())
    out = hamming(X, Y)
    assert out.dtype == torch.int64 and out.device == X.device and tuple(out.shape) == (2, 2)
    g = load_golden("hamming_kats")
    for i in range(6):       # incl. unequal D (zero padding) and D > 128 (5-bit planes up to 255 tokens)
        got = hamming(g[f"r{i}_X"].astype(np.int64), g[f"r{i}_Y"].astype(np.int64))
        assert np.array_equal(got.numpy(), g[f"r{i}_out"]), i
    assert np.array_equal(hamming(g["wide_X"], g["wide_Y"]).numpy(), g["wide_out"])
    assert np.array_equal(hamming(g["i32_X"], g["i32_Y"]).numpy(), g["i32_out"])
    sim = hamming(g["r0_X"].astype(np.int64), g["r0_Y"].astype(np.int64), similarity=True).numpy()
    assert sim.dtype == g["sim_out"].dtype and np.array_equal(sim, g["sim_out"])
    assert np.array_equal(hamming(np.array([[1.5, 2.0]]), np.array([[1.5, 3.0]])).numpy(), [[1]])


# ---------------------------------------------------------------- the generic distance protocol (custom callables)
def test_custom_distance_callable_and_comp(pgraph):
    g = load_golden("ref_synthetic_csv")
    from prograph_amd.distance import hamming

    def my_distance(X, Y, similarity=False):
        return hamming(X, Y, similarity=similarity)
    _check_tuples(pgraph.build_graph(eps=2, distance=my_distance), g, "eps2")
    _check_tuples(pgraph.build_graph(k=2, distance=my_distance), g, "knn2", knn=True)
    _check_tuples(pgraph.build_graph(eps=2, comp=lambda a, b: a <= b), g, "eps2")
    # k > 63: native continuation rounds vs the generic operator-protocol path (stable sort)
    gen = pgraph.build_graph(k=100, distance=my_distance)
    _check_tuples(pgraph.build_graph(k=100), {"knn100_idx": np.stack([a[0] for a in gen]), "knn100_w": np.stack([a[1] for a in gen])},
                  "knn100", knn=True)


@pytest.mark.parametrize("name", ["synth_n1000_l32", "synth_n515_l20_dups", "synth_n300_varlen24"])
def test_constructor_on_synthetic_sets(backend, name, tmp_path, capsys):
    """Prograph(csv) end to end on seeded sets incl. duplicates and variable length (zero padding)."""
    from prograph_amd import Prograph
    g = load_golden(name)
    tok = g["tokens"]
    f = tmp_path / (name + ".csv")
    pd.DataFrame({"Sequence": synth.tokens_to_strings(tok), "Fitness": np.linspace(0, 1, len(tok))}).to_csv(f)
    pg = Prograph(file=str(f))
    capsys.readouterr()
    assert np.array_equal(pg.tokenized, tok)
    _check_tuples(list(pg.graph["Neighbours"]), g, "eps1")
    for key in g.files:
        if key.endswith("_indptr") and "sub" not in key and key != "eps1_indptr":
            _check_tuples(pg.build_graph(eps=int(key[3:-7])), g, key[:-7])
        if key.startswith("knn") and key.endswith("_idx") and "sub" not in key and "sim" not in key:
            _check_tuples(pg.build_graph(k=int(key[3:-4])), g, key[:-4], knn=True)
    if "sub_idxs" in g.files:
        _check_tuples(pg.build_graph(eps=2, idxs=g["sub_idxs"]), g, "eps2_sub")
        _check_tuples(pg.build_graph(k=3, idxs=g["sub_idxs"]), g, "knn3_sub", knn=True)
    d = g["dist_to_ref"][0].astype(np.int64)
    r = int(g["ref_row"])
    want = int(np.bincount(d).argmax())
    assert np.array_equal(pg.indexing(reference_seq=r, distances=want), np.where(d == want)[0])


# ---------------------------------------------------------------- tests/tests.py:139-167: kNN / eps with minkowski on an
# embedded representation — the generic distance-operator protocol (reference semantics on torch ops).  The reference's
# fixture (data/knntest_pgraph.pkl) is a pickle and is not loaded; six 2-D points with the same neighbour structure
# and the same expected answers are used instead.
EMBEDDED = np.array([[0, 0], [0.5, 0], [4, 0], [3, 0], [4.2, 1.5], [4.95, 1.75]], dtype=np.float32)


def test_sequences_beyond_the_fused_limits(backend, tmp_path, capsys):
    """More than 255 positions: the reference has no length limit (its hamming() is a broadcast), the
    fused kernels do.  Constructor, summary, indexing (distances / positions / both), calc_neighbours,
    neighbourhood and build_graph (eps and k) must still work and agree with the oracle: they run on
    the native dense operator, which sums column segments, plus torch ops."""
    from oracle import prograph_oracle as O
    from prograph_amd import Prograph
    N, L = 400, 300
    tok = synth.clustered_tokens(N, L, seed=12, members=40)
    tok[5] = tok[45]                                           # a duplicate
    f = tmp_path / "long.csv"
    pd.DataFrame({"Sequence": synth.tokens_to_strings(tok), "Fitness": np.arange(N, dtype=float)}).to_csv(f)
    pg = Prograph(file=str(f))
    out = capsys.readouterr().out
    t64 = tok.astype(np.int64)
    d0 = O.hamming(t64, t64[:1]).numpy()[0]
    assert f"Number of Sequences : {N}" in out and f"Max Distance        : {int(d0.max())}" in out
    assert f"Number of Distances : {len(np.unique(d0))}" in out and f"Longest Sequence    : {L}" in out
    # indexing: distances, positions (or / and), both, against the oracle's restatement
    dd = [int(x) for x in np.unique(d0)[1:3]]
    assert np.array_equal(pg.indexing(distances=dd), O.indexing(t64, 0, L, distances=dd))
    mutpos = [int(x) for x in np.nonzero((t64 != t64[0]).any(axis=0))[0][:3]]
    for mode in ("or", "and"):
        try:
            want = O.indexing(t64, 0, L, positions=mutpos, Bool=mode)
        except AssertionError:
            with pytest.raises(AssertionError):
                pg.indexing(positions=mutpos, Bool=mode)
            continue
        assert np.array_equal(pg.indexing(positions=mutpos, Bool=mode), want)
    with pytest.raises(AssertionError):
        pg.indexing(distances=[L + 5])
    seq7 = pg("Sequence")[7]
    d7 = O.hamming(t64, t64[7:8]).numpy()[0]
    assert np.array_equal(pg.calc_neighbours(seq7, eps=int(np.sort(d7)[3]), comp=operator.le), np.nonzero(d7 <= np.sort(d7)[3])[0])
    assert list(pg.neighbourhood(seq7, 4).index) == list(np.nonzero(d7 <= 4)[0])
    # graphs: dense kernel over column segments (fp16-coded integers) + the device selection kernels - NOT the
    # torch batch loop: every row against the oracle, every comparator, similarity, stored device graphs
    def no_generic(*a, **kw):
        raise AssertionError("long byte-token sequences must not take the generic batch loop")
    pg._build_graph_generic = no_generic
    D = O.hamming(t64, t64).numpy()
    g = pg.build_graph(k=5)
    e = pg.build_graph(eps=4)
    assert len(g) == N and len(e) == N
    for r in range(N):
        order = np.argsort(D[r], kind="stable")[1:6]
        assert np.array_equal(g[r][0], order) and np.array_equal(g[r][1], D[r][order])
        cols = np.nonzero((D[r] <= 4) & (D[r] > 0))[0]
        assert np.array_equal(e[r][0], cols) and np.array_equal(e[r][1], D[r][cols])
        assert g[r][1].dtype == np.int64 and (len(cols) == 0 or e[r][1].dtype == np.int64)
    far = int(np.sort(D[3])[N // 2])                           # a threshold among unrelated sequences (d > 255)
    assert far > 255
    for comp, eps in ((operator.lt, far), (operator.ge, far + 0.5), (operator.gt, far), (operator.eq, far), (operator.eq, 4.5),
                      (operator.le, 4.999)):
        got = pg.build_graph(eps=eps, comp=comp)
        for r in (0, 3, 5, 45, 399):
            cols = np.nonzero(comp(D[r], eps) & (D[r] > 0))[0]
            assert np.array_equal(got[r][0], cols) and np.array_equal(got[r][1], D[r][cols]), (comp, eps, r)
    s = pg.build_graph(k=3, similarity=True)
    for r in (0, 5, 45):
        order = np.argsort(D[r], kind="stable")[1:4]
        assert np.array_equal(s[r][0], order) and np.allclose(s[r][1], 1 / (1 + D[r][order]))
    sub = pg.build_graph(k=2, idxs=[0, 5, 45, 7, 300])
    Ds = D[np.ix_([0, 5, 45, 7, 300], [0, 5, 45, 7, 300])]
    for r in range(5):
        assert np.array_equal(sub[r][0], np.argsort(Ds[r], kind="stable")[1:3])
    pg.build_graph(eps=4, store="Near")                        # analytics from the device CSR (int16 weights)
    deg = pg.degree(graph="Near")
    assert np.array_equal(deg, np.array([D[r][(D[r] <= 4) & (D[r] > 0)].sum() for r in range(N)], dtype=np.float32))


@pytest.fixture
def knn_test(backend, tmp_path, capsys):
    from prograph_amd import Prograph
    f = tmp_path / "knntest.csv"
    pd.DataFrame({"Sequence": list("ACDEFG"), "Fitness": [1.70406036, 0.32788095, 0.79588575, 0.5333638, -0.18089174, 0.61660484]}).to_csv(f)
    pg = Prograph(file=str(f))
    capsys.readouterr()
    pg.graph["Embedded"] = list(EMBEDDED)
    return pg


def test_knn_graph_generation_minkowski(knn_test):
    from prograph_amd.distance import minkowski
    L = [x[0] for x in knn_test.build_graph(representation="Embedded", k=1, distance=minkowski)]
    assert np.all(np.array(L).reshape(-1,) == np.array([1, 0, 3, 2, 5, 4]))
    L = [x[0] for x in knn_test.build_graph(representation="Embedded", k=2, distance=minkowski)]
    assert np.all(L == np.array([[1, 3], [0, 3], [3, 4], [2, 4], [5, 2], [4, 2]]))
    with pytest.raises(ValueError):
        knn_test.build_graph(representation="Embedded", k=0, distance=minkowski)
    with pytest.raises(TypeError):
        knn_test.build_graph(representation="Embedded", k=0.5, distance=minkowski)
    knn_test.graph["Weighted"] = knn_test.build_graph(eps=2, representation="Embedded", distance=minkowski)
    np.testing.assert_almost_equal(knn_test.degree(graph="Weighted", boolean_weights=True), np.array([1, 1, 3, 2, 3, 2]))
    knn_test.graph["Weighted"] = knn_test.build_graph(k=1, representation="Embedded", distance=minkowski)
    # fp16 staging like the reference (prograph.py:726): sqrt(0.625) rounds to 0.79052734
    np.testing.assert_almost_equal(knn_test.degree(graph="Weighted"), np.array([0.5, 0.5, 1., 1., 0.79052734, 0.79052734]), decimal=7)


def test_minkowski_graphs_match_the_reference(backend, tmp_path, capsys):
    """`build_graph(representation="Embedded", distance=minkowski)` against outputs of the real reference
    (tests/golden/minkowski_f16.npz): the knntest embedding of the golden generator and a seeded
    (1000, 64) fp16 set - bit-exact at these dimensions, tuple formats and dtypes as the reference's."""
    from prograph_amd import Prograph
    from prograph_amd.distance import minkowski
    g = load_golden("minkowski_f16")
    f = tmp_path / "knntest.csv"
    pd.DataFrame({"Sequence": list("ACDEFG"), "Fitness": g["knntest_fitness"]}).to_csv(f)
    pg = Prograph(file=str(f))
    pg.graph["Embedded"] = list(g["knntest_emb"])
    for k in (1, 2, 5):
        L = pg.build_graph(representation="Embedded", k=k, distance=minkowski)
        assert np.array_equal(np.stack([x[0] for x in L]), g[f"knntest_knn{k}_idx"]) and L[0][0].dtype == np.int64
        assert np.array_equal(np.stack([x[1] for x in L]), g[f"knntest_knn{k}_w"]) and L[0][1].dtype == np.float16
    pg.graph["W1"] = pg.build_graph(representation="Embedded", k=1, distance=minkowski)
    assert np.array_equal(pg.degree("W1"), g["knntest_deg_w1"])               # 0.79052734 included, bit for bit
    n = 1000
    tok = synth.clustered_tokens(n, 8, seed=synth.DEFAULT_SEED + 42)
    f2 = tmp_path / "d64.csv"
    pd.DataFrame({"Sequence": synth.tokens_to_strings(tok), "Fitness": np.zeros(n)}).to_csv(f2)
    pg = Prograph(file=str(f2))
    capsys.readouterr()
    pg.graph["Embedded"] = list(g["d64_emb"])
    for k in (1, 5, 16):
        L = pg.build_graph(representation="Embedded", k=k, distance=minkowski)
        assert np.array_equal(np.stack([x[0] for x in L]), g[f"d64_knn{k}_idx"])
        assert np.array_equal(np.stack([x[1] for x in L]), g[f"d64_knn{k}_w"])
    L = pg.build_graph(representation="Embedded", k=4, similarity=True, distance=minkowski)
    assert np.array_equal(np.stack([x[0] for x in L]), g["d64_knn4_sim_idx"]) and np.array_equal(np.stack([x[1] for x in L]), g["d64_knn4_sim_w"])
    for sim, key in ((False, "d64_eps"), (True, "d64_eps_sim")):
        E = pg.build_graph(representation="Embedded", eps=float(g["d64_eps"]), similarity=sim, distance=minkowski)
        ip = np.concatenate([[0], np.cumsum([len(e[0]) for e in E])])
        assert np.array_equal(ip, g[key + "_indptr"])
        assert np.array_equal(np.concatenate([e[0] for e in E]), g[key + "_indices"])
        assert np.array_equal(np.concatenate([np.asarray(e[1], dtype=np.float64) for e in E]), g[key + "_weights"].astype(np.float64))
        empty = [e for e in E if len(e[0]) == 0]
        assert all(e[0].dtype == np.int64 or e[0].dtype == int for e in E) and all(e[1].dtype == int for e in empty)


def test_minkowski_operator(backend):
    from prograph_amd.distance import minkowski
    X = torch.Tensor([[1, 2, 3], [4, 5, 6]]); Y = torch.Tensor([[1, 2, 3], [7, 8, 9]])       # tests/tests.py:192-208
    assert torch.allclose(minkowski(X, Y), torch.Tensor([[0.0, 5.19615242], [10.3923048454, 5.19615242]]))
    assert torch.allclose(minkowski(X, torch.Tensor([1, 2, 3])), torch.Tensor([[0.0, 5.19615242]]))
    assert torch.allclose(minkowski(torch.Tensor([4, 5, 6]), torch.Tensor([1, 2, 3])), torch.Tensor([[5.19615242]]))
    with pytest.raises(ValueError):
        minkowski(torch.Tensor([4, 5, 6]), torch.Tensor())
