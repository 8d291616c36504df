"""
Pins the oracle (oracle/prograph_oracle.py) to the golden vectors generated from the real
reference (oracle/gen_golden.py) and to the literal known answers in the reference's own
tests/tests.py.  CPU only.
"""
import operator

import numpy as np
import pytest
import torch

from oracle import prograph_oracle as O
from conftest import load_golden

SETS = ["synth_n1000_l32", "synth_n2085_l64", "synth_n515_l20_dups", "synth_n300_varlen24"]


def _check_eps(g, name, neigh):
    indptr, idx, w = O.neighbours_to_csr(neigh)
    assert np.array_equal(indptr, g[name + "_indptr"])
    assert np.array_equal(idx, g[name + "_indices"].astype(np.int64))
    gw = g[name + "_weights"]
    if np.issubdtype(gw.dtype, np.integer):
        assert np.array_equal(w.astype(np.int64), gw.astype(np.int64))
    else:
        assert w.dtype == gw.dtype and np.array_equal(w, gw)
    for a in neigh:
        assert a[0].dtype == np.int64


def _check_knn(g, name, neigh):
    idx, w = O.neighbours_to_knn(neigh)
    assert np.array_equal(idx, g[name + "_idx"].astype(np.int64))
    gw = g[name + "_w"]
    if np.issubdtype(gw.dtype, np.integer):
        assert np.array_equal(w.astype(np.int64), gw.astype(np.int64))
    else:
        assert np.array_equal(w, gw)


def test_reference_known_answers_hamming():
    # tests/tests.py:175-191 of the reference
    X = torch.Tensor([[1, 2, 3], [4, 5, 6]]); Y = torch.Tensor([[1, 2, 3], [7, 8, 9]])
    assert torch.all(O.hamming(X, Y) == torch.Tensor([[0, 3], [3, 3]]))
    assert torch.all(O.hamming(X, torch.Tensor([1, 2, 3])) == torch.Tensor([[0, 3]]))
    assert torch.all(O.hamming(torch.Tensor([4, 5, 6]), torch.Tensor([1, 2, 3])) == torch.Tensor([[3]]))
    with pytest.raises(ValueError):
        O.hamming(torch.Tensor([4, 5, 6]), torch.Tensor())
    assert O.hamming(X, Y).dtype == torch.int64


def test_hamming_kats():
    g = load_golden("hamming_kats")
    for i in range(6):
        out = O.hamming(g[f"r{i}_X"].astype(np.int64), g[f"r{i}_Y"].astype(np.int64)).numpy()
        assert np.array_equal(out, g[f"r{i}_out"])
    assert np.array_equal(O.hamming(g["wide_X"].astype(np.int64), g["wide_Y"].astype(np.int64)).numpy(), g["wide_out"])
    assert np.array_equal(O.hamming(g["i32_X"], g["i32_Y"]).numpy(), g["i32_out"])
    sim = O.hamming(g["r0_X"].astype(np.int64), g["r0_Y"].astype(np.int64), similarity=True).numpy()
    assert sim.dtype == g["sim_out"].dtype and np.array_equal(sim, g["sim_out"])


def test_tokenize_known_answers():
    # tests/tests.py:124-133
    assert np.all(O.tokenize("ACA") == np.array([1, 2, 1]))
    assert np.all(O.tokenize(["ACA", "ACC"]) == np.array([[1, 2, 1], [1, 2, 2]]))
    t = O.tokenize(["ACCCACAAA", "ACAA"])
    assert np.all(t == np.array([[1, 2, 2, 2, 1, 2, 1, 1, 1], [1, 2, 1, 1, 0, 0, 0, 0, 0]]))
    assert len(O.tokenize([])) == 0


def test_reference_csv_graphs():
    g = load_golden("ref_synthetic_csv")
    tok = g["tokens"].astype(np.int64)
    _check_eps(g, "eps1", O.build_graph(tok, eps=1))
    _check_eps(g, "eps2", O.build_graph(tok, eps=2))
    _check_eps(g, "eps3", O.build_graph(tok, eps=3))
    for nm, op in [("eq", operator.eq), ("lt", operator.lt), ("ge", operator.ge), ("gt", operator.gt)]:
        _check_eps(g, "eps2_" + nm, O.build_graph(tok, eps=2, comp=op))
    _check_eps(g, "eps1_b5", O.build_graph(tok, eps=1, batch_size=5))
    _check_eps(g, "eps1_sub", O.build_graph(tok, eps=1, idxs=g["sub_idxs"]))
    _check_eps(g, "eps1_sim", O.build_graph(tok, eps=1, similarity=True))
    for k in (1, 2, 16):
        _check_knn(g, f"knn{k}", O.build_graph(tok, k=k))
    _check_knn(g, "knn4_sim", O.build_graph(tok, k=4, similarity=True))
    _check_knn(g, "knn3_sub", O.build_graph(tok, k=3, idxs=g["sub_idxs"]))
    # tests/tests.py:157-158 and :135-137
    neigh = O.build_graph(tok, eps=1)
    assert np.all(O.degree(neigh) == 27) and np.array_equal(O.degree(neigh), g["degree"])
    I, J, V = O.neighbour_coords(neigh)
    dense = np.zeros((3, 3))
    for i, j, v in zip(I, J, V):
        if i < 3 and j < 3:
            dense[i, j] = v
    assert np.array_equal(dense, g["adj33"]) and np.array_equal(dense, [[0, 1, 1], [1, 0, 1], [1, 1, 0]])


def test_reference_csv_indexing():
    g = load_golden("ref_synthetic_csv")
    tok = g["tokens"].astype(np.int64)
    ix = lambda **kw: O.indexing(tok, kw.pop("ref", 0), 3, **kw)
    assert np.array_equal(ix(positions=[1, 2]), g["ix_pos12"]) and len(g["ix_pos12"]) == 99
    assert np.array_equal(ix(positions=[1, 2], Bool="and"), g["ix_pos12_and"])
    assert np.array_equal(ix(distances=3), g["ix_d3"]) and len(g["ix_d3"]) == 729
    assert np.array_equal(ix(distances=2), g["ix_d2"]) and len(g["ix_d2"]) == 243
    assert np.array_equal(ix(distances=[1, 3]), g["ix_d13"]) and len(g["ix_d13"]) == 756
    assert np.array_equal(ix(positions=[1, 2], distances=2), g["ix_pos12_d2"]) and len(g["ix_pos12_d2"]) == 81
    a, b = ix(positions=[1, 2], distances=2, complement=True)
    assert np.array_equal(a, g["ix_pos12_d2_c0"]) and np.array_equal(b, g["ix_pos12_d2_c1"]) and b[12] == 30
    assert np.array_equal(ix(ref=int(g["LDC_idx"]), positions=[1]), g["ix_LDC_pos1"])
    assert np.array_equal(ix(ref=int(g["LDC_idx"]), distances=1), g["ix_LDC_d1"])
    with pytest.raises(AssertionError):
        ix(distances=[1, 2, 4])
    assert len(ix(percentage=0.7)) == 700
    assert len(ix(positions=[1, 2], distances=2, percentage=0.3)) == 24
    assert np.array_equal(O.calc_neighbours(tok, int(g["ACL_idx"])), g["calc_neigh_ACL"])
    assert np.array_equal(O.calc_neighbours(tok, int(g["ACL_idx"]), eps=2, comp=operator.le), g["calc_neigh_ACL_le2"])
    # tests/tests.py:63-64: calc_neighbours("ACL") == stored Neighbours of "ACL"
    r = int(g["ACL_idx"])
    assert np.array_equal(g["calc_neigh_ACL"], g["eps1_indices"][g["eps1_indptr"][r]:g["eps1_indptr"][r + 1]])
    assert O.summary_numbers(tok, 0) == (int(g["str_maxdist"]), int(g["str_ndist"])) == (3, 4)


@pytest.mark.parametrize("name", SETS)
def test_synthetic_sets(name):
    g = load_golden(name)
    tok = g["tokens"].astype(np.int64)
    for key in g.files:
        if key.endswith("_indptr") and not key.endswith("sub_indptr"):
            e = int(key[3:-7])
            _check_eps(g, f"eps{e}", O.build_graph(tok, eps=e))
        if key.endswith("_idx") and key.startswith("knn") and "sub" not in key and "sim" not in key:
            k = int(key[3:-4])
            _check_knn(g, f"knn{k}", O.build_graph(tok, k=k))
    if "knn3_sim_idx" in g.files:
        _check_knn(g, "knn3_sim", O.build_graph(tok, k=3, similarity=True))
    if "sub_idxs" in g.files:
        _check_eps(g, "eps2_sub", O.build_graph(tok, eps=2, idxs=g["sub_idxs"]))
        _check_knn(g, "knn3_sub", O.build_graph(tok, k=3, idxs=g["sub_idxs"]))
    d = O.hamming(tok, tok[int(g["ref_row"])].reshape(1, -1)).numpy()
    assert np.array_equal(d, g["dist_to_ref"].astype(np.int64))


def test_build_graph_argument_errors():
    tok = load_golden("ref_synthetic_csv")["tokens"].astype(np.int64)
    with pytest.raises(ValueError):
        O.build_graph(tok)                       # neither
    with pytest.raises(ValueError):
        O.build_graph(tok, eps=1, k=1)           # both
    with pytest.raises(ValueError):
        O.build_graph(tok, k=0)                  # tests/tests.py:149-151
    with pytest.raises(TypeError):
        O.build_graph(tok, k=0.5)                # tests/tests.py:152-154


def test_levenshtein_definition():
    rng = np.random.RandomState(3)
    for _ in range(200):
        la, lb = rng.randint(1, 14, size=2)
        a = rng.randint(1, 4, size=la); b = rng.randint(1, 4, size=lb)
        full = O.levenshtein_full(a, la, b, lb)
        for band in (1, 3, 8):
            got = O.levenshtein_banded(a, la, b, lb, band)
            assert got == min(full, band + 1) if full <= band else got == band + 1


def test_fast_c_leg_matches_the_scalar_leg():
    """oracle.c's vectorised leg (32 bytes per compare; used for the full 200 000 x 200 000 checks on the GPU box) is
    pinned to its scalar leg - itself pinned to the Python oracle and so to the reference's golden vectors - on ragged
    lengths, duplicates, every comparator, row windows, and on the golden sets directly."""
    from oracle import c_oracle as C
    rng = np.random.RandomState(3)
    for l in [1, 7, 8, 9, 31, 32, 33, 40, 63, 64, 65, 100, 128, 255]:
        tok = rng.randint(0, 5, size=(257, l)).astype(np.uint8)
        tok[5] = tok[200]
        tok[17, : l // 2] = tok[3, : l // 2]
        for cmp in range(5):
            a, b = C.eps_csr(tok, cmp, 2.5, row0=3, nrows=200), C.eps_csr(tok, cmp, 2.5, row0=3, nrows=200, fast=True)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (l, cmp)
        for k in (1, 4, 19, 300):
            a, b = C.knn(tok, k), C.knn(tok, k, fast=True)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (l, k)
    for name in SETS:
        g = load_golden(name)
        ix, d = C.knn(g["tokens"], 16, fast=True) if "knn16_idx" in g.files else (None, None)
        if ix is not None:
            assert np.array_equal(ix, g["knn16_idx"]) and np.array_equal(d, g["knn16_w"])
        ip, ii, w = C.eps_csr(g["tokens"], 0, 2, fast=True)
        assert np.array_equal(ip, g["eps2_indptr"]) and np.array_equal(ii, g["eps2_indices"]) and np.array_equal(w, g["eps2_weights"])


def test_c_oracle_matches_python_oracle():
    """The C leg (oracle/oracle.c) is pinned to the Python oracle, which is pinned to the reference."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    for name in SETS:
        g = load_golden(name)
        tok = g["tokens"]
        for key in g.files:
            if key.endswith("_indptr") and "sub" not in key:
                e = int(key[3:-7])
                ip, ix, w = C.eps_csr(tok, 0, e)
                assert np.array_equal(ip, g[f"eps{e}_indptr"]) and np.array_equal(ix, g[f"eps{e}_indices"])
                assert np.array_equal(w, g[f"eps{e}_weights"])
            if key.startswith("knn") and key.endswith("_idx") and "sub" not in key and "sim" not in key:
                k = int(key[3:-4])
                ix, d = C.knn(tok, k)
                assert np.array_equal(ix, g[f"knn{k}_idx"]) and np.array_equal(d, g[f"knn{k}_w"])
    g = load_golden("ref_synthetic_csv")
    for nm, code in [("eq", 2), ("lt", 1), ("ge", 3), ("gt", 4)]:
        ip, ix, w = C.eps_csr(g["tokens"], code, 2)
        assert np.array_equal(ip, g[f"eps2_{nm}_indptr"]) and np.array_equal(ix, g[f"eps2_{nm}_indices"])
    k = load_golden("hamming_kats")
    assert np.array_equal(C.hamming(k["r0_X"], k["r0_Y"]), k["r0_out"])
    assert np.array_equal(C.hamming(k["wide_X"], k["wide_Y"]), k["wide_out"])
    # generator: Python and C restatements agree bit for bit
    for (n, l, seed, mem) in [(1000, 32, 20260104, 256), (777, 20, 5, 64), (300, 64, 99, 256)]:
        assert np.array_equal(C.synth(n, l, seed, mem), synth.clustered_tokens(n, l, seed=seed, members=mem))
    ip, ix, w = C.eps_csr(g["tokens"], 0, 1, row0=100, nrows=50)
    assert np.array_equal(ip, g["eps1_indptr"][100:151] - g["eps1_indptr"][100])


def test_minkowski_oracle_is_pinned_to_the_reference():
    """the fp16 Minkowski restatement against outputs of the real reference
    (oracle/gen_golden.py::gen_minkowski), including the known answers of tests/tests.py:139-167."""
    g = load_golden("minkowski_f16")
    emb = g["knntest_emb"]
    L1 = O.build_graph(emb, k=1, distance=O.minkowski)
    assert np.array_equal(np.array([x[0] for x in L1]).reshape(-1), [1, 0, 3, 2, 5, 4])                       # tests.py:141-144
    L2 = O.build_graph(emb, k=2, distance=O.minkowski)
    assert np.array_equal(np.array([x[0] for x in L2]), [[1, 3], [0, 3], [3, 4], [2, 4], [5, 2], [4, 2]])     # :145-148
    assert np.allclose(O.degree(L1), [0.5, 0.5, 1., 1., 0.79052734, 0.79052734], atol=1e-7)                   # :164-167
    E2 = O.build_graph(emb, eps=2, distance=O.minkowski)
    assert np.array_equal(O.degree(E2, boolean_weights=True), [1, 1, 3, 2, 3, 2])                             # :159-162
    for name in ("d2", "d64", "d1280"):
        e = g[f"{name}_emb"]
        d = O.minkowski(torch.as_tensor(e), torch.as_tensor(e[:64]))
        assert d.dtype == torch.float16 and np.array_equal(d.numpy(), g[f"{name}_dist64"])
        for k in (1, 5, 16):
            idx, w = O.neighbours_to_knn(O.build_graph(e, k=k, distance=O.minkowski))
            assert np.array_equal(idx, g[f"{name}_knn{k}_idx"]) and np.array_equal(w, g[f"{name}_knn{k}_w"])
        ip, ix, w = O.neighbours_to_csr(O.build_graph(e, eps=float(g[f"{name}_eps"]), distance=O.minkowski))
        assert np.array_equal(ip, g[f"{name}_eps_indptr"]) and np.array_equal(ix, g[f"{name}_eps_indices"])
        assert np.array_equal(np.asarray(w, dtype=np.float64), g[f"{name}_eps_weights"].astype(np.float64))
