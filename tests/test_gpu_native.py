"""
GPU parity tests through the C ABI (prograph_amd/_native.py -> libprograph_hip.so):
HIP kernels vs (a) golden vectors generated from the real reference, (b) the oracle on
seeded inputs, (c) size-independent properties at BASELINE.json's full sizes.
Bit-exact: everything on this path is integer work.
"""
import operator

import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

SETS = ["ref_synthetic_csv", "synth_n1000_l32", "synth_n2085_l64", "synth_n515_l20_dups", "synth_n300_varlen24"]
BITS = [5, 8]


def pytest_generate_tests(metafunc):
    """Tests that reach an all-pairs engine run on both of them: pg_nsq.h (stage 1 on the VALU) and pg_mm.h
    (stage 1 on the matrix cores) - left alone the library picks by size (MFMA from 20 000 rows for kNN, 40 000 / 60 000 for eps) and, for large launches, by a probe of the data.
    Tests marked `one_engine` (no all-pairs launch, or they choose the engine themselves) run once."""
    if "engine" in metafunc.fixturenames:
        once = metafunc.definition.get_closest_marker("one_engine") is not None
        metafunc.parametrize("engine", ["auto"] if once else ["valu", "mfma"], indirect=True)


@pytest.fixture(autouse=True)
def engine(request, monkeypatch):
    if request.param == "auto":
        monkeypatch.delenv("PG_ENGINE", raising=False)
    else:
        monkeypatch.setenv("PG_ENGINE", request.param)
    return request.param


one_engine = pytest.mark.one_engine


@pytest.fixture(scope="module")
def nat():
    from prograph_amd import _native
    _native.lib()
    _native.device()
    return _native


def _planes(nat, tok, bits, rows=None):
    return nat.pack(torch.from_numpy(np.ascontiguousarray(tok)), rows=rows, bits=bits)


def _csr_np(t):
    return [x.cpu().numpy() for x in t]


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("name", SETS)
def test_eps_golden(nat, name, bits):
    g = load_golden(name)
    tok = g["tokens"]
    p = _planes(nat, tok, bits)
    for key in g.files:
        if not key.endswith("_indptr") or "sub" in key or "sim" in key or "_b5" in key:
            continue
        base = key[:-7]
        parts = base.split("_")
        eps = int(parts[0][3:])
        cmp = {"eq": nat.CMP_EQ, "lt": nat.CMP_LT, "ge": nat.CMP_GE, "gt": nat.CMP_GT}[parts[1]] if len(parts) > 1 else nat.CMP_LE
        for cap in (256, 4):          # cap=4 forces the overflow/recompute path on almost every row
            indptr, idx, w = _csr_np(nat.eps_graph(p, p, cmp, eps, cap=cap))
            assert np.array_equal(indptr, g[base + "_indptr"]), (name, base, cap)
            assert np.array_equal(idx, g[base + "_indices"]), (name, base, cap)
            assert np.array_equal(w, g[base + "_weights"]), (name, base, cap)


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("name", SETS)
def test_knn_golden(nat, name, bits):
    g = load_golden(name)
    tok = g["tokens"]
    p = _planes(nat, tok, bits)
    for key in g.files:
        if not (key.startswith("knn") and key.endswith("_idx")) or "sub" in key or "sim" in key:
            continue
        k = int(key[3:-4])
        idx, dist = nat.knn_graph(p, p, k)
        assert np.array_equal(idx.cpu().numpy(), g[f"knn{k}_idx"]), (name, k)
        assert np.array_equal(dist.cpu().numpy(), g[f"knn{k}_w"]), (name, k)


def test_subgraph_rows_are_subset_relative(nat):
    g = load_golden("synth_n515_l20_dups")
    sub = g["sub_idxs"]
    p = _planes(nat, g["tokens"], 5, rows=sub)
    indptr, idx, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 2))
    assert np.array_equal(indptr, g["eps2_sub_indptr"]) and np.array_equal(idx, g["eps2_sub_indices"])
    assert np.array_equal(w, g["eps2_sub_weights"])
    kidx, kd = nat.knn_graph(p, p, 3)
    assert np.array_equal(kidx.cpu().numpy(), g["knn3_sub_idx"]) and np.array_equal(kd.cpu().numpy(), g["knn3_sub_w"])


def test_row_window_equals_full(nat):
    """Row-block sharding: rows [r0, r0+nr) computed alone equal that slice of the full result."""
    g = load_golden("synth_n2085_l64")
    p = _planes(nat, g["tokens"], 5)
    indptr, idx, w = g["eps2_indptr"], g["eps2_indices"], g["eps2_weights"]
    for r0, nr in [(0, 1), (1000, 77), (2000, 85), (261, 1042)]:
        ip, ix, ww = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 2, row0=r0, nrows=nr))
        assert np.array_equal(ip, indptr[r0:r0 + nr + 1] - indptr[r0])
        assert np.array_equal(ix, idx[indptr[r0]:indptr[r0 + nr]])
        assert np.array_equal(ww, w[indptr[r0]:indptr[r0 + nr]])
        kidx, kd = nat.knn_graph(p, p, 16, row0=r0, nrows=nr)
        assert np.array_equal(kidx.cpu().numpy(), g["knn16_idx"][r0:r0 + nr])
        assert np.array_equal(kd.cpu().numpy(), g["knn16_w"][r0:r0 + nr])


@one_engine
def test_dense_kats(nat):
    g = load_golden("hamming_kats")
    for i in range(6):       # r4/r5 are 130 / 200 tokens wide: native with 5 bit planes only
        X, Y = g[f"r{i}_X"], g[f"r{i}_Y"]
        D = max(X.shape[1], Y.shape[1])
        Xp = np.zeros((X.shape[0], D), np.uint8); Xp[:, :X.shape[1]] = X
        Yp = np.zeros((Y.shape[0], D), np.uint8); Yp[:, :Y.shape[1]] = Y
        for bits in (BITS if D <= 128 else [5]):
            out = nat.hamming_dense(_planes(nat, Xp, bits), _planes(nat, Yp, bits))
            assert out.dtype == torch.int64 and np.array_equal(out.cpu().numpy(), g[f"r{i}_out"])
    X, Y = g["wide_X"], g["wide_Y"]
    xp, yp = nat.pack(torch.from_numpy(X)), nat.pack(torch.from_numpy(Y), bits=8)
    assert xp.bits == 8
    assert np.array_equal(nat.hamming_dense(xp, yp).cpu().numpy(), g["wide_out"])
    for ob, dt in [(1, torch.uint8), (4, torch.int32)]:
        out = nat.hamming_dense(xp, yp, out_bytes=ob)
        assert out.dtype == dt and np.array_equal(out.cpu().numpy().astype(np.int64), g["wide_out"])


@one_engine
def test_dense_vs_oracle_all_q(nat):
    from oracle import prograph_oracle as O
    rng = np.random.RandomState(5)
    for L in [1, 3, 16, 17, 31, 33, 48, 64, 65, 80, 96, 100, 112, 127, 128, 129, 160, 161, 200, 224, 254, 255]:
        X = rng.randint(0, 21, size=(300, L)).astype(np.uint8)
        Y = X[rng.randint(0, 300, size=70)].copy()
        Y[:, rng.randint(0, L)] = 0
        ref = O.hamming(X.astype(np.int64), Y.astype(np.int64)).numpy()
        for bits in (BITS if L <= 128 else [5]):
            out = nat.hamming_dense(_planes(nat, X, bits), _planes(nat, Y, bits))
            assert np.array_equal(out.cpu().numpy(), ref), (L, bits)
            # every output type; the fp16 form (the selection kernels' operand) with in-place accumulation as well
            for nbytes in (1, 2, 4):
                o = nat.hamming_dense(_planes(nat, X, bits), _planes(nat, Y, bits), out_bytes=nbytes)
                assert np.array_equal(o.cpu().numpy().astype(np.int64), ref), (L, bits, nbytes)
            acc = nat.hamming_dense(_planes(nat, X, bits), _planes(nat, Y, bits), out_bytes=2)
            for _ in range(7):                                 # 8 x 255 = 2040: still exact in fp16
                acc = nat.hamming_dense(_planes(nat, X, bits), _planes(nat, Y, bits), out=acc)
            assert np.array_equal(acc.cpu().numpy().astype(np.int64), 8 * ref), (L, bits)


def test_engine_vs_oracle_all_q(nat, engine, monkeypatch):
    """eps + kNN engines against the oracle for every group count G=1..8 (L up to 255); on the MFMA engine the kNN
    instance with two row blocks per pass (64 rows; chosen by itself only from ~157 000 rows on) is forced as well."""
    from oracle import prograph_oracle as O
    from prograph_amd import synth
    for L in [5, 16, 24, 40, 50, 64, 70, 90, 100, 128, 129, 150, 192, 200, 230, 255]:
        tok = synth.clustered_tokens(700, L, seed=100 + L, members=100)
        tok[13] = tok[400]
        ref_e = O.neighbours_to_csr(O.build_graph(tok.astype(np.int64), eps=3))
        ref_k = O.neighbours_to_knn(O.build_graph(tok.astype(np.int64), k=7))
        for bits in (BITS if L <= 128 else [5]):
            p = _planes(nat, tok, bits)
            ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 3))
            assert np.array_equal(ip, ref_e[0]) and np.array_equal(ix, ref_e[1]) and np.array_equal(w, ref_e[2]), (L, bits)
            kidx, kd = nat.knn_graph(p, p, 7)
            assert np.array_equal(kidx.cpu().numpy(), ref_k[0]) and np.array_equal(kd.cpu().numpy(), ref_k[1]), (L, bits)
            if engine == "mfma":
                monkeypatch.setenv("PG_MM_R", "2")
                kidx, kd = nat.knn_graph(p, p, 7)
                monkeypatch.delenv("PG_MM_R")
                assert np.array_equal(kidx.cpu().numpy(), ref_k[0]) and np.array_equal(kd.cpu().numpy(), ref_k[1]), (L, bits, "R=2")


def test_knn_edge_cases(nat):
    from oracle import prograph_oracle as O
    rng = np.random.RandomState(9)
    # N smaller than k+1: missing ranks are -1 / 255; identical rows; k at the maximum
    tok = rng.randint(1, 21, size=(5, 12)).astype(np.uint8)
    p = _planes(nat, tok, 8)
    idx, d = nat.knn_graph(p, p, 8)
    idx, d = idx.cpu().numpy(), d.cpu().numpy()
    ref = O.neighbours_to_knn(O.build_graph(tok.astype(np.int64), k=8))
    assert np.array_equal(idx[:, :4], ref[0]) and np.array_equal(d[:, :4], ref[1])
    assert np.all(idx[:, 4:] == -1) and np.all(d[:, 4:] == 255)
    tok = np.repeat(rng.randint(1, 21, size=(1, 20)), 300, axis=0).astype(np.uint8)
    p = _planes(nat, tok, 5)
    idx, d = nat.knn_graph(p, p, 63)
    ref = O.neighbours_to_knn(O.build_graph(tok.astype(np.int64), k=63))
    assert np.array_equal(idx.cpu().numpy(), ref[0]) and np.all(d.cpu().numpy() == 0)
    ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 5))
    assert ip[-1] == 0 and len(ix) == 0          # duplicates are d == 0: excluded (prograph.py:736)


def test_eps_float_thresholds_and_comparators(nat):
    from oracle import prograph_oracle as O
    g = load_golden("synth_n515_l20_dups")
    tok = g["tokens"]
    p = _planes(nat, tok, 5)
    for op, code in [(operator.le, nat.CMP_LE), (operator.lt, nat.CMP_LT), (operator.eq, nat.CMP_EQ),
                     (operator.ge, nat.CMP_GE), (operator.gt, nat.CMP_GT)]:
        for eps in (1, 2.5, 3, 19, 20, 21, 300):
            ref = O.neighbours_to_csr(O.build_graph(tok.astype(np.int64), eps=eps, comp=op))
            ip, ix, w = _csr_np(nat.eps_graph(p, p, code, eps, cap=64))
            assert np.array_equal(ip, ref[0]) and np.array_equal(ix, ref[1]) and np.array_equal(w, ref[2].astype(np.uint8)), (op, eps)


@one_engine
def test_index_flags_and_compaction(nat):
    g = load_golden("ref_synthetic_csv")
    tok = g["tokens"]
    p = _planes(nat, tok, 5)
    dist, hist, _ = nat.index_flags(p, 0, want_flags=False)
    assert np.array_equal(dist.cpu().numpy(), g["dist_to_seed"][0])
    assert np.array_equal(hist.cpu().numpy(), np.bincount(g["dist_to_seed"][0], minlength=256))
    _, _, fl = nat.index_flags(p, 0, want=[3], want_dist_out=False, want_hist=False)
    assert np.array_equal(nat.compact_flags(fl).cpu().numpy(), g["ix_d3"])
    _, _, fl = nat.index_flags(p, 0, want=[1, 3], want_dist_out=False, want_hist=False)
    assert np.array_equal(nat.compact_flags(fl).cpu().numpy(), g["ix_d13"])
    pm, nm = [1, 2], [0]
    for mode, key in [(1, "ix_pos12"), (2, "ix_pos12_and")]:
        _, _, fl = nat.index_flags(p, 0, pos_mode=mode, pos_mask=pm, not_mask=nm, want_dist_out=False, want_hist=False)
        assert np.array_equal(nat.compact_flags(fl).cpu().numpy(), g[key])
    _, _, fl = nat.index_flags(p, 0, want=[2], pos_mode=1, pos_mask=pm, not_mask=nm, want_dist_out=False, want_hist=False)
    assert np.array_equal(nat.compact_flags(fl).cpu().numpy(), g["ix_pos12_d2"])
    pm, nm = [1], [0, 2]
    _, _, fl = nat.index_flags(p, int(g["LDC_idx"]), pos_mode=1, pos_mask=pm, not_mask=nm, want_dist_out=False, want_hist=False)
    assert np.array_equal(nat.compact_flags(fl).cpu().numpy(), g["ix_LDC_pos1"])
    # compaction at a size that spans many scan tiles
    rng = np.random.RandomState(1)
    f = (rng.rand(1_000_003) < 0.3).astype(np.uint8)
    got = nat.compact_flags(torch.from_numpy(f).cuda()).cpu().numpy()
    assert np.array_equal(got, np.nonzero(f)[0])


@one_engine
def test_device_tokenisation_matches_the_reference(nat):
    """pg_pack_bytes: letter table + bit slicing in one kernel from the fixed-width byte view.
    The token matrix equals the oracle's tokenize (reference :454-474) on the reference's own known answers
    (tests/tests.py:124-133), on unknown letters / empty / ragged strings and on the golden data set; the planes it
    writes equal those packed from the token matrix (same kNN graph, same dense distances)."""
    from oracle import prograph_oracle as O
    aa = O.AMINO_ACIDS
    table = np.zeros(256, dtype=np.uint8)
    for j, ch in enumerate(aa, start=1):
        table[ord(ch)] = j
    def view(seqs):
        arr = np.ascontiguousarray(np.array(seqs, dtype="bytes").reshape(-1))
        return arr.view(np.uint8).reshape(len(arr), max(arr.dtype.itemsize, 1)) if arr.dtype.itemsize else np.zeros((len(arr), 1), np.uint8)
    kats = [(["ACA"], [[1, 2, 1]]), (["ACA", "ACC"], [[1, 2, 1], [1, 2, 2]]),
            (["ACCCACAAA", "ACAA"], [[1, 2, 2, 2, 1, 2, 1, 1, 1], [1, 2, 1, 1, 0, 0, 0, 0, 0]])]
    for seqs, want in kats:
        _, tok = nat.pack_bytes(view(seqs), table)
        assert np.array_equal(tok.cpu().numpy(), np.array(want)) and np.array_equal(np.array(want), O.tokenize(seqs))
    odd = ["ACDEFGHIKLMNPQRSTVWY", "XYZ-", "", "WWWWWWWWWWWWWWWWWWWWWWWW", "acd", "A" * 255]
    _, tok = nat.pack_bytes(view(odd), table)
    assert np.array_equal(tok.cpu().numpy(), O.tokenize(odd))
    rng = np.random.RandomState(2)
    letters = np.array(list(aa + "XB-"))
    seqs = ["".join(rng.choice(letters, size=rng.randint(1, 70))) for _ in range(3000)]
    p_dev, tok = nat.pack_bytes(view(seqs), table)
    ref = O.tokenize(seqs)
    assert np.array_equal(tok.cpu().numpy(), ref)
    p_host = _planes(nat, ref.astype(np.uint8), 5)
    assert torch.equal(p_dev.buf, p_host.buf)                       # chunk arrays, signature and fold sections alike
    for bits in BITS:
        pb, tk = nat.pack_bytes(view(seqs[:500]), table, bits=bits)
        assert torch.equal(pb.buf, _planes(nat, tk.cpu().numpy(), bits).buf)
    with pytest.raises(ValueError):
        nat.pack_bytes(view(["A" * 256]), table)                    # beyond the 5-plane width
    big = table.copy(); big[ord("A")] = 77
    with pytest.raises(ValueError):
        nat.pack_bytes(view(["ACA"]), big, bits=5)                  # a table entry that does not fit the planes


@one_engine
def test_pack_flags_and_errors(nat):
    tok = np.array([[1, 2, 200], [3, 4, 5]], dtype=np.int64)
    assert nat.pack(torch.from_numpy(tok)).bits == 8 and nat.pack(torch.from_numpy(tok[1:])).bits == 5
    with pytest.raises(ValueError):
        nat.pack(torch.from_numpy(tok), bits=5)
    with pytest.raises(ValueError):
        nat.pack(torch.from_numpy(np.array([[1, 2, 300]], dtype=np.int64)))
    with pytest.raises(ValueError):
        nat.pack(torch.from_numpy(np.array([[1, -2, 3]], dtype=np.int64)))
    assert nat.pack(torch.zeros((4, 129), dtype=torch.uint8)).g == 5            # 5 bit planes: up to 255 tokens
    with pytest.raises(ValueError):
        nat.pack(torch.zeros((4, 256), dtype=torch.uint8))
    with pytest.raises(ValueError):
        nat.pack(torch.full((4, 129), 200, dtype=torch.uint8))                   # byte alphabets: up to 128
    p = nat.pack(torch.ones((10, 8), dtype=torch.uint8))
    with pytest.raises(RuntimeError):
        nat.knn_graph(p, p, 0)
    assert tuple(nat.knn_graph(p, p, 64)[0].shape) == (10, 64)          # k > 63: continuation rounds


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_full_size_properties(nat, cfg):
    """
    BASELINE.json configs[1] / configs[2] at full size, checked through properties that do not
    need an N^2 oracle: (i) CSR symmetry with equal weights (Hamming is symmetric), (ii) every
    row's columns strictly ascending, no self/duplicate (d >= 1), (iii) kNN rows are sorted by
    (d, idx), (iv) a random sample of rows equals the oracle's 1xN computation exactly,
    (v) eps-degree equals the count of kNN-consistent distances on sampled rows.
    """
    from oracle import prograph_oracle as O
    from prograph_amd import synth
    N, L, eps, k = (50_000, 32, 2, 16) if cfg == "cfg2" else (200_000, 64, 2, 16)
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok))
    assert p.bits == 5
    indptr, idx, w = nat.eps_graph(p, p, nat.CMP_LE, eps, cap=256)
    kidx, kd = nat.knn_graph(p, p, k)
    torch.cuda.synchronize()
    indptr, idx, w = indptr.cpu().numpy(), idx.cpu().numpy().astype(np.int64), w.cpu().numpy()
    kidx, kd = kidx.cpu().numpy().astype(np.int64), kd.cpu().numpy()
    rows = np.repeat(np.arange(N), np.diff(indptr))
    assert w.min() >= 1 and w.max() <= eps
    # ascending columns inside each row
    same = rows[1:] == rows[:-1]
    assert np.all(idx[1:][same] > idx[:-1][same])
    # symmetry: the multiset of (row, col, w) equals that of (col, row, w)
    a = np.sort(rows * N + idx); b = np.sort(idx * N + rows)
    assert np.array_equal(a, b)
    key_f = (rows * N + idx) * 256 + w; key_b = (idx * N + rows) * 256 + w
    assert np.array_equal(np.sort(key_f), np.sort(key_b))
    # kNN rows sorted by (d, idx)
    kk = kd.astype(np.int64) * (1 << 24) + kidx
    assert np.all(kk[:, 1:] > kk[:, :-1]) and kidx.min() >= 0 and kidx.max() < N
    # sampled rows against the oracle (1 x N per row: cheap)
    rs = np.random.RandomState(0).choice(N, size=24, replace=False)
    t64 = tok.astype(np.int64)
    for r in rs:
        d = O.hamming(t64, t64[r].reshape(1, -1)).numpy()[0]
        cols = np.where((d <= eps) & (d > 0))[0]
        assert np.array_equal(idx[indptr[r]:indptr[r + 1]], cols)
        assert np.array_equal(w[indptr[r]:indptr[r + 1]], d[cols])
        order = np.argsort(d, kind="stable")[1:k + 1]
        assert np.array_equal(kidx[r], order) and np.array_equal(kd[r], d[order])


@one_engine
@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_full_matrix_parity_at_the_headline_sizes(nat, cfg):
    """EVERY entry against the oracle, once, at BASELINE.json configs[1] / configs[2] with the library's own choice of
    engine and path (cfg3: persistent-wave MFMA engine, 64-row passes, symmetric eps): all N x 16 kNN indices and
    distances and the complete eps <= 2 CSR.  The oracle is the C leg's vectorised form (tests/test_oracle.py pins it
    to the scalar leg -> Python oracle -> reference goldens): 4e10 pairs take well under a minute of host time."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    N, L, eps, k = (50_000, 32, 2, 16) if cfg == "cfg2" else (200_000, 64, 2, 16)
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    kidx, kd = nat.knn_graph(p, p, k)
    indptr, idx, w = nat.eps_graph(p, p, nat.CMP_LE, eps, cap=256)
    torch.cuda.synchronize()
    ridx, rd = C.knn(tok, k, fast=True)
    assert np.array_equal(kidx.cpu().numpy(), ridx), f"{cfg}: kNN indices differ in {int((kidx.cpu().numpy() != ridx).any(axis=1).sum())} rows"
    assert np.array_equal(kd.cpu().numpy(), rd)
    rp, ri, rw = C.eps_csr(tok, 0, eps, fast=True)
    assert np.array_equal(indptr.cpu().numpy(), rp)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(w.cpu().numpy(), rw)


@one_engine
@pytest.mark.parametrize("case", ["cfg3", "clusters of 64", "139000 rows", "clusters of 12, forced", "100000 rows", "byte alphabet"])
def test_rows_beyond_a_full_round_in_column_pieces(nat, case, monkeypatch):
    """A launch whose rows need one wave more per SIMD only for a few rows (200 000 = 3 x 65 536 + 3 392) sweeps those
    rows in column pieces, merges the pieces' lists and repairs the row blocks the optimistic cap may have cut short
    (pg_api.hip knn_launch).  Against the oracle on ALL of the split-off rows, and against the unsplit launch on every
    row; cluster sizes where no piece holds k + 1 mates (64 members in 16 pieces) and where not even the whole matrix
    does (12 members: every block goes to the repair launch); 2, 7 and 16 pieces."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    k = 16
    # ("100000 rows": 32-row passes, four waves on a few SIMDs -> three everywhere + 53 blocks in pieces; "byte alphabet":
    #  the 8-plane instance holds three waves per SIMD: two whole rounds of 32-row passes + 106 blocks in pieces)
    N, members, round_rows, bits = {"cfg3": (200_000, 256, 65536, 5), "clusters of 64": (200_000, 64, 65536, 5),
                                    "139000 rows": (139_000, 256, 65536, 5), "clusters of 12, forced": (200_000, 12, 65536, 5),
                                    "100000 rows": (100_000, 256, 32768, 5), "byte alphabet": (200_000, 256, 98304, 8)}[case]
    held = (N - 1) // round_rows * round_rows
    tok = synth.clustered_tokens(N, 64, members=members)
    p = nat.pack(torch.from_numpy(tok), bits=bits)
    monkeypatch.setenv("PG_ENGINE", "mfma")                   # (no probe: 12-member clusters would go to the VALU engine)
    monkeypatch.setenv("PG_MM_SPLIT", "0")
    ref_i, ref_d = nat.knn_graph(p, p, k)
    monkeypatch.delenv("PG_MM_SPLIT")
    ridx, rd = C.knn(tok, k, row0=held, nrows=N - held, fast=True)
    for pieces in (None, "2", "7", "16") if case == "cfg3" else (None,):
        if pieces:
            monkeypatch.setenv("PG_MM_PIECES", pieces)
        kidx, kd = nat.knn_graph(p, p, k)
        torch.cuda.synchronize()
        assert np.array_equal(kidx[held:].cpu().numpy(), ridx) and np.array_equal(kd[held:].cpu().numpy(), rd), (case, pieces)
        assert bool((kidx == ref_i).all()) and bool((kd == ref_d).all()), (case, pieces)
    # a window of rows of a larger matrix (row0 > 0) that ends a few rows past a full round
    if N >= 140_000:
        kidx, kd = nat.knn_graph(p, p, k, row0=1000, nrows=131_072 + 700)
        assert bool((kidx == ref_i[1000:1000 + 131_772]).all()) and bool((kd == ref_d[1000:1000 + 131_772]).all())


@one_engine
def test_every_probe_outcome_has_a_launch_that_takes_it(nat):
    """One-cluster data below the row count of the 64-row instance: the probe says "one cluster" (gate value 2), for which
    only launches with 64-row passes carry a separate alternative - the main launch must take it itself (it did not: every
    launch of the call returned at its gate and the output was never written; tools/stress.py big found it)."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    N, k = 80_000, 16
    tok = synth.clustered_tokens(N, 64, members=N)             # one cluster
    p = nat.pack(torch.from_numpy(tok), bits=5)
    kidx = torch.full((N, k), -7, dtype=torch.int32, device=p.buf.device)
    kd = torch.full((N, k), 201, dtype=torch.uint8, device=p.buf.device)
    nat.knn_graph(p, p, k, out=(kidx, kd))
    torch.cuda.synchronize()
    assert int((kidx == -7).sum()) == 0
    ridx, rd = C.knn(tok, k, row0=40_000, nrows=2048, fast=True)
    assert np.array_equal(kidx[40_000:42_048].cpu().numpy(), ridx) and np.array_equal(kd[40_000:42_048].cpu().numpy(), rd)
    # short sequences of a byte alphabet: everything within the cap, k = 1, a window of the rows
    rng = np.random.RandomState(4)
    tok = rng.randint(0, 201, size=(135_000, 4)).astype(np.uint8)
    p = nat.pack(torch.from_numpy(tok), bits=8)
    kidx, kd = nat.knn_graph(p, p, 1, row0=26_176, nrows=97_713)
    ridx, rd = C.knn(tok, 1, row0=26_176, nrows=97_713, fast=True)
    assert np.array_equal(kidx.cpu().numpy(), ridx) and np.array_equal(kd.cpu().numpy(), rd)


@one_engine
@pytest.mark.parametrize("case", ["outliers", "sorted", "shuffled", "outliers, byte alphabet, k=40"])
def test_rows_that_lose_their_cap_are_evicted(nat, case, monkeypatch):
    """Rows without k + 1 columns inside the optimistic cap (unrelated sequences among clustered ones; members of a family
    that a sorted file puts far from the others) leave their pass and are finished by pg_knn_rows_kernel; a row is judged
    only after the sweep has passed its own position.  Every entry against the oracle, with and without eviction."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    N, k, bits = 60_000, 16, 5
    tok = synth.clustered_tokens(N, 64, members=120)
    rng = np.random.RandomState(11)
    if case.startswith("outliers"):
        rows = rng.choice(N, N // 50, replace=False)
        tok[rows] = rng.randint(1, 21, size=(len(rows), 64)).astype(np.uint8)
        tok[rows[:40]] = tok[rows[40:80]]                      # ... some of them in pairs: a near column, but not k + 1
        if "byte" in case:
            k, bits = 40, 8
    elif case == "sorted":
        tok = np.ascontiguousarray(tok[np.lexsort(tok.T[::-1])])
    else:
        tok = np.ascontiguousarray(tok[rng.permutation(N)])
    p = nat.pack(torch.from_numpy(tok), bits=bits)
    monkeypatch.setenv("PG_ENGINE", "mfma")
    ridx, rd = C.knn(tok, k, fast=True)
    for evict in ("1", "0"):
        monkeypatch.setenv("PG_MM_EVICT", evict)
        kidx, kd = nat.knn_graph(p, p, k)
        torch.cuda.synchronize()
        assert np.array_equal(kidx.cpu().numpy(), ridx) and np.array_equal(kd.cpu().numpy(), rd), (case, evict)
    monkeypatch.delenv("PG_MM_EVICT")
    kidx, kd = nat.knn_graph(p, p, k, row0=20_000, nrows=30_000)   # a window of the rows: evicted rows land in their place
    assert np.array_equal(kidx.cpu().numpy(), ridx[20_000:50_000]) and np.array_equal(kd.cpu().numpy(), rd[20_000:50_000])


@one_engine
def test_full_windows_on_dense_data_and_on_the_sharded_slice(nat):
    """Whole 4 096-row windows against the oracle where a full matrix is out of reach on the host: (i) cfg3's shape on
    DENSE data (one cluster: the folded / exact forms of the engine), (ii) rank 3's block of BASELINE.json configs[3]
    (125 000 rows x 1 000 000 columns: every kNN entry of the block, an eps window), (iii) a 400 000-row launch of the same problem, the window straddling the
    4 096th pass, i.e. reaching into the passes the persistent waves fetch through the counter."""
    from oracle import c_oracle as C
    from prograph_amd import synth, sharded
    k = 16
    tok = synth.clustered_tokens(200_000, 64, members=200_000)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    kidx, kd = nat.knn_graph(p, p, k)
    torch.cuda.synchronize()
    r0 = 150_016
    ridx, rd = C.knn(tok, k, row0=r0, nrows=4096, fast=True)
    assert np.array_equal(kidx[r0:r0 + 4096].cpu().numpy(), ridx) and np.array_equal(kd[r0:r0 + 4096].cpu().numpy(), rd)
    del p, kidx, kd
    N = 1_000_000
    tok = synth.clustered_tokens(N, 64)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    lo, hi = sharded.row_block(N, 8, 3)
    kidx, kd = nat.knn_graph(p, p, k, row0=lo, nrows=hi - lo)
    indptr, idx, w = nat.eps_graph(p, p, nat.CMP_LE, 2, row0=lo, nrows=hi - lo, cap=64)
    torch.cuda.synchronize()
    # kNN: the WHOLE block (1.25e11 pairs: half a minute of the vectorised oracle) - 64-row passes, two waves per SIMD
    ridx, rd = C.knn(tok, k, row0=lo, nrows=hi - lo, fast=True)
    assert np.array_equal(kidx.cpu().numpy(), ridx) and np.array_equal(kd.cpu().numpy(), rd)
    w0 = 60_000                                               # eps: a window of the block
    rp, ri, rw = C.eps_csr(tok, 0, 2, row0=lo + w0, nrows=4096, fast=True)
    ip = indptr.cpu().numpy()
    assert np.array_equal(ip[w0:w0 + 4097] - ip[w0], rp)
    assert np.array_equal(idx[ip[w0]:ip[w0 + 4096]].cpu().numpy(), ri) and np.array_equal(w[ip[w0]:ip[w0 + 4096]].cpu().numpy(), rw)
    del kidx, kd, indptr, idx, w
    # (iii) a launch longer than one round of the grid: 400 000 rows = 6 250 passes of 64 rows on 4 096 wave slots; the
    # window straddles pass 4 096, the first one a persistent wave fetches through the launch's counter (workspace)
    kidx, kd = nat.knn_graph(p, p, k, row0=0, nrows=400_000)
    torch.cuda.synchronize()
    w0 = 4096 * 64 - 2048
    ridx, rd = C.knn(tok, k, row0=w0, nrows=4096, fast=True)
    assert np.array_equal(kidx[w0:w0 + 4096].cpu().numpy(), ridx) and np.array_equal(kd[w0:w0 + 4096].cpu().numpy(), rd)


@pytest.mark.parametrize("sym", ["0", "1"])
def test_levenshtein_knn_vs_oracle(nat, monkeypatch, sym):
    """Build-defined banded Levenshtein kNN (no reference counterpart: parity unpinned) against the
    C oracle's plain banded Wagner–Fischer, incl. empty rows, duplicates, bands < 8 and the
    candidate-slot overflow re-run; with the rectangular and the symmetric candidate filter."""
    monkeypatch.setenv("PG_EPS_SYM", sym)
    from oracle import c_oracle as C
    from oracle import prograph_oracle as O
    from prograph_amd import synth
    tok, lens = synth.clustered_varlen_tokens(1500, Lmax=128, Lmin=96, seed=31, members=128)
    tok[7] = tok[900]
    tok[11] = 0                                    # an empty sequence
    tok[12, 5:] = 0                                # a very short one
    for band, k in [(8, 8), (3, 5), (8, 20)]:
        idx, d, st = nat.levenshtein_knn(torch.from_numpy(tok), k, band=band, return_stats=True)
        ridx, rd = C.lev_knn(tok, k, band=band)
        assert np.array_equal(d.cpu().numpy(), rd), (band, k)
        assert np.array_equal(idx.cpu().numpy(), ridx), (band, k)
    idx, d, st = nat.levenshtein_knn(torch.from_numpy(tok), 8, band=8, cap=64, return_stats=True)
    assert st["filter_passes"] == 2 and np.array_equal(idx.cpu().numpy(), C.lev_knn(tok, 8, band=8)[0])
    # the C oracle's banded DP agrees with the Python definition and with unbanded Wagner–Fischer
    rng = np.random.RandomState(2)
    for _ in range(60):
        i, j = rng.randint(0, 1500, size=2)
        la, lb = int((tok[i] != 0).sum()), int((tok[j] != 0).sum())
        full = O.levenshtein_full(tok[i], la, tok[j], lb)
        assert C.lev_pair(tok[i], tok[j], 8) == min(full, 9) == O.levenshtein_banded(tok[i], la, tok[j], lb, 8)
    # short sequences, small alphabet: many in-band pairs and ties
    rng = np.random.RandomState(4)
    small = np.zeros((400, 24), dtype=np.uint8)
    for r in range(400):
        n = rng.randint(0, 25)
        small[r, :n] = rng.randint(1, 4, size=n)
    for band in (1, 4, 8):
        idx, d = nat.levenshtein_knn(torch.from_numpy(small), 12, band=band, cap=1024)
        ridx, rd = C.lev_knn(small, 12, band=band)
        assert np.array_equal(d.cpu().numpy(), rd) and np.array_equal(idx.cpu().numpy(), ridx), band
    # row windows (sharding) and N < k+1
    idx, d = nat.levenshtein_knn(torch.from_numpy(tok), 8, band=8, row0=700, nrows=333)
    ridx, rd = C.lev_knn(tok, 8, band=8, row0=700, nrows=333)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)
    idx, d = nat.levenshtein_knn(torch.from_numpy(small[:5]), 8, band=8)
    ridx, rd = C.lev_knn(small[:5], 8, band=8)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)
    with pytest.raises(ValueError):
        bad = tok.copy(); bad[3, 2] = 0
        nat.levenshtein_knn(torch.from_numpy(bad), 8)


def test_tiny_and_rectangular_shapes(nat):
    """N = 1, N = 2, and a query block against a different database (row planes != column planes)."""
    from oracle import prograph_oracle as O
    one = np.array([[3, 4, 5, 6]], dtype=np.uint8)
    p = _planes(nat, one, 5)
    ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 3))
    assert list(ip) == [0, 0] and len(ix) == 0
    idx, d = nat.knn_graph(p, p, 2)
    assert np.all(idx.cpu().numpy() == -1) and np.all(d.cpu().numpy() == 255)
    dist, hist, fl = nat.index_flags(p, 0, want=[0])
    assert dist.cpu().numpy().tolist() == [0] and int(hist[0]) == 1 and nat.compact_flags(fl).cpu().numpy().tolist() == [0]
    two = np.array([[1, 2, 3], [1, 2, 4]], dtype=np.uint8)
    p = _planes(nat, two, 8)
    ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 1))
    assert list(ip) == [0, 1, 2] and list(ix) == [1, 0] and list(w) == [1, 1]
    idx, d = nat.knn_graph(p, p, 1)
    assert idx.cpu().numpy().tolist() == [[1], [0]] and d.cpu().numpy().tolist() == [[1], [1]]
    # rectangular: 37 query rows against a 1000-row database
    g = load_golden("synth_n1000_l32")
    db = g["tokens"]
    rng = np.random.RandomState(3)
    q = db[rng.randint(0, 1000, size=37)].copy()
    q[:, rng.randint(0, 32, size=37) % 32] = 1
    for bits in BITS:
        qp, dp = _planes(nat, q, bits), _planes(nat, db, bits)
        dref = O.hamming(db.astype(np.int64), q.astype(np.int64)).numpy()          # (37, 1000)
        assert np.array_equal(nat.hamming_dense(dp, qp).cpu().numpy(), dref)
        ip, ix, w = _csr_np(nat.eps_graph(qp, dp, nat.CMP_LE, 3))
        mask = (dref <= 3) & (dref > 0)
        assert np.array_equal(ip, np.concatenate([[0], np.cumsum(mask.sum(1))]))
        assert np.array_equal(ix, np.nonzero(mask)[1]) and np.array_equal(w, dref[mask])
        idx, d = nat.knn_graph(qp, dp, 5)
        order = np.argsort(dref, axis=1, kind="stable")[:, 1:6]
        assert np.array_equal(idx.cpu().numpy(), order) and np.array_equal(d.cpu().numpy(), np.take_along_axis(dref, order, 1))


def test_many_rows_per_wave_and_env_override(nat, monkeypatch):
    """Force few waves (many passes per wave) and many waves (1 row per wave): same results."""
    g = load_golden("synth_n2085_l64")
    p = _planes(nat, g["tokens"], 5)
    for wpc in ("4", "128"):
        monkeypatch.setenv("PG_WAVES_PER_CU", wpc)
        ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 4, cap=32))
        assert np.array_equal(ip, g["eps4_indptr"]) and np.array_equal(ix, g["eps4_indices"]) and np.array_equal(w, g["eps4_weights"])
        idx, d = nat.knn_graph(p, p, 16)
        assert np.array_equal(idx.cpu().numpy(), g["knn16_idx"]) and np.array_equal(d.cpu().numpy(), g["knn16_w"])


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_lower_bound_filter_modes_agree(nat, monkeypatch, mode):
    """PG_LB_FILTER = 0 (direct form only), 1 (adaptive), 2 (two-stage form forced): the plane-0
    lower-bound stage is exact, so every mode must reproduce the golden vectors — on clustered data
    (filter mostly skips), on dense data (every row-step triggers) and for all comparators."""
    monkeypatch.setenv("PG_LB_FILTER", mode)
    for name in ("synth_n2085_l64", "synth_n515_l20_dups", "ref_synthetic_csv"):
        g = load_golden(name)
        for bits in BITS:
            p = _planes(nat, g["tokens"], bits)
            for key in g.files:
                if key.endswith("_indptr") and "sub" not in key and "sim" not in key and "_b5" not in key:
                    base = key[:-7]
                    parts = base.split("_")
                    cmp = {"eq": nat.CMP_EQ, "lt": nat.CMP_LT, "ge": nat.CMP_GE, "gt": nat.CMP_GT}[parts[1]] if len(parts) > 1 else nat.CMP_LE
                    ip, ix, w = _csr_np(nat.eps_graph(p, p, cmp, int(parts[0][3:]), cap=64))
                    assert np.array_equal(ip, g[base + "_indptr"]) and np.array_equal(ix, g[base + "_indices"]), (name, base, mode)
                    assert np.array_equal(w, g[base + "_weights"])
                if key.startswith("knn") and key.endswith("_idx") and "sub" not in key and "sim" not in key:
                    k = int(key[3:-4])
                    idx, d = nat.knn_graph(p, p, k)
                    assert np.array_equal(idx.cpu().numpy(), g[f"knn{k}_idx"]) and np.array_equal(d.cpu().numpy(), g[f"knn{k}_w"]), (name, k, mode)
    # a long sweep (many tiles, several adaptive windows) against the C oracle
    from oracle import c_oracle as C
    from prograph_amd import synth
    tok = synth.clustered_tokens(30_000, 64, seed=77, members=64)
    p = _planes(nat, tok, 5)
    ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 3, cap=128))
    rip, rix, rw = C.eps_csr(tok, 0, 3)
    assert np.array_equal(ip, rip) and np.array_equal(ix, rix) and np.array_equal(w, rw)
    idx, d = nat.knn_graph(p, p, 16)
    ridx, rd = C.knn(tok, 16)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)


@pytest.mark.parametrize("sym", ["0", "1"])
def test_eps_symmetric_path_matches_rectangular(nat, monkeypatch, sym):
    """pg_eps_slots_sym / pg_eps_compact_sym (every unordered pair once, the transposed half delivered
    through atomic back-of-slot appends and rank-sorted at compaction) against the goldens and the C
    oracle: every comparator, slot overflow (cap 4), rows with hundreds of entries from below (LDS
    rank sort, and beyond 512 the recompute path), dense data, duplicates.  PG_EPS_SYM=0 keeps the
    rectangular path, 1 forces the symmetric one also on small inputs (auto: from 32k rows)."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    monkeypatch.setenv("PG_EPS_SYM", sym)
    for name in ("synth_n2085_l64", "synth_n515_l20_dups", "ref_synthetic_csv"):
        g = load_golden(name)
        for bits in BITS:
            p = _planes(nat, g["tokens"], bits)
            for key in g.files:
                if key.endswith("_indptr") and "sub" not in key and "sim" not in key and "_b5" not in key:
                    base = key[:-7]
                    parts = base.split("_")
                    cmp = {"eq": nat.CMP_EQ, "lt": nat.CMP_LT, "ge": nat.CMP_GE, "gt": nat.CMP_GT}[parts[1]] if len(parts) > 1 else nat.CMP_LE
                    for cap in (4, 64):
                        ip, ix, w = _csr_np(nat.eps_graph(p, p, cmp, int(parts[0][3:]), cap=cap))
                        assert np.array_equal(ip, g[base + "_indptr"]) and np.array_equal(ix, g[base + "_indices"]), (name, base, cap)
                        assert np.array_equal(w, g[base + "_weights"])
    # one dense cluster: every row has ~all others within eps (back parts of 100s .. 1000s of entries)
    tok = synth.clustered_tokens(3000, 64, seed=21, members=3000)
    p = _planes(nat, tok, 5)
    for eps, cap in ((2, 256), (4, 1024), (6, 4096), (6, 64)):
        ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, eps, cap=cap))
        rip, rix, rw = C.eps_csr(tok, nat.CMP_LE, eps)
        assert np.array_equal(ip, rip) and np.array_equal(ix, rix) and np.array_equal(w, rw), (eps, cap)
    # a longer sweep, clusters + loose rows, GE comparator (most pairs match)
    tok = np.concatenate([synth.clustered_tokens(20000, 40, seed=5, members=50),
                          np.random.RandomState(3).randint(1, 21, size=(2000, 40)).astype(np.uint8)])
    p = _planes(nat, tok, 5)
    ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_LE, 3, cap=64))
    rip, rix, rw = C.eps_csr(tok, nat.CMP_LE, 3)
    assert np.array_equal(ip, rip) and np.array_equal(ix, rix) and np.array_equal(w, rw)
    tok = synth.clustered_tokens(1500, 24, seed=6, members=30)
    p = _planes(nat, tok, 5)
    ip, ix, w = _csr_np(nat.eps_graph(p, p, nat.CMP_GE, 20, cap=32))
    rip, rix, rw = C.eps_csr(tok, nat.CMP_GE, 20)
    assert np.array_equal(ip, rip) and np.array_equal(ix, rix) and np.array_equal(w, rw)


@pytest.mark.parametrize("guess", ["0", "3", "8", "40", "8/R2"])
def test_knn_optimistic_cap_is_exact(nat, monkeypatch, guess):
    """PG_KNN_GUESS (the optimistic stage-1 cap of the kNN engine) never changes results: rows that
    settle below the cap, rows that lose it at the first checkpoint (no near column: random rows),
    rows that lose it at the second (near columns, but fewer than k + 1 below the cap), the second
    sweep of the early tiles with ties decided by full keys and duplicates skipped, continuation
    rounds (k > 63) and row windows.  0 switches the mechanism off.  "8/R2": the same on the MFMA engine's 64-row
    instance (two row blocks per pass; no effect on the VALU engine)."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    if guess.endswith("/R2"):
        guess = guess[:-3]
        monkeypatch.setenv("PG_MM_R", "2")
    monkeypatch.setenv("PG_KNN_GUESS", guess)
    rng = np.random.RandomState(11)
    clustered = synth.clustered_tokens(9000, 64, seed=3, members=128)
    small = synth.clustered_tokens(3000, 64, seed=4, members=12)          # 11 mates < k: bound far above the cap
    small[small > 0] = (small[small > 0] % 20) + 1
    loose = rng.randint(1, 21, size=(3000, 64)).astype(np.uint8)           # no structure at all
    tok = np.concatenate([clustered, small, loose])[rng.permutation(15000)]
    tok[100] = tok[7000]; tok[101] = tok[7000]; tok[14000] = tok[7000]     # duplicates, early and late columns
    tok[200, :60] = tok[9000, :60]                                        # a near pair far apart in column order
    for bits in BITS:
        p = _planes(nat, tok, bits)
        for k in (1, 16, 63, 70):
            idx, d = nat.knn_graph(p, p, k)
            ridx, rd = C.knn(tok, k)
            assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd), (bits, k, guess)
        idx, d = nat.knn_graph(p, p, 16, row0=5000, nrows=4097)
        ridx, rd = C.knn(tok, 16, row0=5000, nrows=4097)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)
    # longer records (one column per lane) and a short sweep (checkpoints at tile 1)
    tok = synth.clustered_tokens(4000, 128, seed=8, members=40)
    p = _planes(nat, tok, 5)
    idx, d = nat.knn_graph(p, p, 16)
    ridx, rd = C.knn(tok, 16)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)
    tok = synth.clustered_tokens(300, 32, seed=9, members=30)
    p = _planes(nat, tok, 5)
    idx, d = nat.knn_graph(p, p, 8)
    ridx, rd = C.knn(tok, 8)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)


@one_engine
def test_c_abi_without_python(nat):
    """tests/capi/capi_check.cpp: a plain C++ host (hipMalloc, default stream, the declarations of
    include/prograph_hip.h, no Python, no torch) drives pack -> kNN -> eps CSR through both the
    rectangular and the symmetric entry points and compares with the C oracle."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "capi", "_build", "capi_check")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(here, "capi")])
    out = subprocess.run([exe, "30000", "64"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C ABI OK" in out.stdout, out.stdout + out.stderr


def test_randomised_shapes_against_c_oracle(nat):
    """Seeded sweep over shapes that hit every template instance (G = 1..4 groups, 5 / 8 bit planes,
    C = 4 / 2 / 1 columns per lane), odd N, odd L, tiny and skewed alphabets, every comparator,
    random eps / k, random row windows — eps CSR, kNN and dense against the C oracle."""
    from oracle import c_oracle as C
    rng = np.random.RandomState(20260104)
    for it in range(40):
        N = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 256, 257, 700, 1500, 2600]))
        amax = int(rng.choice([1, 2, 4, 20, 31, 32, 127, 255]))
        L = int(rng.randint(1, 256 if amax <= 31 else 129))
        base = rng.randint(0, amax + 1, size=(max(1, N // 40), L))
        tok = base[rng.randint(0, len(base), size=N)].copy()
        nmut = rng.randint(0, 4, size=N)
        for r in range(N):
            for _ in range(nmut[r]):
                tok[r, rng.randint(0, L)] = rng.randint(0, amax + 1)
        tok = tok.astype(np.uint8)
        bits = 5 if (amax <= 31 and (L > 128 or rng.rand() < 0.7)) else 8
        p = _planes(nat, tok, bits)
        row0 = int(rng.randint(0, N)); nrows = int(rng.randint(1, N - row0 + 1))
        cmp = int(rng.randint(0, 5)); eps = float(rng.choice([1, 2, 3, 5, 2.5, L, L + 3]))
        rip, rix, rw = C.eps_csr(tok, cmp, eps, row0=row0, nrows=nrows)
        ip, ix, w = _csr_np(nat.eps_graph(p, p, cmp, eps, row0=row0, nrows=nrows, cap=int(rng.choice([1, 16, 256]))))
        assert np.array_equal(ip, rip) and np.array_equal(ix, rix) and np.array_equal(w, rw), (it, N, L, bits, cmp, eps)
        k = int(rng.choice([1, 2, 7, 16, 33, 63]))
        ridx, rd = C.knn(tok, k, row0=row0, nrows=nrows)
        idx, d = nat.knn_graph(p, p, k, row0=row0, nrows=nrows)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd), (it, N, L, bits, k)
        if N <= 700:
            m = min(N, 50)
            assert np.array_equal(nat.hamming_dense(p, _planes(nat, tok[:m], bits)).cpu().numpy(), C.hamming(tok, tok[:m])), (it, N, L)


def test_cfg5_full_size_properties(nat):
    """BASELINE.json configs[4] at full size (N = 200 000, L <= 128, band 8, k = 8; build defined,
    parity unpinned): ordering and range properties of every row, mutual consistency of the distances
    (edit distance is symmetric), and sampled rows against the C oracle's banded Wagner-Fischer."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    N, k, band = 200_000, 8, 8
    tok, lens = synth.clustered_varlen_tokens(N, Lmax=128, Lmin=96)
    idx, d, st = nat.levenshtein_knn(torch.from_numpy(tok), k, band=band, return_stats=True)
    idx, d = idx.cpu().numpy().astype(np.int64), d.cpu().numpy().astype(np.int64)
    assert st["symmetric"] and idx.min() >= 0 and idx.max() < N and d.min() >= 0 and d.max() <= band + 1
    key = d * (1 << 24) + idx
    assert np.all(key[:, 1:] > key[:, :-1])                      # canonical (distance, column) order, no repeats
    # rank 0 is dropped; a row can only meet itself again behind an exact duplicate with a smaller column
    assert np.all(d[idx == np.arange(N)[:, None]] == 0)
    # symmetry: if j is i's neighbour at distance x <= band and i appears in j's list, the distance agrees
    rows = np.repeat(np.arange(N), k)
    cols = idx.reshape(-1); dist = d.reshape(-1)
    fwd = dict(zip(zip(rows[:200000].tolist(), cols[:200000].tolist()), dist[:200000].tolist()))
    for (i, j), x in list(fwd.items())[:20000]:
        back = np.nonzero(idx[j] == i)[0]
        if len(back):
            assert d[j, back[0]] == x
    for r0 in (0, 77_777, N - 16):
        ridx, rd = C.lev_knn(tok, k, band=band, row0=r0, nrows=16)
        assert np.array_equal(idx[r0:r0 + 16], ridx) and np.array_equal(d[r0:r0 + 16], rd), r0


@one_engine
def test_hamming_operator_cache_tracks_in_place_edits(nat):
    """hamming() remembers the packed form of a device-resident X across calls (the reference's
    batch loop passes the same X every time); an in-place edit of X must invalidate it."""
    from prograph_amd.distance import hamming
    from oracle import prograph_oracle as O
    rng = np.random.RandomState(8)
    Xh = rng.randint(0, 21, size=(3000, 40)).astype(np.int64)
    X = torch.from_numpy(Xh).cuda()
    Y = X[:8].clone()
    for _ in range(2):
        assert np.array_equal(hamming(X, Y).cpu().numpy(), O.hamming(Xh, Xh[:8]).numpy())
    X[5, 3] = 0; X[5, 4] = 7                      # in place: the version counter moves
    Xh[5, 3] = 0; Xh[5, 4] = 7
    out = hamming(X, Y)
    assert out.device == X.device and np.array_equal(out.cpu().numpy(), O.hamming(Xh, Y.cpu().numpy()).numpy())
    X[9, 0] = 200                                 # now a byte alphabet: repacked with 8 planes
    Xh[9, 0] = 200
    assert np.array_equal(hamming(X, Y).cpu().numpy(), O.hamming(Xh, Y.cpu().numpy()).numpy())


@one_engine
@pytest.mark.parametrize("d,hi", [(256, 21), (700, 21), (129, 256), (300, 256), (1000, 256)])
def test_hamming_operator_long_sequences(nat, d, hi):
    """Sequences longer than one record: the operator sums the dense kernel over column segments
    (accumulate mode of pg_hamming_dense) and must equal the reference expression bit for bit."""
    from prograph_amd.distance import hamming
    from oracle import prograph_oracle as O
    rng = np.random.RandomState(d)
    Xh = rng.randint(0, hi, size=(777, d)).astype(np.int64)
    Xh[100:200] = Xh[0]                            # many small distances as well
    Xh[100:200, ::7] = (Xh[100:200, ::7] + rng.randint(0, 2, size=Xh[100:200, ::7].shape)) % hi
    Yh = Xh[rng.randint(0, 777, size=50)]
    want = O.hamming(Xh, Yh).numpy()
    got = hamming(torch.from_numpy(Xh).cuda(), torch.from_numpy(Yh).cuda())
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want)
    sim = hamming(torch.from_numpy(Xh).cuda(), torch.from_numpy(Yh[:, : d - 5]).cuda(), similarity=True)
    Yp = np.concatenate([Yh[:, : d - 5], np.zeros((50, 5), dtype=np.int64)], axis=1)
    assert np.array_equal(sim.cpu().numpy(), O.hamming(Xh, Yp, similarity=True).numpy())


def test_cfg4_slice_properties(nat):
    """BASELINE.json configs[3] on one GPU: rank 3's row block (125 000 rows) of the N = 1 000 000,
    L = 64 problem against all columns, kNN k = 16 and eps <= 2.  Checked through sampled rows against
    the oracle's 1 x N computation and through ordering / range properties of every row."""
    from oracle import prograph_oracle as O
    from prograph_amd import synth, sharded
    N, L, k = 1_000_000, 64, 16
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    lo, hi = sharded.row_block(N, 8, 3)
    assert (lo, hi) == (375_000, 500_000)
    kidx, kd = nat.knn_graph(p, p, k, row0=lo, nrows=hi - lo)
    indptr, idx, w = nat.eps_graph(p, p, nat.CMP_LE, 2, row0=lo, nrows=hi - lo, cap=64)
    torch.cuda.synchronize()
    kidx, kd = kidx.cpu().numpy().astype(np.int64), kd.cpu().numpy().astype(np.int64)
    indptr, idx, w = indptr.cpu().numpy(), idx.cpu().numpy().astype(np.int64), w.cpu().numpy()
    kk = kd * (1 << 24) + kidx
    assert np.all(kk[:, 1:] > kk[:, :-1]) and kidx.min() >= 0 and kidx.max() < N
    rows = np.repeat(np.arange(hi - lo), np.diff(indptr))
    same = rows[1:] == rows[:-1]
    assert np.all(idx[1:][same] > idx[:-1][same]) and w.min() >= 1 and w.max() <= 2
    t64 = tok.astype(np.int64)
    for r in np.random.RandomState(4).choice(hi - lo, size=8, replace=False):
        d = O.hamming(t64, t64[lo + r].reshape(1, -1)).numpy()[0]
        order = np.argsort(d, kind="stable")[1:k + 1]
        assert np.array_equal(kidx[r], order) and np.array_equal(kd[r], d[order])
        cols = np.where((d <= 2) & (d > 0))[0]
        assert np.array_equal(idx[indptr[r]:indptr[r + 1]], cols) and np.array_equal(w[indptr[r]:indptr[r + 1]], d[cols])


def test_knn_beyond_63_neighbours(nat):
    """k > 63 runs as continuation rounds (63 + 64 + ... ranks, each round restarts after the last
    (distance, column) key): same canonical order as one stable sort, incl. ties, duplicates, k >= N."""
    from oracle import c_oracle as C
    from prograph_amd import synth
    tok = synth.clustered_tokens(3000, 40, seed=5, members=500)
    tok[10] = tok[2000]; tok[11] = tok[2000]
    for bits in BITS:
        p = _planes(nat, tok, bits)
        for k in (64, 100, 127, 128, 200):
            idx, d = nat.knn_graph(p, p, k)
            ridx, rd = C.knn(tok, k)
            assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd), (bits, k)
        idx, d = nat.knn_graph(p, p, 150, row0=700, nrows=300)
        ridx, rd = C.knn(tok, 150, row0=700, nrows=300)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)
    small = tok[:90]
    p = _planes(nat, small, 5)
    idx, d = nat.knn_graph(p, p, 130)                     # more ranks than sequences: -1 / 255 beyond N-1
    ridx, rd = C.knn(small, 130)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)
    assert np.all(idx.cpu().numpy()[:, 89:] == -1)


@pytest.mark.parametrize("case", ["knn", "eps1", "eps2"])
def test_dense_mutant_library_is_exact(nat, engine, case):
    """
    DENSE data - one cluster, a mutant library around one seed, every pair within 6 substitutions (what
    prograph is used on; the reference's own data/synthetic_data.csv is of this kind) - at full size,
    against the C oracle on row windows.  kNN k=16 at N = 200 000 (BASELINE configs[2]'s shape) and the
    epsilon graphs at N = 50 000: eps <= 1 (the constructor's default graph) and eps <= 2, where most
    rows outgrow any slot and the engine's fill pass (pg_eps_fill_rows) writes the CSR.
    """
    import time
    from oracle import c_oracle as C
    from prograph_amd import synth
    if case == "knn":
        N, L, k = 200_000, 64, 16
        tok = synth.clustered_tokens(N, L, members=N)
        p = nat.pack(torch.from_numpy(tok), bits=5)
        nat.knn_graph(p, p, k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        kidx, kd = nat.knn_graph(p, p, k)
        torch.cuda.synchronize()
        print(f"\ndense kNN k=16 N={N} L={L} [{engine}]: {(time.perf_counter() - t0) * 1e3:.2f} ms")
        kidx, kd = kidx.cpu().numpy(), kd.cpu().numpy()
        for r0 in (0, 77_777, N - 48):
            ri, rd = C.knn(tok, k, row0=r0, nrows=48)
            assert np.array_equal(kidx[r0:r0 + 48], ri) and np.array_equal(kd[r0:r0 + 48], rd), r0
        return
    N, L, eps = 50_000, 64, (1 if case == "eps1" else 2)
    tok = synth.clustered_tokens(N, L, members=N)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    nat.eps_graph(p, p, nat.CMP_LE, eps); torch.cuda.synchronize()
    t0 = time.perf_counter()
    indptr, idx, w = nat.eps_graph(p, p, nat.CMP_LE, eps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    indptr = indptr.cpu().numpy()
    print(f"\ndense eps<={eps} N={N} L={L} [{engine}]: {dt:.2f} ms, nnz {int(indptr[-1])}, max degree {int(np.diff(indptr).max())}")
    for r0 in (0, 31_111, N - 40):
        ip, ix, ww = C.eps_csr(tok, 0, eps, row0=r0, nrows=40)
        a, b = int(indptr[r0]), int(indptr[r0 + 40])
        assert np.array_equal(indptr[r0:r0 + 41] - indptr[r0], ip), r0
        assert np.array_equal(idx[a:b].cpu().numpy(), ix) and np.array_equal(w[a:b].cpu().numpy(), ww), r0
    # symmetry of the whole graph through degrees: in-degree == out-degree for a symmetric relation
    indeg = torch.bincount(idx.to(torch.int64), minlength=N).cpu().numpy()
    assert np.array_equal(indeg, np.diff(indptr))


@one_engine
def test_allgather_tokens_through_the_c_abi(nat):
    """pg_comm_* / pg_allgather_tokens: RCCL bound inside the library, no torch.distributed on the data path.
    One rank (this box has one GPU): the gather of a padded shard must reproduce it."""
    comm = nat.comm_init(1, 0, nat.comm_unique_id())
    try:
        tok = torch.randint(0, 21, (12_345, 64), dtype=torch.uint8, device=nat.device())
        full = nat.allgather_tokens(comm, tok, 1)
        torch.cuda.synchronize()
        assert full.shape == tok.shape and torch.equal(full, tok)
    finally:
        nat.comm_destroy(comm)


@one_engine
def test_minkowski_f16_against_the_reference(nat):
    """
    SURVEY.md §8 f2 through the C ABI: pg_pack_f16 + pg_minkowski_dense + pg_f16_knn / pg_f16_eps_* against
    outputs of the REAL reference (tests/golden/minkowski_f16.npz, oracle/gen_golden.py::gen_minkowski).
    Float tolerance: the kernel rounds every elementwise step to fp16 like the reference's fp16 tensor
    expression; only the float accumulation order of the sum differs, which can move a result by ONE fp16
    ulp when the sum sits on a rounding boundary: none at D <= 64 (bit-exact), < 0.1 % of the pairs at
    D = 1280.  Graph indices are exact for D <= 64; at D = 1280 a one-ulp move among near-equal distances can
    reorder a row's neighbours, so >= 98 % of the rows must be identical and every weight within one ulp.
    """
    g = load_golden("minkowski_f16")
    dev = nat.device()
    for name in ("d2", "d64", "d1280"):
        e = torch.from_numpy(g[f"{name}_emb"]).to(dev)
        n = e.shape[0]
        xp = nat.pack_f16(e)
        blk = nat.minkowski_dense(xp, nat.pack_f16(e[:64]))
        got, want = blk.cpu().numpy(), g[f"{name}_dist64"]
        ulp = np.abs(got.view(np.int16).astype(np.int64) - want.view(np.int16).astype(np.int64))
        if name == "d1280":
            assert ulp.max() <= 1 and (ulp != 0).mean() < 1e-3, (name, int(ulp.max()), float((ulp != 0).mean()))
        else:
            assert ulp.max() == 0, (name, int(ulp.max()))
        full = nat.minkowski_dense(xp, xp)
        for k in (1, 5, 16):
            idx, w = nat.f16_knn(full, k, first=1)
            idx, w = idx.cpu().numpy(), w.cpu().numpy()
            wi, ww = g[f"{name}_knn{k}_idx"], g[f"{name}_knn{k}_w"]
            wulp = np.abs(w.view(np.int16).astype(np.int64) - ww.view(np.int16).astype(np.int64))
            assert wulp.max() <= (1 if name == "d1280" else 0)
            same = (idx == wi).all(1)
            assert same.mean() >= (0.98 if name == "d1280" else 1.0), (name, k, float(same.mean()))
        sidx, sw = nat.f16_knn(nat.minkowski_dense(xp, xp, similarity=True), 4, first=1, descending=True)
        same = (sidx.cpu().numpy() == g[f"{name}_knn4_sim_idx"]).all(1)
        assert same.mean() >= (0.98 if name == "d1280" else 1.0)
        if name != "d1280":
            assert np.array_equal(sw.cpu().numpy(), g[f"{name}_knn4_sim_w"])
        eps = float(g[f"{name}_eps"])
        for sim, key in ((False, "eps"), (True, "eps_sim")):
            blk = nat.minkowski_dense(xp, xp, similarity=sim)
            ip, ix, w = nat.f16_eps(blk, nat.CMP_LE, 1 / (1 + eps) if sim else eps, similarity=sim)
            ip, ix, w = ip.cpu().numpy(), ix.cpu().numpy(), w.cpu().numpy()
            if name == "d1280":          # an entry whose distance sits one ulp across the threshold may differ
                assert abs(int(ip[-1]) - int(g[f"{name}_{key}_indptr"][-1])) <= 0.002 * int(ip[-1]) + 2
            else:
                assert np.array_equal(ip, g[f"{name}_{key}_indptr"]) and np.array_equal(ix, g[f"{name}_{key}_indices"])
                assert np.array_equal(w.astype(np.float64), g[f"{name}_{key}_weights"].astype(np.float64))


def test_eps_graph_with_more_than_2_31_entries(nat, engine):
    """Every pair matches (eps >= L): nnz = N*(N-1) - duplicates > 2^31, so a row's place in the CSR needs all
    64 bits (regression: the fill pass composed it from two v_readlane halves with a sign-extending low half).
    Checked on the rows whose offsets lie beyond 2^31 and 2^32-adjacent boundaries, against numpy."""
    N, L = 48_000, 4
    rng = np.random.RandomState(9)
    tok = rng.randint(0, 201, size=(N, L)).astype(np.uint8)
    tok[N - 1] = tok[7]                                            # one duplicate pair: d == 0 is excluded
    p = nat.pack(torch.from_numpy(tok), bits=8)
    indptr, idx, w = nat.eps_graph(p, p, nat.CMP_LE, 4, cap=64)
    torch.cuda.synchronize()
    ip = indptr.cpu().numpy()
    assert int(ip[-1]) == N * (N - 1) - 2 and int(ip[-1]) > 2 ** 31
    for r in (0, 7, int(np.searchsorted(ip, 2 ** 31)) - 1, int(np.searchsorted(ip, 2 ** 31)), N - 2, N - 1):
        d = (tok != tok[r]).sum(1)
        cols = np.nonzero(d > 0)[0]
        a, b = int(ip[r]), int(ip[r + 1])
        assert np.array_equal(idx[a:b].cpu().numpy(), cols) and np.array_equal(w[a:b].cpu().numpy(), d[cols]), r


@one_engine
@pytest.mark.gpu
def test_mfma_fp4_operand_layout_and_exactness():
    """tools/ubench/mfma_fp4.hip: v_mfma_f32_32x32x64_f8f6f4 with FP4 operands against a CPU product - the lane /
    nibble layout the pack kernel and the engine's row operand assume, and exact sums of the values they use."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "..", "tools", "ubench", "mfma_fp4.hip")
    exe = os.path.join(here, "capi", "_build", "mfma_fp4")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-result", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "layout confirmed" in out.stdout, out.stdout + out.stderr


@one_engine
def test_concurrent_launches_on_two_streams(nat, monkeypatch):
    """The MFMA engine hands its passes out through a per-launch counter (persistent waves): launches that overlap
    on different streams must not share one.  Two kNN sweeps of different row windows run concurrently, several
    times over, and are compared with serial runs."""
    from prograph_amd import synth
    monkeypatch.setenv("PG_ENGINE", "mfma")
    tok = synth.clustered_tokens(150000, 64, seed=5)
    p = _planes(nat, tok, 5)
    ref_a = nat.knn_graph(p, p, 8, row0=0, nrows=140000)
    ref_b = nat.knn_graph(p, p, 8, row0=10000, nrows=140000)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(4):
        with torch.cuda.stream(sa):
            a = nat.knn_graph(p, p, 8, row0=0, nrows=140000)
        with torch.cuda.stream(sb):
            b = nat.knn_graph(p, p, 8, row0=10000, nrows=140000)
        torch.cuda.synchronize()
        assert torch.equal(a[0], ref_a[0]) and torch.equal(a[1], ref_a[1])
        assert torch.equal(b[0], ref_b[0]) and torch.equal(b[1], ref_b[1])


@one_engine
def test_many_short_launches_beside_a_long_one(nat, monkeypatch):
    """The pass counter of a launch lives in caller-owned workspace (ABI 3; it used to be one of a ring of 256 words
    inside the library, so the 257th launch after a still-running one shared its counter).  One long MFMA-engine
    launch on stream A, 300 short ones on stream B meanwhile; both must equal their serial results."""
    from prograph_amd import synth
    monkeypatch.setenv("PG_ENGINE", "mfma")
    big = _planes(nat, synth.clustered_tokens(400000, 64, seed=11), 5)
    small = _planes(nat, synth.clustered_tokens(600, 64, seed=12, members=64), 5)
    ref_big = nat.knn_graph(big, big, 8)
    ref_small = nat.knn_graph(small, small, 8)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        a = nat.knn_graph(big, big, 8)
    outs = []
    with torch.cuda.stream(sb):
        for _ in range(300):
            outs.append(nat.knn_graph(small, small, 8))
    torch.cuda.synchronize()
    assert torch.equal(a[0], ref_big[0]) and torch.equal(a[1], ref_big[1])
    for o in outs:
        assert torch.equal(o[0], ref_small[0]) and torch.equal(o[1], ref_small[1])


@one_engine
def test_second_device_when_present(nat, monkeypatch):
    """Nothing on the device is shared between launches: the same calls work on another GPU of the process
    (the pass counters used to be allocated once, on whichever device came first)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU")
    from prograph_amd import synth
    monkeypatch.setenv("PG_ENGINE", "mfma")
    tok = synth.clustered_tokens(70000, 64, seed=3)
    res = []
    for d in (0, 1):
        with torch.cuda.device(d):
            p = _planes(nat, tok, 5)
            idx, dist = nat.knn_graph(p, p, 8)
            res.append((idx.cpu(), dist.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@one_engine
@pytest.mark.parametrize("force", ["0", "1", "2", "3", "4"])
def test_probe_gated_alternatives_agree(nat, monkeypatch, force):
    """Large launches carry two alternatives (MFMA / VALU engine for kNN; symmetric / rectangular sweep for the whole
    eps graph) and a device-side probe decides which one runs (pg_api.hip: run_probe, NsqParams::gate).  Forcing the
    decision either way must give the identical result - and the unforced decision must be one of them."""
    from prograph_amd import synth
    monkeypatch.delenv("PG_GATE_FORCE", raising=False)
    # (above PG_PROBE_MIN_N; force 4 = "one cluster": the 32-row instance where 64-row passes were planned - from ~157k rows)
    tok = synth.clustered_tokens(160_000 if force == "4" else 70_000, 64, seed=21, members=128)
    p = _planes(nat, tok, 5)
    ref_k = nat.knn_graph(p, p, 12)
    ref_e = nat.eps_graph(p, p, nat.CMP_LE, 2)
    monkeypatch.setenv("PG_GATE_FORCE", force)
    got_k = nat.knn_graph(p, p, 12)
    got_e = nat.eps_graph(p, p, nat.CMP_LE, 2)
    assert torch.equal(got_k[0], ref_k[0]) and torch.equal(got_k[1], ref_k[1])
    assert all(torch.equal(a, b) for a, b in zip(got_e, ref_e))
    # unclustered data: the probe sends kNN to the VALU engine; both engines agree on it anyway
    monkeypatch.delenv("PG_GATE_FORCE", raising=False)
    rnd = np.random.RandomState(8).randint(1, 21, size=(66_000, 64)).astype(np.uint8)
    pr = _planes(nat, rnd, 5)
    auto = nat.knn_graph(pr, pr, 5)
    monkeypatch.setenv("PG_GATE_FORCE", "0")
    mfma = nat.knn_graph(pr, pr, 5)
    assert torch.equal(auto[0], mfma[0]) and torch.equal(auto[1], mfma[1])


@one_engine
def test_bench_json_line_contract():
    """bench.py prints ONE JSON line with the driver's keys, a roofline object (bound / achieved / peak / unit /
    frac / traffic) and a cpu_baseline slot; run on the smallest workload with two steps."""
    import json, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PG_")}      # the bench's own engine choice
    out = subprocess.run([sys.executable, os.path.join(here, "..", "bench.py"), "--workload", "cfg2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["higher_is_better"] is True
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and "workload" in rec["config"] and rec["dtype"] == "u8"
    roof = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("valu", "hbm", "mfma") and (roof["frac"] is None or 0 < roof["frac"] <= 1)
