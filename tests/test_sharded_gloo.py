"""
N > 1 path on CPU: world_size-2 (and 3) `gloo` process groups exercise the row-block
partitioning, the one all-gather of the token shards (incl. a short last block) and the
concatenation of per-rank CSR / kNN slices.  The compute entry points are answered by the
test-only fake backend (tests/fake_native.py = the oracle); on the GPU box the same host code
runs on the HIP kernels (tests/test_gpu_native.py::test_row_window_equals_full covers the
row-window kernels themselves).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, load_golden


def _worker(rank, world, port, name, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fake_native
        from prograph_amd import _native, sharded

        class MP:            # minimal monkeypatch stand-in
            @staticmethod
            def setattr(obj, name, val):
                setattr(obj, name, val)
        fake_native.install(MP)
        g = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
        tok = g["tokens"]
        n = tok.shape[0]
        lo, hi = sharded.row_block(n, world, rank)
        local = torch.from_numpy(tok[lo:hi].copy())
        full = sharded.allgather_tokens(local, n)
        assert np.array_equal(full.numpy(), tok)
        ge = sharded.build_graph_sharded(local, n, eps=2)
        gk = sharded.build_graph_sharded(local, n, k=16)
        assert ge.row0 == lo and gk.row0 == lo
        csr = sharded.gather_csr_to_host(ge)
        kidx, kw = gk.host()
        allk = [None] * world
        dist.all_gather_object(allk, (kidx, kw))
        if rank == 0:
            q.put(("ok", csr, np.concatenate([a[0] for a in allk]), np.concatenate([a[1] for a in allk])))
    except Exception as e:      # surface the failure in the parent
        if rank == 0:
            q.put(("err", repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "synth_n1000_l32"), (3, "synth_n2085_l64")])
def test_row_block_sharding_matches_single(world, name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == "ok", res
    _, (indptr, idx, w), kidx, kw = res
    g = load_golden(name)
    assert np.array_equal(indptr, g["eps2_indptr"]) and np.array_equal(idx, g["eps2_indices"])
    assert np.array_equal(w, g["eps2_weights"])
    assert np.array_equal(kidx, g["knn16_idx"]) and np.array_equal(kw, g["knn16_w"])


def test_row_blocks_partition_exactly():
    from prograph_amd import sharded
    for n in (1, 7, 8, 9, 1000, 1_000_000, 200_001):
        for world in (1, 2, 3, 4, 8):
            blocks = sharded.shard_rows(n, world)
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert all(0 <= hi - lo <= -(-n // world) for lo, hi in blocks)


def _gpu_worker(rank, world, port, name, q):
    """Two ranks share the one GPU of the test box: real HIP kernels, gloo collectives."""
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from prograph_amd import sharded
        torch.cuda.set_device(0)
        g = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
        tok = g["tokens"]
        n = tok.shape[0]
        lo, hi = sharded.row_block(n, world, rank)
        local = torch.from_numpy(tok[lo:hi].copy()).cuda()
        ge = sharded.build_graph_sharded(local, n, eps=2, bits=5)
        gk = sharded.build_graph_sharded(local, n, k=16, bits=5)
        csr = sharded.gather_csr_to_host(ge)
        allk = [None] * world
        dist.all_gather_object(allk, gk.host())
        if rank == 0:
            q.put(("ok", csr, np.concatenate([a[0] for a in allk]), np.concatenate([a[1] for a in allk])))
    except Exception as e:
        if rank == 0:
            q.put(("err", repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_single():
    name, world = "synth_n2085_l64", 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == "ok", res
    _, (indptr, idx, w), kidx, kw = res
    g = load_golden(name)
    assert np.array_equal(indptr, g["eps2_indptr"]) and np.array_equal(idx, g["eps2_indices"]) and np.array_equal(w, g["eps2_weights"])
    assert np.array_equal(kidx, g["knn16_idx"]) and np.array_equal(kw, g["knn16_w"])
