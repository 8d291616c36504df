"""
Row-block sharding of the N x N pair space over the GPUs of one node.

The reference has no multi-GPU code; its only parallel structure is that every batch of rows
is independent (prograph/prograph.py:731-732, :756-760).  That independence is the shard
boundary here: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI),
rank r owns a contiguous block of rows, ONE all-gather of the byte-token shards up front
gives every GPU the full (N, L) matrix (64 MB at N = 1M, L = 64), and after that no rank
talks to another: each computes `rows_local x N` pairs and keeps its CSR / kNN slice with
GLOBAL column indices.  Concatenating the slices in rank order is the single-GPU result.

Everything here is host logic.  The all-gather itself is a C-ABI call (`pg_allgather_tokens`: RCCL
bound inside libprograph_hip.so); `torch.distributed` only carries the 128-byte communicator id
between the ranks once (any backend) and serves the CPU rehearsal of the partitioning (gloo, tensors
on the host: tests/test_sharded_gloo.py).  The compute entry points are the same C-ABI calls as the
single-GPU path (prograph_amd/_native.py).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _native
from .graph import CSRGraph, KNNGraph


def row_block(n, world, rank):
    """Contiguous block of rank `rank`: [r*ceil(n/world), min(n, (r+1)*ceil(n/world)))."""
    per = -(-int(n) // int(world))
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard_rows(n, world):
    return [row_block(n, world, r) for r in range(world)]


_COMMS = {}     # (world, rank, device index) -> RCCL communicator handle of this process


def _comm_init_guarded(world, rank, dev, id_bytes, timeout_s):
    """`pg_comm_init` (ncclCommInitRank: a collective - it returns only when every rank has entered it) under a watchdog:
    a peer that died before entering would leave this rank blocked inside RCCL's bootstrap for good, so after
    `timeout_s` the process reports and exits non-zero (no re-exec, nothing to unwind: the thread is stuck in RCCL)."""
    import os
    import sys
    import threading
    box = {}

    def run():
        try:
            with torch.cuda.device(dev):
                box["comm"] = _native.comm_init(world, rank, id_bytes)
        except Exception as e:          # noqa: BLE001 - handed to the caller
            box["err"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        print(f"[prograph_amd] rank {rank}: pg_comm_init did not return within {timeout_s:.0f} s "
              "(a peer rank never entered it?); giving up", file=sys.stderr, flush=True)
        os._exit(3)
    if "err" in box:
        raise box["err"]
    return box["comm"]


def _comm(world, rank, dev, group=None):
    """This rank's RCCL communicator (created once): rank 0 makes the id, torch.distributed's
    object broadcast carries the 128 bytes - the only use of torch.distributed on the GPU data path.
    Returns False when the ranks agreed to gather through torch.distributed instead.
    Every step that can fail on ONE rank only is followed by an agreement of all ranks before anybody enters
    something collective: (1) can RCCL be bound at all (local check), (2) did rank 0 obtain an id, (3) did
    ncclCommInitRank succeed everywhere.  The collective init itself runs under a timeout (PG_COMM_TIMEOUT, 120 s)."""
    import os
    import sys
    key = (world, rank, dev.index)
    if key in _COMMS:
        return _COMMS[key]

    def agree(flag):                    # True only when `flag` holds on every rank
        if world == 1:
            return bool(flag)
        ok = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        return int(ok.item()) == 1

    def fall_back(why):
        print(f"[prograph_amd] rank {rank}: {why}; the shards are gathered with torch.distributed's RCCL all-gather instead",
              file=sys.stderr, flush=True)
        _COMMS[key] = False
        return False

    if not agree(_native.comm_available()):
        if world == 1:
            _native.comm_unique_id()    # one rank, nothing to fall back to: raise the library's own error
        return fall_back("librccl.so cannot be bound on at least one rank")
    box, err = [None], None
    if rank == 0:
        try:
            box[0] = _native.comm_unique_id()
        except Exception as e:          # noqa: BLE001 - reported below, never silent
            err = e
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    if box[0] is None:
        if world == 1:
            raise err
        return fall_back(f"rank 0 could not create the communicator id ({err!r})")
    comm = None
    try:
        comm = _comm_init_guarded(world, rank, dev, box[0], float(os.environ.get("PG_COMM_TIMEOUT", "120")))
    except Exception as e:              # noqa: BLE001
        err = e
    if not agree(comm is not None):
        if comm is not None:
            _native.comm_destroy(comm)
        if world == 1:
            raise err
        return fall_back(f"pg_comm_init failed on at least one rank ({err!r})")
    _COMMS[key] = comm
    return comm


def allgather_tokens(local_tokens, n_total, group=None):
    """
    Gather the (rows_r, L) uint8 shards of all ranks into the full (n_total, L) matrix on every
    rank with ONE all-gather (the last rank's block may be short: shards are zero-padded to
    ceil(n/world) rows for the collective and the tail is dropped afterwards).  Device shards go
    through `pg_allgather_tokens` (RCCL over xGMI inside the C-ABI library); host tensors (the CPU
    rehearsal) and the several-ranks-on-one-GPU rehearsal (PG_DIST_BACKEND=gloo) use torch's gloo.
    """
    if not dist.is_initialized():
        return local_tokens[:n_total]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-int(n_total) // world)
    L = local_tokens.shape[1]
    if local_tokens.shape[0] != per:
        padded = torch.zeros((per, L), dtype=local_tokens.dtype, device=local_tokens.device)
        padded[: local_tokens.shape[0]] = local_tokens
        local_tokens = padded
    if local_tokens.is_cuda and dist.get_backend(group) != "gloo":
        # (the C-ABI collective moves bytes: other token dtypes go through torch's all-gather of the same library)
        comm = _comm(world, rank, local_tokens.device, group) if local_tokens.dtype == torch.uint8 else False
        if comm is not False:
            return _native.allgather_tokens(comm, local_tokens, world)[:n_total]
        # the C-ABI communicator could not be created (reported on stderr): the same collective through torch's RCCL
        full = torch.empty((per * world, L), dtype=local_tokens.dtype, device=local_tokens.device)
        dist.all_gather_into_tensor(full, local_tokens.contiguous(), group=group)
        return full[:n_total]
    full = torch.empty((per * world, L), dtype=local_tokens.dtype, device=local_tokens.device)
    if local_tokens.is_cuda:
        # rehearsal mode (several ranks sharing one GPU, CPU collectives): stage through the host
        host = torch.empty((per * world, L), dtype=local_tokens.dtype)
        dist.all_gather_into_tensor(host, local_tokens.cpu().contiguous(), group=group)
        full.copy_(host)
    else:
        dist.all_gather_into_tensor(full, local_tokens.contiguous(), group=group)
    return full[:n_total]


def build_graph_sharded(local_tokens, n_total, eps=None, k=None, comp_code=_native.CMP_LE, cap=256,
                        bits=None, group=None):
    """
    This rank's slice of the epsilon / kNN graph of the full matrix.
      local_tokens  (rows_r, L) uint8 tensor on this rank's GPU: rows row_block(n_total, world, rank)
    Returns CSRGraph / KNNGraph with `row0` = first global row of the slice.
    """
    if bool(eps) == bool(k):
        raise ValueError("Epsilon or K must be provided, but both cannot be as they are different methods of graph construction.")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = row_block(n_total, world, rank)
    full = allgather_tokens(local_tokens, n_total, group=group)
    planes = _native.pack(full, bits=bits)       # bits=None: 5 planes when every token is <= 31
    if hi <= lo:
        return None
    if eps:
        indptr, indices, wts = _native.eps_graph(planes, planes, comp_code, eps, row0=lo, nrows=hi - lo, cap=cap)
        return CSRGraph(indptr, indices, wts, n_total, row0=lo)
    idx, d = _native.knn_graph(planes, planes, k, row0=lo, nrows=hi - lo)
    return KNNGraph(idx, d, n_total, row0=lo)


def gather_csr_to_host(graph, group=None):
    """Optional: concatenate every rank's CSR slice on rank 0 (host side, object gather)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    part = None if graph is None else graph.host()
    if world == 1:
        return part
    parts = [None] * world
    dist.all_gather_object(parts, part, group=group)
    parts = [p for p in parts if p is not None]
    indptr = [parts[0][0]]
    for p in parts[1:]:
        indptr.append(p[0][1:] + indptr[-1][-1])
    return np.concatenate(indptr), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])
