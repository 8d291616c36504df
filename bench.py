#!/usr/bin/env python3
"""
bench.py — the hot path's headline benchmark (BASELINE.json: "sequence-pairs/s (and HBM GB/s vs
roofline) for N x N Hamming + adjacency build").

  python bench.py --gpus 1 --steps K --warmup W            (default: configs[2], N=200k L=64 kNN16)
  python -m torch.distributed.run --nproc-per-node G ... bench.py --gpus G --steps K --warmup W

A step = one pass of the hot path over one synthetic token matrix that is already resident in
HBM as row-major uint8 (SURVEY.md §8-d generator): [all-gather of the row shards when G > 1] ->
plane packing -> the fused all-pairs kernel (Hamming + kNN selection, or Hamming + epsilon
slots + scan + CSR compaction).  Prints ONE JSON line on rank 0.

Workloads (--workload):
  cfg3  N=200 000, L=64, kNN k=16        the configuration the target is quoted on (default, G=1)
  cfg2  N= 50 000, L=32, eps d<=2, full CSR
  cfg3w the weak-scaling series anchored at cfg3 (default for G > 1): a square N_G x N_G problem with
        N_G = 200 000 * sqrt(G), row-block sharded over G GPUs, so every GPU always evaluates 4.0e10
        ordered pairs (G=1 is cfg3 itself, G=8 is N=565 688); the all-gather and the pack are in the step
  cfg4  N=1 000 000, L=64, kNN k=16, row-block sharded: every GPU computes N/8 rows x N columns
        (BASELINE.json configs[3]; G GPUs cover G/8 of the rows, G=8 is the full graph)
  cfg5  N=200 000, variable length 96..128, banded Levenshtein (band 8), kNN k=8 — build defined
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    "cfg2": dict(N=50_000, L=32, mode="eps", eps=2, k=None, shards=1),
    "cfg3": dict(N=200_000, L=64, mode="knn", eps=None, k=16, shards=1),
    "cfg3w": dict(N=200_000, L=64, mode="knn", eps=None, k=16, shards=1),     # N is scaled by sqrt(G) in main()
    "cfg4": dict(N=1_000_000, L=64, mode="knn", eps=None, k=16, shards=8),
    # build-defined (no reference counterpart, parity unpinned): variable length 96..128, band 8
    "cfg5": dict(N=200_000, L=128, mode="lev", eps=None, k=8, shards=1, band=8),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto"] + list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU budget of the baseline sample")
    return ap.parse_args()


def cpu_baseline(tok_host, wl, budget_s):
    """
    The oracle's torch-CPU restatement of the reference algorithm (batches of 8 rows -> broadcast
    != -> sum -> where / full sort; oracle/prograph_oracle.py, pinned to the reference's golden
    vectors), timed on this box's host cores on a bounded sample: the first R rows against all N
    columns.  R is calibrated from a 64-row pilot to about `budget_s` seconds.
    """
    from oracle import prograph_oracle as O
    N = tok_host.shape[0]
    t64 = tok_host.astype(np.int64)
    kw = dict(eps=wl["eps"]) if wl["mode"] == "eps" else dict(k=wl["k"])
    t0 = time.perf_counter()
    if wl["mode"] == "lev":
        return cpu_baseline_lev(tok_host, wl, budget_s)
    O.build_graph(t64, row_limit=64, **kw)
    pilot = time.perf_counter() - t0
    R = int(max(64, min(4096, (budget_s / max(pilot, 1e-6)) * 64)))
    R = min((R // 8) * 8, (N // 8) * 8)
    t0 = time.perf_counter()
    O.build_graph(t64, row_limit=R, **kw)
    dt = time.perf_counter() - t0
    return {"value": R * N / dt, "unit": "sequence-pairs/s", "cores": int(torch.get_num_threads()),
            "kind": "port", "host_cpus": os.cpu_count(), "seconds": round(dt, 2),
            "sample": f"first {R} rows x all {N} columns of the same token matrix, batch_size=8, "
                      f"{'eps<=%d where/gather' % wl['eps'] if wl['mode'] == 'eps' else 'stable sort, k=%d' % wl['k']}"
                      f" (torch CPU, fp16 staging like prograph.py:726)"}


def cpu_baseline_lev(tok_host, wl, budget_s):
    """No reference code exists for Levenshtein: the baseline is the oracle's C banded
    Wagner-Fischer + canonical selection (oracle/oracle.c, OpenMP) on a bounded row sample."""
    from oracle import c_oracle as C
    N = tok_host.shape[0]
    t0 = time.perf_counter()
    C.lev_knn(tok_host, wl["k"], band=wl["band"], row0=0, nrows=8)
    pilot = time.perf_counter() - t0
    R = int(max(8, min(2048, (budget_s / max(pilot, 1e-6)) * 8)))
    t0 = time.perf_counter()
    C.lev_knn(tok_host, wl["k"], band=wl["band"], row0=0, nrows=R)
    dt = time.perf_counter() - t0
    return {"value": R * N / dt, "unit": "sequence-pairs/s", "cores": os.cpu_count(), "kind": "port",
            "host_cpus": os.cpu_count(), "seconds": round(dt, 2),
            "sample": f"first {R} rows x all {N} columns, banded Wagner-Fischer (band {wl['band']}) + (d,idx) top-{wl['k']}, C/OpenMP oracle; "
                      "build-defined workload, no reference implementation exists"}


def main():
    a = parse()
    # stdout must carry exactly ONE JSON line: park fd 1 on stderr while libraries (RCCL prints a
    # version banner to stdout at communicator creation) run, restore it for the final print
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    G = a.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if G != world:
        if world == 1 and G > 1:
            raise SystemExit("--gpus > 1 must be launched with torch.distributed.run (one rank per GPU)")
        G = world
    from prograph_amd import _native, synth, sharded
    import torch.distributed as dist

    # PG_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("PG_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = G > 1 or os.environ.get("PG_FORCE_DIST") == "1"      # PG_FORCE_DIST: rehearse the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    _native.lib()

    name = a.workload if a.workload != "auto" else ("cfg3" if G == 1 else "cfg3w")
    wl = dict(WORKLOADS[name])
    if name == "cfg3w":
        # weak scaling: N_G^2 / G = 200000^2 pairs per GPU, N_G a multiple of 8*G (equal row blocks)
        unit = 8 * G
        wl["N"] = int(-(-int(round(200_000 * (G ** 0.5))) // unit) * unit)
    N, L = wl["N"], wl["L"]
    if wl["shards"] > 1:
        per = N // wl["shards"]                 # rows per GPU, fixed (weak scaling)
        lo, hi = rank * per, (rank + 1) * per
    else:
        lo, hi = sharded.row_block(N, G, rank)
    rows_local = hi - lo

    # synthetic input, resident in HBM before the timed region.  With G > 1 every rank owns its
    # row shard and the full matrix is all-gathered inside the step (the path's one collective).
    if wl["mode"] == "lev":
        if G != 1 or use_dist:
            raise SystemExit("cfg5 is a single-GPU workload")
        tok_host, _ = synth.clustered_varlen_tokens(N, Lmax=L, Lmin=96)
        tok_dev = torch.from_numpy(tok_host).to(dev)
        shard_dev = None
    elif not use_dist:
        tok_host = synth.clustered_tokens(N, L)
        tok_dev = torch.from_numpy(tok_host).to(dev)
        shard_dev = None
    else:
        tok_host = None
        glo, ghi = sharded.row_block(N, G, rank)
        shard_dev = torch.from_numpy(synth.clustered_tokens(N, L, row0=glo, nrows=ghi - glo)).to(dev)

    k = wl["k"] or 1
    cap = 256
    if wl["mode"] == "lev":
        pass
    elif wl["mode"] == "eps":
        slot_idx = torch.empty(rows_local * cap, dtype=torch.int32, device=dev)
        slot_w = torch.empty(rows_local * cap, dtype=torch.uint8, device=dev)
        counts = torch.empty(rows_local, dtype=torch.int32, device=dev)
    else:
        out = (torch.empty((rows_local, k), dtype=torch.int32, device=dev),
               torch.empty((rows_local, k), dtype=torch.uint8, device=dev))
    kern_ev = []
    result = {}

    def step_lev(record):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        idx, d, st = _native.levenshtein_knn(tok_dev, k, band=wl["band"], cap=512, return_stats=True)
        e1.record()
        result.update(st)
        if record:
            kern_ev.append((e0, e1))

    def step(record):
        if wl["mode"] == "lev":
            return step_lev(record)
        full = tok_dev if not use_dist else sharded.allgather_tokens(shard_dev, N)
        planes = _native.pack(full, bits=5)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        if wl["mode"] == "eps" and lo == 0 and rows_local == N and os.environ.get("PG_EPS_SYM", "auto") != "0":
            # the whole square graph on one GPU: the symmetric path (what Prograph.build_graph takes)
            L_ = _native.lib()
            counts_lo = torch.empty(rows_local, dtype=torch.int32, device=dev)
            sargs = (_native._ptr(planes.buf), planes.npad, planes.n, planes.g * 32, planes.bits, _native.CMP_LE,
                     float(wl["eps"]), cap, _native._ptr(slot_idx), _native._ptr(slot_w), _native._ptr(counts),
                     _native._ptr(counts_lo))
            _native._check(L_.pg_eps_slots_sym(*sargs, _native._stream()), "pg_eps_slots_sym")
            e1.record()
            total = counts + counts_lo
            indptr = torch.empty(rows_local + 1, dtype=torch.int64, device=dev)
            scratch = torch.empty(int(L_.pg_scan_scratch_bytes(rows_local)), dtype=torch.uint8, device=dev)
            _native._check(L_.pg_exclusive_scan(_native._ptr(total), rows_local, _native._ptr(indptr),
                                                _native._ptr(scratch), _native._stream()), "scan")
            nnz = int(indptr[-1].item())
            indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
            weights = torch.empty(max(nnz, 1), dtype=torch.uint8, device=dev)
            _native._check(L_.pg_eps_compact_sym(*sargs, _native._ptr(indptr), _native._ptr(indices),
                                                 _native._ptr(weights), _native._stream()), "compact_sym")
            result["nnz"] = nnz
            result["path"] = "symmetric (every unordered pair once)"
        elif wl["mode"] == "eps":
            _native.eps_slots_only(planes, planes, _native.CMP_LE, wl["eps"], lo, rows_local, cap, slot_idx, slot_w, counts)
            e1.record()
            indptr = torch.empty(rows_local + 1, dtype=torch.int64, device=dev)
            scratch = torch.empty(int(_native.lib().pg_scan_scratch_bytes(rows_local)), dtype=torch.uint8, device=dev)
            L_ = _native.lib()
            _native._check(L_.pg_exclusive_scan(_native._ptr(counts), rows_local, _native._ptr(indptr),
                                                _native._ptr(scratch), _native._stream()), "scan")
            nnz = int(indptr[-1].item())
            indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
            weights = torch.empty(max(nnz, 1), dtype=torch.uint8, device=dev)
            args = (_native._ptr(planes.buf), planes.npad, lo, rows_local, _native._ptr(planes.buf), planes.npad, planes.n,
                    planes.g * 32, planes.bits, _native.CMP_LE, float(wl["eps"]), cap)
            _native._check(L_.pg_eps_compact(*args, _native._ptr(slot_idx), _native._ptr(slot_w), _native._ptr(counts),
                                             _native._ptr(indptr), _native._ptr(indices), _native._ptr(weights),
                                             _native._stream()), "compact")
            result["nnz"] = nnz
        else:
            _native.knn_graph(planes, planes, k, row0=lo, nrows=rows_local, out=out)
            e1.record()
        if record:
            kern_ev.append((e0, e1))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # PCIe-inclusive rate (never `value`): host numpy tokens -> HBM, one step, results -> host
    pcie_ms = None
    if not use_dist:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tok_dev = torch.from_numpy(tok_host).to(dev)
        step(False)
        if wl["mode"] == "knn":
            _ = out[0].cpu(), out[1].cpu()
        torch.cuda.synchronize()
        pcie_ms = (time.perf_counter() - t1) * 1e3

    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in kern_ev]))
    total_pairs = float(rows_local) * N * G                      # every rank does rows_local x N
    ms_per_step = elapsed / a.steps * 1e3
    value = total_pairs * a.steps / elapsed

    if rank == 0:
        out_bytes = 5 * k * rows_local if wl["mode"] in ("knn", "lev") else 8 * (rows_local + 1) + 5 * result.get("nnz", 0)
        alg_bytes = float(rows_local) * N * L + rows_local * L + out_bytes      # SURVEY.md §8-d, per launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # HBM traffic and VALU issue share are PMC measurements of this same command, collected in
        # separate rocprofv3 passes (tools/profile.sh) and committed under profiles/
        traffic = valu_frac = None
        pmc = os.path.join(REPO, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(name, {})
                traffic, valu_frac = rec.get("hbm_bytes_per_launch"), rec.get("valu_issue_frac")
            except Exception:
                traffic = valu_frac = None
        line = {
            "metric": "sequence-pairs/s", "value": value, "unit": "sequence-pairs/s", "n_gpus": G,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{name}: N={N} L={L} {'Levenshtein' if wl['mode'] == 'lev' else 'Hamming'}, " +
                                   (f"kNN k={k}" if wl["mode"] == "knn" else
                                    f"banded Levenshtein band={wl.get('band')} kNN k={k} (build defined, parity unpinned)" if wl["mode"] == "lev"
                                    else f"eps d<={wl['eps']} full CSR") +
                                   (f", row-block sharded, {rows_local} rows/GPU x {N} columns, RCCL all-gather in step" if G > 1 or wl["shards"] > 1 else ""),
                       "N": N, "L": L, "rows_per_gpu": rows_local, "alphabet": "5-bit tokens 1..20",
                       "parallelism": f"rowblock{G}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "pg_nsq_kernel", "kernel_ms": kern_ms, "algorithmic_bytes": alg_bytes,
                         "note": "algorithmic bytes = L per ordered pair (SURVEY.md 8-d); the operand matrix is "
                                 "cache resident so this HBM-equivalent rate is not capped at 1; the kernel is "
                                 "VALU-issue bound: valu_frac = SQ_INSTS_VALU / time / (256 CUs x 4 SIMDs x 2.4 GHz / 4), "
                                 "measured with rocprofv3 (profiles/pmc_summary.json)",
                         "valu_frac": valu_frac},
        }
        if wl["mode"] == "eps":
            line["config"]["nnz"] = result.get("nnz")
            if result.get("path"):
                line["config"]["path"] = result["path"]
        if wl["mode"] == "lev":
            line["config"].update({"candidates": result.get("candidates"), "filter_passes": result.get("filter_passes")})
            line["roofline"]["kernel"] = "pg_lev_* (profile + bag filter pg_nsq_kernel<BagMetric> + pg_lev_select_kernel)"
            line["roofline"]["note"] = ("kernel_ms spans the three Levenshtein launches; algorithmic bytes = L per ordered pair as for "
                                        "Hamming; the path is VALU bound (10 ops/pair filter + 4 ops/DP cell on candidates)")
            line["roofline"]["valu_frac"] = None
        if pcie_ms is not None:
            line["pcie_inclusive"] = {"ms_per_step": pcie_ms, "value": float(rows_local) * N / (pcie_ms * 1e-3),
                                      "note": "host tokens H2D + step + results D2H (pageable memory); reported only, not `value`"}
        if not use_dist and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tok_host, wl, a.cpu_seconds)
        else:
            line["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
