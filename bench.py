#!/usr/bin/env python3
"""
bench.py — the hot path's headline benchmark (BASELINE.json: "sequence-pairs/s (and HBM GB/s vs
roofline) for N x N Hamming + adjacency build").

  python bench.py --gpus 1 --steps K --warmup W            (default: configs[2], N=200k L=64 kNN16)
  python -m torch.distributed.run --nproc-per-node G ... bench.py --gpus G --steps K --warmup W
                                                           (default for G > 1: configs[3], N=1M row-block sharded)

(The timed step packs with check=False: the device word that flags a token outside the alphabet is read once after
the timed steps, not per step - `Prograph._byte_planes` pays that host sync on every pack.)
A step = one pass of the hot path over one synthetic token matrix that is already resident in
HBM as row-major uint8 (SURVEY.md §8-d generator): [all-gather of the row shards when G > 1] ->
plane packing -> the fused all-pairs kernel (Hamming + kNN selection, or Hamming + epsilon
slots + scan + CSR compaction).  Prints ONE JSON line on rank 0.

Workloads (--workload):
  cfg3  N=200 000, L=64, kNN k=16        the configuration the target is quoted on (default, G=1)
  cfg2  N= 50 000, L=32, eps d<=2, full CSR
  cfg4  N=1 000 000, L=64, kNN k=16, row-block sharded: every GPU computes N/8 = 125 000 rows x N columns
        (BASELINE.json configs[3]; default for G > 1; G GPUs cover G/8 of the rows, G=8 is the full graph;
        the all-gather and the pack are inside the step; PG_FORCE_DIST=1 runs the 1-GPU arm through RCCL)
  cfg3w a square weak-scaling series anchored at cfg3: N_G = 200 000 * sqrt(G), row-block sharded
  cfg5  N=200 000, variable length 96..128, banded Levenshtein (band 8), kNN k=8 — build defined
  cfg3d cfg3's shape on DENSE data (one cluster: a mutant library around one seed, every pair within 6)

  cfg3b8 cfg3's shape packed with 8 bit planes (any byte alphabet: the engine's HammingMetric<2,8> instances)
  mink64 / mink1280  N=50 000 fp16 embeddings of D = 64 / 1280, Minkowski p=2 kNN k=16

With one GPU and the default workload the line also carries `extra`: short runs of cfg2, cfg3d, cfg5, cfg3b8,
mink64 and mink1280 in the same invocation (their own ms_per_step / kernel_ms / value), mirrored - workload,
ms_per_step, kernel_ms, frac - in `roofline.other_workloads` (the driver keeps top-level keys only).

roofline: the all-pairs kernel is bound by vector-instruction issue, not by HBM (the operand matrix is
cache resident): `frac` = wave-instructions per second / the SIMD-32 issue peak, with the per-launch
instruction count taken from the committed rocprofv3 PMC pass of the SAME kernel sources (sha
checked, null otherwise) and the kernel time measured in THIS run.  The HBM-equivalent streaming rate
of SURVEY.md §8-d (L bytes per ordered pair) is reported under `hbm_equivalent`, never as `frac`.
"""
import argparse
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CUS, SIMDS, CLOCK_HZ = 256, 4, 2.4e9
# MI355X_MICROARCH.md "Wave scheduling": a wave64 VALU instruction issues over 2 cycles on a SIMD-32
VALU_PEAK = CUS * SIMDS * CLOCK_HZ / 2.0          # 1.2288e12 wave-instructions/s
# measured mixed-stream ceiling of this kernel's instruction classes (v_bcnt / v_cmp / v_or3 at ~4.2 cycles,
# logic ops at ~2.3, profiles/r01_valu_issue_microbench.txt): reported beside the guide's peak
VALU_MIX_CEILING = CUS * SIMDS * CLOCK_HZ / 4.2
MFMA_I8_CYCLES = 32.0                              # v_mfma_f32_32x32x64_f8f6f4 (FP4) = the int8 32x32x32 form: cycles per instruction per SIMD

WORKLOADS = {
    "cfg2": dict(N=50_000, L=32, mode="eps", eps=2, k=None, shards=1),
    "cfg3": dict(N=200_000, L=64, mode="knn", eps=None, k=16, shards=1),
    "cfg3d": dict(N=200_000, L=64, mode="knn", eps=None, k=16, shards=1, dense=True),
    "cfg3w": dict(N=200_000, L=64, mode="knn", eps=None, k=16, shards=1),     # N is scaled by sqrt(G)
    "cfg4": dict(N=1_000_000, L=64, mode="knn", eps=None, k=16, shards=8),
    # build-defined (no reference counterpart, parity unpinned): variable length 96..128, band 8
    "cfg5": dict(N=200_000, L=128, mode="lev", eps=None, k=8, shards=1, band=8),
    "cfg3b8": dict(N=200_000, L=64, mode="knn", eps=None, k=16, shards=1, bits=8),
    # Minkowski p = 2 on fp16 embeddings: L = embedding dimension
    "mink64": dict(N=50_000, L=64, mode="mink", eps=None, k=16, shards=1),
    "mink1280": dict(N=50_000, L=1280, mode="mink", eps=None, k=16, shards=1),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto"] + list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the sub-records (cfg2, cfg3d, cfg5, cfg3b8, mink64, mink1280)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU budget of the baseline sample")
    return ap.parse_args()


def kernel_src_sha():
    """sha of the kernel sources: PMC-derived numbers in profiles/pmc_summary.json are only quoted
    for the build they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "prograph_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


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


class Ctx:
    pass


def run_workload(name, G, rank, dev, use_dist, backend, steps, warmup, want_pcie):
    """Times `steps` steps of one workload; returns (record dict for rank 0, host tokens or None, wl)."""
    from prograph_amd import _native, synth, sharded
    import torch.distributed as dist

    wl = dict(WORKLOADS[name])
    if name == "cfg3w":
        unit = 8 * G                      # N_G^2 / G = 200000^2 pairs per GPU, N_G a multiple of 8*G
        wl["N"] = int(-(-int(round(200_000 * (G ** 0.5))) // unit) * unit)
    N, L = wl["N"], wl["L"]
    if wl["shards"] > 1:
        per = N // wl["shards"]                 # rows per GPU, fixed (weak scaling)
        lo, hi = rank * per, (rank + 1) * per
    else:
        lo, hi = sharded.row_block(N, G, rank)
    rows_local = hi - lo
    members = N if wl.get("dense") else 256

    # synthetic input, resident in HBM before the timed region.  With G > 1 every rank owns its
    # row shard and the full matrix is all-gathered inside the step (the path's one collective).
    if wl["mode"] == "mink":
        if G != 1 or use_dist:
            raise SystemExit("the Minkowski workloads are single-GPU")
        return run_minkowski(name, wl, dev, steps, warmup)
    if wl["mode"] == "lev":
        if G != 1 or use_dist:
            raise SystemExit("cfg5 is a single-GPU workload")
        tok_host, _ = synth.clustered_varlen_tokens(N, Lmax=L, Lmin=96)
        tok_dev = torch.from_numpy(tok_host).to(dev)
        shard_dev = None
    elif not use_dist:
        tok_host = synth.clustered_tokens(N, L, members=members)
        tok_dev = torch.from_numpy(tok_host).to(dev)
        shard_dev = None
    else:
        tok_host = None
        tok_dev = None
        glo, ghi = sharded.row_block(N, G, rank)
        shard_dev = torch.from_numpy(synth.clustered_tokens(N, L, row0=glo, nrows=ghi - glo, members=members)).to(dev)

    k = wl["k"] or 1
    cap = 256
    L_ = _native.lib()
    if wl["mode"] == "eps":
        slot_idx = torch.empty(rows_local * cap, dtype=torch.int32, device=dev)
        slot_w = torch.empty(rows_local * cap, dtype=torch.uint8, device=dev)
        counts = torch.empty(rows_local, dtype=torch.int32, device=dev)
    elif wl["mode"] == "knn":
        out = (torch.empty((rows_local, k), dtype=torch.int32, device=dev),
               torch.empty((rows_local, k), dtype=torch.uint8, device=dev))
    kern_ev = []
    result = {}
    c = Ctx()
    c.tok_dev = tok_dev
    c.flags = []

    def step(record):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        if wl["mode"] == "lev":
            e0.record()
            idx, d, st = _native.levenshtein_knn(c.tok_dev, k, band=wl["band"], cap=512, return_stats=True)
            e1.record()
            result.update(st)
            if record:
                kern_ev.append((e0, e1))
            return
        full = c.tok_dev if not use_dist else sharded.allgather_tokens(shard_dev, N)
        planes = _native.pack(full, bits=wl.get("bits", 5), check=False)   # validity word read once after the timed steps
        c.flags.append(planes.flags)
        e0.record()
        if wl["mode"] == "eps" and lo == 0 and rows_local == N and os.environ.get("PG_EPS_SYM", "auto") != "0":
            # the whole square graph on one GPU: the symmetric path (what Prograph.build_graph takes)
            counts_lo = torch.empty(rows_local, dtype=torch.int32, device=dev)
            sargs = (_native._ptr(planes.buf), planes.npad, planes.n, planes.g * 32, planes.bits, _native.CMP_LE,
                     float(wl["eps"]), cap, _native._ptr(slot_idx), _native._ptr(slot_w), _native._ptr(counts),
                     _native._ptr(counts_lo))
            _native._check(L_.pg_eps_slots_sym(*sargs, _native._ptr(_native.workspace(rows_local, dev)), _native._stream()),
                           "pg_eps_slots_sym")
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
                                                 _native._ptr(weights), 0, _native._stream()), "compact_sym")
            result["nnz"] = nnz
            result["path"] = "symmetric (every unordered pair once)"
        elif wl["mode"] == "eps":
            _native.eps_slots_only(planes, planes, _native.CMP_LE, wl["eps"], lo, rows_local, cap, slot_idx, slot_w, counts)
            e1.record()
            indptr = torch.empty(rows_local + 1, dtype=torch.int64, device=dev)
            scratch = torch.empty(int(L_.pg_scan_scratch_bytes(rows_local)), dtype=torch.uint8, device=dev)
            _native._check(L_.pg_exclusive_scan(_native._ptr(counts), rows_local, _native._ptr(indptr),
                                                _native._ptr(scratch), _native._stream()), "scan")
            nnz = int(indptr[-1].item())
            indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
            weights = torch.empty(max(nnz, 1), dtype=torch.uint8, device=dev)
            args = (_native._ptr(planes.buf), planes.npad, lo, rows_local, _native._ptr(planes.buf), planes.npad, planes.n,
                    planes.g * 32, planes.bits, _native.CMP_LE, float(wl["eps"]), cap)
            _native._check(L_.pg_eps_compact(*args, _native._ptr(slot_idx), _native._ptr(slot_w), _native._ptr(counts),
                                             _native._ptr(indptr), _native._ptr(indices), _native._ptr(weights), 0,
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

    for _ in range(warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # PCIe-inclusive rate (never `value`): host numpy tokens -> HBM, one step, results -> host
    pcie_ms = None
    if want_pcie and not use_dist:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        c.tok_dev = torch.from_numpy(tok_host).to(dev)
        step(False)
        if wl["mode"] == "knn":
            _ = out[0].cpu(), out[1].cpu()
        torch.cuda.synchronize()
        pcie_ms = (time.perf_counter() - t1) * 1e3

    if c.flags and int(torch.stack(c.flags).max().item()):
        raise SystemExit("pg_pack_planes flagged a token outside the 5-bit alphabet")
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in kern_ev]))
    ms_per_step = elapsed / steps * 1e3
    value = float(rows_local) * N * G * steps / elapsed         # every rank does rows_local x N
    thr_default = "20000" if wl["mode"] == "knn" else ("36000" if L <= 32 else "28000")     # pg_api.hip: use_mm_engine
    engine = "mfma" if rows_local >= int(os.environ.get("PG_ENGINE_MIN_ROWS", thr_default)) else "valu"
    engine = os.environ.get("PG_ENGINE", engine)
    if wl["mode"] == "lev":
        engine = "valu"                       # the bag filter runs on pg_nsq_kernel<BagMetric> (SAD is not bilinear)
    rec = {"name": name, "N": N, "L": L, "k": k, "rows_local": rows_local, "kern_ms": kern_ms, "ms_per_step": ms_per_step,
           "value": value, "pcie_ms": pcie_ms, "result": result, "engine": engine}
    return rec, tok_host, wl


def run_minkowski(name, wl, dev, steps, warmup):
    """kNN graph of N fp16 embeddings under Minkowski p = 2 with the reference's fp16 rounding: per block of rows the
    distance block (pg_minkowski_dense) and the canonical ranks 1..k (pg_f16_knn), as Prograph._build_graph_minkowski
    walks them.  Kernel time = the HIP-event time of all launches of a step."""
    from prograph_amd import _native
    N, D, k = wl["N"], wl["L"], wl["k"]
    g = torch.Generator(device="cpu").manual_seed(20260104)
    X = torch.randn((N, D), generator=g, dtype=torch.float32).to(torch.float16).to(dev)
    rows_per_block = max(64, min(N, (1 << 27) // N))
    kern_ev = []

    def step(record):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        xp = _native.pack_f16(X)
        last = None
        for r0 in range(0, N, rows_per_block):
            yp = _native.pack_f16(X[r0:r0 + rows_per_block])
            block = _native.minkowski_dense(xp, yp)
            last = _native.f16_knn(block, k, first=1)
        e1.record()
        if record:
            kern_ev.append((e0, e1))
        return last

    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in kern_ev]))
    rec = {"name": name, "N": N, "L": D, "k": k, "rows_local": N, "kern_ms": kern_ms, "ms_per_step": elapsed / steps * 1e3,
           "value": float(N) * N * steps / elapsed, "pcie_ms": None, "result": {}, "engine": "valu (pg_minkowski_dense)"}
    return rec, None, wl


def workload_text(rec, wl, G):
    if wl["mode"] == "mink":
        return f"{rec['name']}: N={rec['N']} fp16 embeddings D={rec['L']}, Minkowski p=2 (fp16 rounding of the reference), kNN k={rec['k']}"
    N, L, k = rec["N"], rec["L"], rec["k"]
    kind = (f"kNN k={k}" if wl["mode"] == "knn" else
            f"banded Levenshtein band={wl.get('band')} kNN k={k} (build defined, parity unpinned)" if wl["mode"] == "lev"
            else f"eps d<={wl['eps']} full CSR")
    data = ", dense data (one cluster)" if wl.get("dense") else ""
    if wl.get("bits") == 8:
        data += ", packed with 8 bit planes (byte alphabet)"
    shard = (f", row-block sharded, {rec['rows_local']} rows/GPU x {N} columns, RCCL all-gather in step"
             if G > 1 or wl["shards"] > 1 else "")
    return f"{rec['name']}: N={N} L={L} {'Levenshtein' if wl['mode'] == 'lev' else 'Hamming'}, {kind}{data}{shard}"


def roofline(rec, wl, pmc, sha):
    """VALU-issue roofline of the dominant kernel, instruction counts from the committed PMC pass of
    the same sources, time from this run."""
    k, rows_local, N, L = rec["k"], rec["rows_local"], rec["N"], rec["L"]
    if wl["mode"] == "mink":
        # VALU bound by construction (csrc/pg_mink.hip): 3 vector instructions per 2 elements per pair (v_pk_add_f16,
        # v_pk_mul_f16, v_dot2c_f32_f16); the algorithmic count, no counters needed
        valu = 1.5 * L * float(N) * N / 64.0
        achieved = valu / (rec["kern_ms"] * 1e-3)
        alg_bytes = float(N) * N * (2 * L + 2)
        return {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK, "unit": "wave-instructions/s", "frac": achieved / VALU_PEAK,
                "traffic": None, "kernel": "pg_minkowski_dense (+ pg_pack_f16, pg_f16_knn)", "kernel_ms": rec["kern_ms"],
                "valu_wave_instr_per_step": valu,
                "hbm_equivalent": {"achieved_GBs": alg_bytes / (rec["kern_ms"] * 1e-3) / 1e9, "algorithmic_bytes": alg_bytes,
                                   "note": "2*D bytes of fp16 operand per ordered pair + the fp16 distance"},
                "note": "algorithmic instruction count 1.5 * D * N^2 / 64 (not a counter); kernel_ms spans all launches of a step"}
    out_bytes = 5 * k * rows_local if wl["mode"] in ("knn", "lev") else 8 * (rows_local + 1) + 5 * rec["result"].get("nnz", 0)
    alg_bytes = float(rows_local) * N * L + rows_local * L + out_bytes      # SURVEY.md §8-d, per step
    hbm_eq = alg_bytes / (rec["kern_ms"] * 1e-3) / 1e9
    p = pmc.get(rec["name"]) or {}
    fresh = bool(p) and p.get("kernel_src_sha") == sha
    valu = p.get("valu_wave_instr_per_step") if fresh else None
    if wl["mode"] == "lev" and fresh:
        # kernel_ms spans ALL launches of a Levenshtein step, so does the instruction count (profile + bag filter +
        # pack + exact distances / selection, summed per step by tools/summarize_prof.py)
        valu = p.get("valu_wave_instr_per_step_all_kernels")
    mfma = p.get("mfma_instr_per_step") if fresh else None
    achieved = valu / (rec["kern_ms"] * 1e-3) if valu else None
    r = {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK, "unit": "wave-instructions/s",
         "frac": achieved / VALU_PEAK if achieved else None,
         "traffic": p.get("hbm_bytes_per_step") if fresh else None,
         "kernel": p.get("kernel_name") if fresh else ("pg_mm_kernel" if rec["engine"] == "mfma" else "pg_nsq_kernel"),
         "kernel_ms": rec["kern_ms"],
         "valu_wave_instr_per_step": valu,
         "frac_of_measured_mix_ceiling": achieved / VALU_MIX_CEILING if achieved else None,
         "mfma": None,
         "hbm_equivalent": {"achieved_GBs": hbm_eq, "x_hbm_peak": hbm_eq / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes,
                            "note": "SURVEY.md 8-d streaming rate (L bytes per ordered pair); the operand matrix is cache "
                                    "resident, so this is not a fraction of any peak"},
         "pmc_source": p.get("source") if fresh else None,
         "pmc_kernel_src_sha": p.get("kernel_src_sha"), "kernel_src_sha": sha,
         "note": "peak = 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md); "
                 "SQ_INSTS_VALU per step (summed over the dominant kernel's launches of a step: at cfg3 the main launch, the "
                 "column pieces of the rows beyond a full round and their repair launch) is deterministic for a build + "
                 "workload and comes from the committed rocprofv3 PMC pass (null when the kernel sources changed since); "
                 "kernel_ms is this run's HIP-event time around all launches of the call"}
    if mfma:
        r["mfma"] = {"instr_per_step": mfma, "pipe_busy_frac": mfma * MFMA_I8_CYCLES / (CUS * SIMDS * CLOCK_HZ * rec["kern_ms"] * 1e-3),
                     "note": "v_mfma_f32_32x32x64_f8f6f4 with FP4 operands (stage-1 signature filter, K = 54 signature bits + 10 bias slots), 32 cycles per instruction per SIMD at 2.4 GHz"}
    if wl["mode"] == "lev":
        r["kernel"] = "pg_lev_* (profile + bag filter pg_nsq_kernel<BagMetric> + pg_lev_select_kernel)"
        r["note"] = "kernel_ms spans the Levenshtein launches (filter + exact distances + selection); VALU-issue bound; " + r["note"]
    return r


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
    from prograph_amd import _native
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

    # one GPU: the headline configuration; several GPUs: BASELINE.json configs[3] (SURVEY.md §8-d weak scaling)
    name = a.workload if a.workload != "auto" else ("cfg3" if G == 1 else "cfg4")
    rec, tok_host, wl = run_workload(name, G, rank, dev, use_dist, backend, a.steps, a.warmup, want_pcie=True)

    if rank == 0:
        sha = kernel_src_sha()
        pmc = {}
        pmc_path = os.path.join(REPO, "profiles", "pmc_summary.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
            except Exception:
                pmc = {}
        line = {
            "metric": "sequence-pairs/s", "value": rec["value"], "unit": "sequence-pairs/s", "n_gpus": G,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload_text(rec, wl, G), "N": rec["N"], "L": rec["L"], "rows_per_gpu": rec["rows_local"],
                       "alphabet": "5-bit tokens 1..20", "parallelism": f"rowblock{G}", "engine": rec["engine"]},
            "roofline": roofline(rec, wl, pmc, sha),
        }
        if wl["mode"] == "eps":
            line["config"]["nnz"] = rec["result"].get("nnz")
            if rec["result"].get("path"):
                line["config"]["path"] = rec["result"]["path"]
        if wl["mode"] == "lev":
            line["config"].update({"candidates": rec["result"].get("candidates"), "filter_passes": rec["result"].get("filter_passes")})
        if rec["pcie_ms"] is not None:
            line["pcie_inclusive"] = {"ms_per_step": rec["pcie_ms"], "value": float(rec["rows_local"]) * rec["N"] / (rec["pcie_ms"] * 1e-3),
                                      "note": "host tokens H2D + step + results D2H (pageable memory); reported only, not `value`"}
        # the other single-GPU configurations, a few steps each, in the same invocation
        if G == 1 and not use_dist and a.workload == "auto" and not a.no_extra:
            extra = []
            for sub in ("cfg2", "cfg3d", "cfg5", "cfg3b8", "mink64", "mink1280"):
                torch.cuda.empty_cache()
                nst, nwu = (2, 1) if sub == "mink1280" else (5, 2)
                srec, _, swl = run_workload(sub, 1, 0, dev, False, backend, steps=nst, warmup=nwu, want_pcie=False)
                e = {"workload": workload_text(srec, swl, 1), "steps": nst, "warmup": nwu, "ms_per_step": srec["ms_per_step"],
                     "value": srec["value"], "unit": "sequence-pairs/s", "engine": srec["engine"],
                     "roofline": roofline(srec, swl, pmc, sha)}
                if swl["mode"] == "eps":
                    e["nnz"] = srec["result"].get("nnz")
                    e["path"] = srec["result"].get("path")
                if swl["mode"] == "lev":
                    e["candidates"] = srec["result"].get("candidates")
                extra.append(e)
            line["extra"] = extra
            # (the driver's record keeps top-level keys only: the sub-records' essentials ride inside `roofline`)
            line["roofline"]["other_workloads"] = [
                {"workload": e["workload"], "ms_per_step": e["ms_per_step"], "kernel_ms": e["roofline"]["kernel_ms"],
                 "value": e["value"], "engine": e["engine"], "bound": e["roofline"]["bound"], "frac": e["roofline"]["frac"],
                 "kernel": e["roofline"]["kernel"]} for e in extra]
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
