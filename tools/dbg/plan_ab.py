"""New plan (fewest waves per SIMD) against round 2's spreading rule (PG_MM_PLAN=2), MFMA engine forced: kNN and eps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
os.environ["PG_ENGINE"] = "mfma"
def timeit(f, iters=7):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for N, L, bits in ((100000, 128, 5), (100000, 64, 5), (140000, 64, 5), (200000, 64, 5), (200000, 64, 8), (100000, 64, 8), (270000, 64, 5), (300000, 64, 5)):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=bits); dev = p.buf.device; cap = 256
    si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev); cl = torch.empty(N, dtype=torch.int32, device=dev)
    out = (torch.empty((N, 16), dtype=torch.int32, device=dev), torch.empty((N, 16), dtype=torch.uint8, device=dev))
    L_ = nat.lib(); ws = nat.workspace(N, dev)
    sym = lambda: nat._check(L_.pg_eps_slots_sym(nat._ptr(p.buf), p.npad, p.n, p.g * 32, p.bits, nat.CMP_LE, 2.0, cap, nat._ptr(si), nat._ptr(sw), nat._ptr(cnt), nat._ptr(cl), nat._ptr(ws), nat._stream()), "sym")
    rect = lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)
    knn = lambda: nat.knn_graph(p, p, 16, out=out)
    res = []
    for name, f in (("knn", knn),):
        best = {}
        for rnd in range(3):
            for label, env in (("new", {}), ("no split", {"PG_MM_SPLIT": "0"}), ("old plan", {"PG_MM_PLAN": "2", "PG_MM_SPLIT": "0"})):
                os.environ.update(env); t = timeit(f)
                for k_ in env: os.environ.pop(k_)
                best[label] = min(best.get(label, 1e9), t)
        res.append(name + ": " + "  ".join(f"{k_} {v:.3f}" for k_, v in best.items()))
    os.environ.pop("PG_MM_PLAN", None)
    print(f"N={N} L={L} bits={bits}  " + "   ".join(res), flush=True)
