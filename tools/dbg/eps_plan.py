"""eps slots on the MFMA engine: round 2's pass plan (rows spread over ~97 % of the slots) against the kNN plan (fewest waves per SIMD)."""
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
for N, L in ((30000, 64), (50000, 32), (50000, 64), (100000, 128), (100000, 64), (150000, 64), (200000, 64)):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=5); dev = p.buf.device; cap = 256
    si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev); cl = torch.empty(N, dtype=torch.int32, device=dev)
    L_ = nat.lib(); ws = nat.workspace(N, dev)
    sym = lambda: nat._check(L_.pg_eps_slots_sym(nat._ptr(p.buf), p.npad, p.n, p.g * 32, p.bits, nat.CMP_LE, 2.0, cap, nat._ptr(si), nat._ptr(sw), nat._ptr(cnt), nat._ptr(cl), nat._ptr(ws), nat._stream()), "sym")
    rect = lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)
    res = []
    for name, f in (("rect", rect), ("sym", sym)):
        best = {}
        for rnd in range(2):
            for label, env in (("r2 plan", {}), ("knn plan", {"PG_MM_PLAN": "1"}), ("rpw32", {"PG_ROWS_PER_WAVE": "32"}), ("rpw16", {"PG_ROWS_PER_WAVE": "16"}), ("rpw8", {"PG_ROWS_PER_WAVE": "8"})):
                os.environ.update(env); t = timeit(f)
                for k_ in env: os.environ.pop(k_)
                best[label] = min(best.get(label, 9e9), t)
        res.append(name + ": " + "  ".join(f"{k_} {v:.3f}" for k_, v in best.items()))
    print(f"N={N} L={L}  " + "   |   ".join(res), flush=True)
