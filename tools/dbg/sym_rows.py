import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
os.environ["PG_ENGINE"] = "mfma"
def timeit(f, iters=9):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for N, L in ((40000, 32), (50000, 32), (50000, 64), (70000, 64)):
    p = nat.pack(torch.from_numpy(synth.clustered_tokens(N, L)), bits=5); dev = p.buf.device; cap = 256
    si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev); cl = torch.empty(N, dtype=torch.int32, device=dev)
    L_ = nat.lib(); ws = nat.workspace(N, dev)
    sym = lambda: nat._check(L_.pg_eps_slots_sym(nat._ptr(p.buf), p.npad, p.n, p.g * 32, p.bits, nat.CMP_LE, 2.0, cap, nat._ptr(si), nat._ptr(sw), nat._ptr(cnt), nat._ptr(cl), nat._ptr(ws), nat._stream()), "sym")
    best = {}
    for rnd in range(2):
        for r in ("plan", "4", "6", "8", "10", "12", "16"):
            if r == "plan": os.environ.pop("PG_ROWS_PER_WAVE", None)
            else: os.environ["PG_ROWS_PER_WAVE"] = r
            best[r] = min(best.get(r, 9e9), timeit(sym))
    os.environ.pop("PG_ROWS_PER_WAVE", None)
    print(f"N={N} L={L} eps sym: " + "  ".join(f"{k_}={v:.3f}" for k_, v in best.items()), flush=True)
