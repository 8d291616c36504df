"""eps graph of the full square problem: symmetric (every unordered pair once) vs rectangular path,
whole pipeline (slots + scan + compact, one host sync for nnz)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth

def timeit(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

for N, L, eps in ((50000, 32, 2), (200000, 64, 2), (200000, 64, 1), (100000, 128, 3), (20000, 32, 2)):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    res = []
    for sym in ("0", "1"):
        os.environ["PG_EPS_SYM"] = sym
        for rpw in ("0", "4", "8", "12", "16", "24"):
            if sym == "0" and rpw != "0": continue
            if rpw == "0": os.environ.pop("PG_ROWS_PER_WAVE", None)
            else: os.environ["PG_ROWS_PER_WAVE"] = rpw
            t = timeit(lambda: nat.eps_graph(p, p, nat.CMP_LE, eps, cap=128))
            res.append(f"sym{sym}/rpw{rpw}={t:.3f}")
    os.environ.pop("PG_ROWS_PER_WAVE", None)
    print(f"N={N} L={L} eps<={eps}: " + "  ".join(res), flush=True)
