"""The optimistic kNN cap (PG_KNN_GUESS) on the round-3 kernel: cfg3, N=100k L=128, the cfg4 block, MFMA engine forced."""
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
for N, L, rows in ((200000, 64, None), (100000, 128, None), (50000, 32, None), (1000000, 64, 125000)):
    p = nat.pack(torch.from_numpy(synth.clustered_tokens(N, L)), bits=5)
    nr = rows or N
    out = (torch.empty((nr, 16), dtype=torch.int32, device=p.buf.device), torch.empty((nr, 16), dtype=torch.uint8, device=p.buf.device))
    best = {}
    for rnd in range(2):
        for g in ("default", "5", "6", "7", "8", "9", "10", "12"):
            if g == "default": os.environ.pop("PG_KNN_GUESS", None)
            else: os.environ["PG_KNN_GUESS"] = g
            t = timeit(lambda: nat.knn_graph(p, p, 16, row0=0, nrows=nr, out=out))
            best[g] = min(best.get(g, 9e9), t)
    os.environ.pop("PG_KNN_GUESS", None)
    print(f"N={N} L={L} rows={nr}: " + "  ".join(f"G{k_}={v:.3f}" for k_, v in best.items()), flush=True)
