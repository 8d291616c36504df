"""kNN time vs the optimistic stage-1 cap (PG_KNN_GUESS) on several shapes."""
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

for N, L, members in ((50000, 32, 256), (200000, 64, 256), (200000, 64, 64), (200000, 64, 32), (100000, 128, 256), (20000, 32, 256)):
    tok = synth.clustered_tokens(N, L, members=members)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
    res = []
    for g in ("0", "4", "6", "8", "10", "12"):
        os.environ["PG_KNN_GUESS"] = g
        res.append((g, timeit(lambda: nat.knn_graph(p, p, 16, out=out))))
    print(f"N={N} L={L} members={members}: " + "  ".join(f"G{g}={t:.3f}" for g, t in res), flush=True)
