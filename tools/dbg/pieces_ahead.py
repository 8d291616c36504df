"""Column pieces as waves of their own from the start (PG_MM_PIECES_AHEAD, default) against pieces through the counter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
def timeit(f, iters=9):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for N, bits, members in ((200000, 5, 256), (139000, 5, 256), (200000, 5, 64), (100000, 5, 256)):
    tok = synth.clustered_tokens(N, 64, members=members)
    p = nat.pack(torch.from_numpy(tok), bits=bits)
    out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
    best = {}
    for rnd in range(3):
        for label, env in (("ahead", {}), ("counter", {"PG_MM_PIECES_AHEAD": "0"}), ("no split", {"PG_MM_SPLIT": "0"})):
            os.environ.update(env); t = timeit(lambda: nat.knn_graph(p, p, 16, out=out))
            for k_ in env: os.environ.pop(k_)
            best[label] = min(best.get(label, 9e9), t)
    print(f"N={N} members={members}: " + "  ".join(f"{k_} {v:.3f}" for k_, v in best.items()), flush=True)
