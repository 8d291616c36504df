"""PG_MM_ROTATE 1 / 0 and PG_MM_EVICT 1 / 0 on cfg3 and neighbours, interleaved."""
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
for N, L, members, bits in ((200000, 64, 256, 8), (100000, 64, 256, 8), (200000, 64, 256, 5), (270000, 64, 256, 5)):
    p = nat.pack(torch.from_numpy(synth.clustered_tokens(N, L, members=members)), bits=bits)
    out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
    best = {}
    for rnd in range(3):
        for label, env in (("default", {}), ("no eviction", {"PG_MM_EVICT": "0"})):
            os.environ.update(env); t = timeit(lambda: nat.knn_graph(p, p, 16, out=out))
            for k_ in env: os.environ.pop(k_)
            best[label] = min(best.get(label, 9e9), t)
    print(f"N={N} L={L} bits={bits}: " + "  ".join(f"{k_} {v:.3f}" for k_, v in best.items()), flush=True)
