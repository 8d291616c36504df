"""One GPU's slice of BASELINE configs[3] (125 000 rows x 1 000 000 columns, kNN 16): 32-row against 64-row passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
def timeit(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
N = 1_000_000
p = nat.pack(torch.from_numpy(synth.clustered_tokens(N, 64)), bits=5)
rows = 125_000
out = (torch.empty((rows, 16), dtype=torch.int32, device=p.buf.device), torch.empty((rows, 16), dtype=torch.uint8, device=p.buf.device))
best = {}
for rnd in range(2):
    for label, env in (("auto", {}), ("mfma R=1", {"PG_ENGINE": "mfma", "PG_MM_R": "1"}), ("mfma R=2", {"PG_ENGINE": "mfma", "PG_MM_R": "2"}),
                       ("R=1 rpw 32", {"PG_ENGINE": "mfma", "PG_MM_R": "1", "PG_ROWS_PER_WAVE": "32"}), ("R=2 rpw 64", {"PG_ENGINE": "mfma", "PG_MM_R": "2", "PG_ROWS_PER_WAVE": "64"})):
        os.environ.update(env); t = timeit(lambda: nat.knn_graph(p, p, 16, row0=375_000, nrows=rows, out=out))
        for k_ in env: os.environ.pop(k_)
        best[label] = min(best.get(label, 9e9), t)
print("  ".join(f"{k_}: {v:.3f}" for k_, v in best.items()))
