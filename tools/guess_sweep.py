"""Optimistic cap G0 of the kNN sweep (PG_KNN_GUESS) on the MFMA engine: kNN 16 kernel time by shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
os.environ["PG_ENGINE"] = "mfma"
def t(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
CASES = (("cfg3", 200000, 64, 256, None), ("c64", 200000, 64, 64, None), ("c2048", 200000, 64, 2048, None), ("100k/128", 100000, 128, 256, None),
         ("1M slice", 1000000, 64, 256, 125000), ("random", 200000, 64, 0, None))
for name, N, L, members, nrows in CASES:
    tok = synth.clustered_tokens(N, L, members=members) if members else np.random.RandomState(1).randint(1, 21, size=(N, L)).astype(np.uint8)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    nr = nrows or N
    out = (torch.empty((nr, 16), dtype=torch.int32, device=p.buf.device), torch.empty((nr, 16), dtype=torch.uint8, device=p.buf.device))
    line = f"{name:9s}"
    for g in (5, 6, 7, 8, 10, 12, 16):
        os.environ["PG_KNN_GUESS"] = str(g)
        line += f"  G0={g}: {t(lambda: nat.knn_graph(p, p, 16, row0=0, nrows=nr, out=out)):7.3f}"
    print(line, flush=True)
