"""Rows per wave of the MFMA engine against the tail of the grid (kNN 16, N=200k): PG_ROWS_PER_WAVE sweep."""
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
for name, tok in (("cfg3", synth.clustered_tokens(200000, 64)), ("dense", synth.clustered_tokens(200000, 64, members=200000))):
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((200000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((200000, 16), dtype=torch.uint8, device=p.buf.device))
    for rpw in (32, 30, 28, 25, 24, 20, 16, 64):
        os.environ["PG_ROWS_PER_WAVE"] = str(rpw)
        print(f"{name:6s} rows/wave {rpw:3d}  {t(lambda: nat.knn_graph(p, p, 16, out=out)):8.3f} ms", flush=True)
