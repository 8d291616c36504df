"""How the kNN sweep of the MFMA engine scales with the number of row passes (tail of the grid):
kNN 16 of the first `nrows` rows against all N=200k columns."""
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
    for nrows in (32768, 65536, 98304, 131072, 163840, 200000):
        out = (torch.empty((nrows, 16), dtype=torch.int32, device=p.buf.device), torch.empty((nrows, 16), dtype=torch.uint8, device=p.buf.device))
        ms = t(lambda: nat.knn_graph(p, p, 16, row0=0, nrows=nrows, out=out))
        print(f"{name:6s} rows {nrows:7d} = {nrows / 32 / 1024:5.2f} waves/SIMD  {ms:8.3f} ms   {ms / nrows * 32768:7.3f} ms per 32k rows", flush=True)
