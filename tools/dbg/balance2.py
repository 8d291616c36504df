"""R=1 vs R=2 below one R=1 round, and the cost of waves per SIMD for R=1 (rows x 200 000 columns, cfg3 data)."""
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
tok = synth.clustered_tokens(200000, 64)
full = nat.pack(torch.from_numpy(tok), bits=5)
for nrows in (32768, 65536, 80000, 98304, 100000, 115000, 131072, 262144 - 200000 + 100000):
    nrows = min(nrows, 200000)
    out = (torch.empty((nrows, 16), dtype=torch.int32, device=full.buf.device), torch.empty((nrows, 16), dtype=torch.uint8, device=full.buf.device))
    res = []
    for R, rpw in (("1", None), ("1", "32"), ("1", "24"), ("2", "64"), ("2", "48"), ("2", "40")):
        if rpw: os.environ["PG_ROWS_PER_WAVE"] = rpw
        else: os.environ.pop("PG_ROWS_PER_WAVE", None)
        os.environ["PG_MM_R"] = R
        res.append(f"R{R}/{rpw or 'plan'}: {timeit(lambda: nat.knn_graph(full, full, 16, row0=0, nrows=nrows, out=out)):.3f}")
    print(f"rows {nrows}  " + "  ".join(res), flush=True)
