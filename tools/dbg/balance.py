"""Does the MFMA kNN kernel's time follow the workgroups per CU?  N = 196 608 = 3072 passes of 64 rows = exactly three
workgroups on every CU, beside the 200 000-row plan (3847 passes of 52 rows: four workgroups on most CUs) and 64-row passes there."""
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
for nrows in (200000, 196608, 180000, 160000, 147456, 131072):
    out = (torch.empty((nrows, 16), dtype=torch.int32, device=full.buf.device), torch.empty((nrows, 16), dtype=torch.uint8, device=full.buf.device))
    res = []
    for rpw in (None, "64", "56", "48", "32"):
        if rpw: os.environ["PG_ROWS_PER_WAVE"] = rpw
        else: os.environ.pop("PG_ROWS_PER_WAVE", None)
        os.environ["PG_MM_R"] = "1" if rpw == "32" else "2"
        res.append(f"rpw={rpw or 'plan'}: {timeit(lambda: nat.knn_graph(full, full, 16, row0=0, nrows=nrows, out=out)):.3f}")
    print(f"rows {nrows} x 200000 cols  " + "  ".join(res), flush=True)
