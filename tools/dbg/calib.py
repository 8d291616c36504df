"""Calibration of the kNN plan's cost table: T(R, waves per SIMD, columns) = S + H * columns.  Rows = whole waves on every SIMD,
two column counts (300 000 and 900 000, clusters of 256 either way), column pieces off, MFMA engine forced."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
os.environ["PG_ENGINE"] = "mfma"; os.environ["PG_MM_SPLIT"] = "0"
def timeit(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
res = {}
for N in (300_000, 900_000):
    p = nat.pack(torch.from_numpy(synth.clustered_tokens(N, 64)), bits=5)
    for R, rb in ((1, 32), (2, 64)):
        for w in (1, 2, 3, 4):
            rows = w * 1024 * rb
            out = (torch.empty((rows, 16), dtype=torch.int32, device=p.buf.device), torch.empty((rows, 16), dtype=torch.uint8, device=p.buf.device))
            os.environ["PG_MM_R"] = str(R); os.environ["PG_ROWS_PER_WAVE"] = str(rb)
            res[(N, R, w)] = timeit(lambda: nat.knn_graph(p, p, 16, row0=1024, nrows=rows, out=out))
    del p
for R in (1, 2):
    for w in (1, 2, 3, 4):
        a, b = res[(300_000, R, w)], res[(900_000, R, w)]
        H = (b - a) / 6.0; S = a - 3.0 * H          # per 100 000 columns
        print(f"R={R} w={w}: 300k cols {a:.3f} ms, 900k cols {b:.3f} ms  ->  S = {S:.3f} ms, H = {H:.4f} ms per 100k columns", flush=True)
