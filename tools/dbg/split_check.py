"""Column pieces for the rows beyond a full round (knn_launch): exactness against the unsplit launch and timing."""
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
for name, tok in (("cfg3", synth.clustered_tokens(200000, 64)), ("c64", synth.clustered_tokens(200000, 64, members=64)), ("small clusters", synth.clustered_tokens(204800, 64, members=12))):
    N = tok.shape[0]
    p = nat.pack(torch.from_numpy(tok), bits=5)
    os.environ["PG_MM_SPLIT"] = "0"
    ref = nat.knn_graph(p, p, 16); t0 = timeit(lambda: nat.knn_graph(p, p, 16))
    os.environ.pop("PG_MM_SPLIT")
    for pieces in (None, "2", "5", "8", "16"):
        if pieces: os.environ["PG_MM_PIECES"] = pieces
        else: os.environ.pop("PG_MM_PIECES", None)
        got = nat.knn_graph(p, p, 16)
        ok = bool((got[0] == ref[0]).all()) and bool((got[1] == ref[1]).all())
        print(f"{name} N={N}: unsplit {t0:.3f} ms; pieces={pieces or 'auto'} {timeit(lambda: nat.knn_graph(p, p, 16)):.3f} ms  identical={ok}", flush=True)
    os.environ.pop("PG_MM_PIECES", None)
