"""Rows-per-wave sweep on row-block slices of big problems (column matrix beyond L2): calibrates the
column-streaming term of plan_rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth


def run(N, L, nrows, rpws, iters=3):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    dev = p.buf.device
    out = (torch.empty((nrows, 16), dtype=torch.int32, device=dev), torch.empty((nrows, 16), dtype=torch.uint8, device=dev))
    res = []
    for w in rpws:
        if w: os.environ["PG_ROWS_PER_WAVE"] = str(w)
        else: os.environ.pop("PG_ROWS_PER_WAVE", None)
        f = lambda: nat.knn_graph(p, p, 16, row0=0, nrows=nrows, out=out)
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(iters):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res.append((w, float(np.median(ts))))
    print(f"N={N} L={L} rows={nrows} knn: " + "  ".join(f"rpw{w}={t:.2f}" for w, t in res), flush=True)


rp = [0, 8, 12, 16, 20, 24, 28]
run(1000000, 64, 125000, rp)
run(500000, 64, 125000, rp)
run(400000, 64, 100000, rp)
run(300000, 64, 125000, rp)
