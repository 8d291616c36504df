"""Sweep rows-per-wave (PG_ROWS_PER_WAVE) to expose the resident-round quantisation of the engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth


def run(N, L, mode, rpws, iters=5):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    cap = 256; dev = p.buf.device
    si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev)
    out = (torch.empty((N, 16), dtype=torch.int32, device=dev), torch.empty((N, 16), dtype=torch.uint8, device=dev))
    f = (lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)) if mode == "eps" else \
        (lambda: nat.knn_graph(p, p, 16, out=out))
    res = []
    for w in rpws:
        if w: os.environ["PG_ROWS_PER_WAVE"] = str(w)
        else: os.environ.pop("PG_ROWS_PER_WAVE", None)
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res.append((w, float(np.median(ts))))
    print(f"N={N} L={L} {mode}: " + "  ".join(f"rpw{w}={t:.3f}" for w, t in res), flush=True)


rp = [0, 4, 8, 12, 16, 20, 24, 28, 32]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    run(50000, 32, "eps", rp); run(50000, 32, "knn", rp); run(20000, 32, "eps", rp); run(100000, 64, "eps", rp)
else:
    run(200000, 64, "knn", rp); run(200000, 64, "eps", rp)
    run(50000, 32, "knn", rp); run(50000, 32, "eps", rp)
    run(125000, 64, "knn", rp)
