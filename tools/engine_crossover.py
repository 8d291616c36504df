"""VALU engine vs MFMA engine by problem size (kNN 16 and eps<=2 slots, clustered data): where the automatic choice
(PG_ENGINE_MIN_ROWS) should switch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
def t(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for N, L in ((4000, 64), (8000, 64), (12000, 64), (16000, 64), (16000, 32), (24000, 32), (32000, 64), (50000, 32), (50000, 64), (65000, 64), (65000, 128), (100000, 64)):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
    cap = 256
    si = torch.empty(N * cap, dtype=torch.int32, device=p.buf.device); sw = torch.empty(N * cap, dtype=torch.uint8, device=p.buf.device)
    cnt = torch.empty(N, dtype=torch.int32, device=p.buf.device)
    line = f"N={N:6d} L={L:3d}"
    for eng in ("valu", "mfma"):
        os.environ["PG_ENGINE"] = eng
        k = t(lambda: nat.knn_graph(p, p, 16, out=out))
        e = t(lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt))
        line += f"   {eng}: kNN {k:6.3f}  eps {e:6.3f}"
    print(line, flush=True)
