"""eps pipeline time vs slot capacity on data with many neighbours per row (overflowing rows are
recomputed by the compaction kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth

def timeit(f, iters=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

for N, L, eps, members in ((100000, 64, 1, 100000), (100000, 64, 3, 256), (50000, 32, 4, 256)):
    tok = synth.clustered_tokens(N, L, members=members)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    res = []
    for cap in (32, 64, 128, 256, 512, 1024):
        ip, ix, w = nat.eps_graph(p, p, nat.CMP_LE, eps, cap=cap)
        t = timeit(lambda: nat.eps_graph(p, p, nat.CMP_LE, eps, cap=cap))
        res.append(f"cap{cap}={t:.2f}")
    deg = (ip[1:] - ip[:-1]).float()
    print(f"N={N} L={L} eps<={eps} members={members} nnz={int(ip[-1])} deg mean {deg.mean():.0f} max {int(deg.max())}: " + "  ".join(res), flush=True)
