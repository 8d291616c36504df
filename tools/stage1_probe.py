"""Stage-1-only probe: eps<=2 on uniform random rows (no pair is near, the filter never triggers)
vs clustered rows, to separate the cost of the lower-bound sweep from the exact-distance work."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth

def timeit(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

N, L = 200000, 64
cap = 64
rng = np.random.RandomState(1)
for name, tok in (("random", rng.randint(1, 21, size=(N, L)).astype(np.uint8)), ("clustered", synth.clustered_tokens(N, L))):
    p = nat.pack(torch.from_numpy(tok), bits=5)
    dev = p.buf.device
    si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev)
    out = (torch.empty((N, 16), dtype=torch.int32, device=dev), torch.empty((N, 16), dtype=torch.uint8, device=dev))
    for filt in ("1", "2", "0"):
        os.environ["PG_LB_FILTER"] = filt
        te = timeit(lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt))
        tk = timeit(lambda: nat.knn_graph(p, p, 16, out=out))
        print(f"{name} filter={filt}: eps<=2 {te:.3f} ms (nnz {int(cnt.sum())})   knn16 {tk:.3f} ms", flush=True)
