import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from prograph_amd import _native as nat, synth
N, L = 200000, 64
tok = synth.clustered_tokens(N, L, members=N)          # ONE cluster: every pair is within 6 substitutions
p = nat.pack(torch.from_numpy(tok), bits=5)
dev = p.buf.device
out = (torch.empty((N,16), dtype=torch.int32, device=dev), torch.empty((N,16), dtype=torch.uint8, device=dev))
for mode in ("0", "1", "2"):
    os.environ["PG_LB_FILTER"] = mode
    nat.knn_graph(p, p, 16, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); nat.knn_graph(p, p, 16, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"dense one-cluster N={N} L={L} kNN16 PG_LB_FILTER={mode}: {np.median(ts):.2f} ms", flush=True)
