"""Full eps graphs (slots + scan + compact + fill pass -> CSR) of dense one-cluster data, wall clock with one sync:
the numbers DESIGN.md quotes for VERDICT r1 item 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
def wall(f, iters=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3, r
for N in (50000, 100000):
    tok = synth.clustered_tokens(N, 64, members=N)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    for eps in (1, 2):
        for eng in ("valu", "mfma"):
            os.environ["PG_ENGINE"] = eng
            ms, (ip, ix, w) = wall(lambda: nat.eps_graph(p, p, nat.CMP_LE, eps))
            deg = (ip[1:] - ip[:-1])
            print(f"dense N={N} L=64 eps<={eps} [{eng}]: {ms:8.2f} ms   nnz {int(ip[-1])}  max degree {int(deg.max())}  ({int(ip[-1]) * 5 / ms / 1e6:.0f} GB/s of CSR output)", flush=True)
    del p
