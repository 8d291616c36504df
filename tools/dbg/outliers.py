"""cfg3 with a share of its rows replaced by unrelated random sequences (rows without near neighbours): what they cost the kNN launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
def timeit(f, iters=7):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
base = synth.clustered_tokens(200000, 64)
rng = np.random.RandomState(5)
for share in (0.0, 0.001, 0.01, 0.05, 0.2):
    tok = base.copy()
    n = int(share * len(tok))
    if n: tok[rng.choice(len(tok), n, replace=False)] = rng.randint(1, 21, size=(n, 64)).astype(np.uint8)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((200000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((200000, 16), dtype=torch.uint8, device=p.buf.device))
    res = []
    for label, env in (("auto", {}), ("mfma", {"PG_ENGINE": "mfma"}), ("valu", {"PG_ENGINE": "valu"})):
        os.environ.update(env); t = timeit(lambda: nat.knn_graph(p, p, 16, out=out))
        for k_ in env: os.environ.pop(k_)
        res.append(f"{label} {t:.3f}")
    print(f"{share * 100:5.1f} % outlier rows: " + "  ".join(res), flush=True)
