import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prograph_amd import _native as nat, synth
def timeit(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
rng = np.random.RandomState(1)
cases = {"dense": synth.clustered_tokens(200000, 64, members=200000), "random": rng.randint(1, 21, size=(200000, 64)).astype(np.uint8),
         "c2048": synth.clustered_tokens(200000, 64, members=2048), "c64": synth.clustered_tokens(200000, 64, members=64)}
for name in sys.argv[1:] or list(cases):
    tok = cases[name]; N = tok.shape[0]
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
    res = []
    for env in ({"PG_ENGINE": "mfma", "PG_MM_R": "1"}, {"PG_ENGINE": "mfma", "PG_MM_R": "2"}, {"PG_ENGINE": "mfma", "PG_MM_R": "1", "PG_MM_SHORT": "0"}, {"PG_ENGINE": "valu"}):
        for k in ("PG_ENGINE", "PG_MM_R", "PG_MM_SHORT"): os.environ.pop(k, None)
        os.environ.update(env)
        res.append((",".join(f"{k[3:]}={v}" for k, v in env.items()), timeit(lambda: nat.knn_graph(p, p, 16, out=out))))
    print(name, "  ".join(f"[{a}] {t:.3f}" for a, t in res), flush=True)
