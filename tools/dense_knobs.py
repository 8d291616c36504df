"""Dense data (one cluster, N=200k, L=64, kNN 16): the MFMA engine under its density-rule knobs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
tok = synth.clustered_tokens(200000, 64, members=200000)
p = nat.pack(torch.from_numpy(tok), bits=5)
out = (torch.empty((200000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((200000, 16), dtype=torch.uint8, device=p.buf.device))
def t(f, iters=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for name, env in (("default", {}), ("level 2 never selective -> direct (PG_MM_L2=1)", {"PG_MM_L2": "1"}), ("direct only (PG_LB_FILTER=0)", {"PG_LB_FILTER": "0"}),
                  ("never level 2 (PG_MM_L1=257)", {"PG_MM_L1": "257"}), ("level 2 from 32 slots (PG_MM_L1=32)", {"PG_MM_L1": "32"}),
                  ("runs of 32 (PG_MM_RUN=32)", {"PG_MM_RUN": "32"}), ("cap 3 (PG_KNN_GUESS=3)", {"PG_KNN_GUESS": "3"}),
                  ("VALU engine", {"PG_ENGINE": "valu"}), ("VALU engine direct only", {"PG_ENGINE": "valu", "PG_LB_FILTER": "0"})):
    for k in ("PG_MM_L1", "PG_MM_L2", "PG_MM_RUN", "PG_LB_FILTER", "PG_KNN_GUESS", "PG_ENGINE"):
        os.environ.pop(k, None)
    os.environ.setdefault("PG_ENGINE", "mfma")
    os.environ.update(env)
    print(f"{name:55s} {t(lambda: nat.knn_graph(p, p, 16, out=out)):8.2f} ms", flush=True)
