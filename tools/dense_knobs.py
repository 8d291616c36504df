"""The MFMA engine's density rule under its knob PG_MM_L1 (lane slots with a candidate, of 256, from which a
super-tile leaves the MFMA form for the dense forms): kNN 16 and eps<=2 slots, N=200k L=64 (eps dense: N=50k)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
os.environ["PG_ENGINE"] = "mfma"
def t(f, iters=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
CASES = (("dense", synth.clustered_tokens(200000, 64, members=200000)), ("cfg3", synth.clustered_tokens(200000, 64)),
         ("c2048", synth.clustered_tokens(200000, 64, members=2048)), ("c20000", synth.clustered_tokens(200000, 64, members=20000)),
         ("dense50k", synth.clustered_tokens(50000, 64, members=50000)), ("random", np.random.RandomState(1).randint(1, 21, size=(200000, 64)).astype(np.uint8)))
L1S = (40, 48, 56, 64, 72, 80, 96)
print("kNN16 / eps<=2 slots (ms) per PG_MM_L1:  " + "  ".join(f"{l:>13d}" for l in L1S))
for name, tok in CASES:
    N = tok.shape[0]
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
    cap = 256 if N > 50000 else 2048
    si = torch.empty(N * cap, dtype=torch.int32, device=p.buf.device); sw = torch.empty(N * cap, dtype=torch.uint8, device=p.buf.device)
    cnt = torch.empty(N, dtype=torch.int32, device=p.buf.device)
    line = f"{name:9s}"
    for l1 in L1S:
        os.environ["PG_MM_L1"] = str(l1)
        k = t(lambda: nat.knn_graph(p, p, 16, out=out))
        e = t(lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)) if name not in ("dense", "c20000") else float("nan")
        line += f"  {k:6.3f}/{e:6.3f}"
    print(line, flush=True)
