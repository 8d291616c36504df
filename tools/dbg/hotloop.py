"""Hot-loop-only timing: random tokens (no candidate anywhere), eps <= 2 slots on the MFMA engine, N = 200k."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PG_ENGINE"] = "mfma"
import numpy as np, torch
from prograph_amd import _native as nat
rng = np.random.RandomState(1)
N = int(os.environ.get("HL_N", "200000"))
tok = rng.randint(1, 21, size=(N, 64)).astype(np.uint8)
p = nat.pack(torch.from_numpy(tok), bits=5)
dev = p.buf.device
cap = 16
si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
cnt = torch.empty(N, dtype=torch.int32, device=dev)
f = lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)
f(); torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(os.path.basename(os.environ.get("PROGRAPH_HIP_LIB", "default")), f"random eps2 N={N}: {np.median(ts):.3f} ms; matches {int(cnt.sum())}")
