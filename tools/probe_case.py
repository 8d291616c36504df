"""One engine configuration, a few launches: the target of rocprofv3 --pmc runs.
usage: probe_case.py {random|clustered} {eps|knn} [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth

data, mode = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N, L, cap = 200000, 64, 64
tok = np.random.RandomState(1).randint(1, 21, size=(N, L)).astype(np.uint8) if data == "random" else synth.clustered_tokens(N, L)
p = nat.pack(torch.from_numpy(tok), bits=5)
dev = p.buf.device
si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
cnt = torch.empty(N, dtype=torch.int32, device=dev)
out = (torch.empty((N, 16), dtype=torch.int32, device=dev), torch.empty((N, 16), dtype=torch.uint8, device=dev))
f = (lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)) if mode == "eps" else (lambda: nat.knn_graph(p, p, 16, out=out))
for _ in range(iters):
    f()
torch.cuda.synchronize()
print("done", data, mode)
