import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from prograph_amd import _native as nat, synth
N, L = 200000, 64
mode = sys.argv[1] if len(sys.argv) > 1 else "knn"
tok = synth.clustered_tokens(N, L)
p = nat.pack(torch.from_numpy(tok), bits=5)
dev = p.buf.device
out = (torch.empty((N,16), dtype=torch.int32, device=dev), torch.empty((N,16), dtype=torch.uint8, device=dev))
cap=256
si = torch.empty(N*cap, dtype=torch.int32, device=dev); sw = torch.empty(N*cap, dtype=torch.uint8, device=dev); cnt = torch.empty(N, dtype=torch.int32, device=dev)
for _ in range(3):
    if mode == "knn": nat.knn_graph(p, p, 16, out=out)
    else: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)
torch.cuda.synchronize()
