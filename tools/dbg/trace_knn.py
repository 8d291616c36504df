"""One kNN call at cfg3 under `rocprofv3 --kernel-trace`: the launches of the call and their durations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
N = int(os.environ.get("TR_N", "200000"))
L = int(os.environ.get("TR_L", "64"))
p = nat.pack(torch.from_numpy(synth.clustered_tokens(N, L)), bits=5)
out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
for _ in range(4):
    nat.knn_graph(p, p, 16, out=out)
torch.cuda.synchronize()
