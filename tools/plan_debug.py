"""Print the row plan (occupancy, rows per wave) the library picks for a few shapes: PG_DEBUG_PLAN=1."""
import os, sys
os.environ["PG_DEBUG_PLAN"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
for N, L, bits in ((50000, 32, 5), (50000, 32, 8), (200000, 64, 5), (200000, 64, 8), (100000, 128, 5), (30000, 200, 5)):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=bits)
    print(f"--- N={N} L={L} bits={bits}: eps then knn", file=sys.stderr, flush=True)
    nat.eps_graph(p, p, nat.CMP_LE, 2)
    nat.knn_graph(p, p, 16)
    torch.cuda.synchronize()
