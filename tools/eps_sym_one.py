import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
tok = synth.clustered_tokens(200000, 64)
p = nat.pack(torch.from_numpy(tok), bits=5)
for sym in ("0", "1"):
    os.environ["PG_EPS_SYM"] = sym
    for _ in range(3):
        nat.eps_graph(p, p, nat.CMP_LE, 2, cap=128)
    torch.cuda.synchronize()
