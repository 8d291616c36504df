import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
from oracle import c_oracle as C
os.environ["PG_ENGINE"] = "mfma"
for N, L, members in ((1000, 64, 100), (3000, 64, 100), (20000, 64, 256)):
    tok = synth.clustered_tokens(N, L, members=members)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    ridx, rd = C.knn(tok, 5, fast=True)
    for rot in ("0", "1"):
        os.environ["PG_MM_ROTATE"] = rot
        idx, d = nat.knn_graph(p, p, 5)
        bi = (idx.cpu().numpy() != ridx).any(axis=1); bd = (d.cpu().numpy() != rd).any(axis=1)
        print(f"N={N} rotate={rot}: rows with wrong idx {int(bi.sum())}, wrong dist {int(bd.sum())}; first bad rows {np.nonzero(bi | bd)[0][:8]}", flush=True)
        if bi.any():
            r = int(np.nonzero(bi | bd)[0][0]); print("   got", idx[r].cpu().numpy(), d[r].cpu().numpy(), "want", ridx[r], rd[r])
