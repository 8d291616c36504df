import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
from oracle import c_oracle as C
tok = synth.clustered_tokens(3000, 40, seed=5, members=500)
tok[10] = tok[2000]; tok[11] = tok[2000]
p = nat.pack(torch.from_numpy(tok), bits=5)
for k in (16, 40, 63, 64, 100):
    idx, d = nat.knn_graph(p, p, k)
    ridx, rd = C.knn(tok, k)
    idx = idx.cpu().numpy(); d = d.cpu().numpy()
    bad = np.argwhere((idx != ridx) | (d != rd))
    print("k", k, "mismatches", len(bad))
    if len(bad):
        r = bad[0][0]
        print(" first rows", np.unique(bad[:, 0])[:10], "ranks of row", r, bad[bad[:, 0] == r][:, 1][:20])
        print(" got ", idx[r][:12], d[r][:12]); print(" want", ridx[r][:12], rd[r][:12])
        j = bad[bad[:, 0] == r][0, 1]
        print(" at rank", j, "got", idx[r][j-2:j+4], d[r][j-2:j+4], "want", ridx[r][j-2:j+4], rd[r][j-2:j+4])
