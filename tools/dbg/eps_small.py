import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat
g = np.load("tests/golden/synth_n2085_l64.npz")
tok = g["tokens"]
p = nat.pack(torch.from_numpy(np.ascontiguousarray(tok)), bits=5)
os.environ["PG_ENGINE"] = "mfma"
indptr, idx, w = [x.cpu().numpy() for x in nat.eps_graph(p, p, nat.CMP_LE, 1, cap=256)]
ref = g["eps1_indptr"]
cnt, rcnt = np.diff(indptr), np.diff(ref)
bad = np.nonzero(cnt != rcnt)[0]
print("rows with wrong count:", len(bad), bad[:40])
for r in bad[:6]:
    print(r, "got", idx[indptr[r]:indptr[r+1]], "want", g["eps1_indices"][ref[r]:ref[r+1]])
