"""Debug helper: MFMA engine vs VALU engine vs golden on one golden set (run on the GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from prograph_amd import _native as nat
name = sys.argv[1] if len(sys.argv) > 1 else "synth_n300_varlen24"
g = load_golden(name)
tok = g["tokens"]
print(name, tok.shape, [k for k in g.files if k.startswith("knn")])
p = nat.pack(torch.from_numpy(np.ascontiguousarray(tok)), bits=5)
for key in g.files:
    if not (key.startswith("knn") and key.endswith("_idx")) or "sub" in key or "sim" in key:
        continue
    k = int(key[3:-4])
    for guess in ("8", "0"):
        os.environ["PG_KNN_GUESS"] = guess
        os.environ["PG_ENGINE"] = "mfma"
        idx, dist = nat.knn_graph(p, p, k)
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
        bad = np.nonzero((idx != g[f"knn{k}_idx"]).any(1) | (dist != g[f"knn{k}_w"]).any(1))[0]
        print(f"k={k} guess={guess}: {len(bad)} bad rows", bad[:20])
        for r in bad[:4]:
            print("  row", r, "got", list(zip(idx[r], dist[r])), "\n      want", list(zip(g[f'knn{k}_idx'][r], g[f'knn{k}_w'][r])))
            d = (tok != tok[r]).sum(1)
            print("      lens: row", int((tok[r] != 0).sum()), " sorted d head", np.sort(d)[:12])

# host emulation of the 31-bit signature bound for the missing pairs
def sig31(t):
    s = 0
    L = len(t)
    for g in range((L + 31) // 32):
        w = 0
        for j in range(32):
            pos = g * 32 + j
            if pos < L and (int(t[pos]) & 1):
                w |= 1 << j
        s ^= w
    return (s ^ (s >> 31)) & 0x7FFFFFFF
sg = [sig31(t) for t in tok]
k = 8
os.environ["PG_KNN_GUESS"] = "8"
idx, dist = nat.knn_graph(p, p, k)
idx = idx.cpu().numpy()
want = g["knn8_idx"]
for r in np.nonzero((idx != want).any(1))[0]:
    miss = [c for c in want[r] if c not in idx[r]]
    for c in miss:
        print("row", r, "missing col", c, "d", int((tok[r] != tok[c]).sum()), "lb", bin(sg[r] ^ sg[c]).count("1"),
              "pass-row", r % 32, "tile", c // 32, "lane", c % 32)
for gs in ("3", "40", "1", "16"):
    os.environ["PG_KNN_GUESS"] = gs
    i2, d2 = nat.knn_graph(p, p, k)
    print("guess", gs, "bad rows", int((i2.cpu().numpy() != want).any(1).sum()))
