"""Graph construction for sequences beyond the fused engines' record (L > 255): dense kernel over column segments +
device selection (`Prograph._build_graph_long`), beside the generic batch loop (torch sort / where over the same operator)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from prograph_amd import synth
from prograph_amd.prograph import Prograph
from oracle import prograph_oracle as O

N, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 400
tok = synth.clustered_tokens(N, L, members=64)
df = pd.DataFrame({"Sequence": synth.tokens_to_strings(tok), "Fitness": np.zeros(N)})
path = os.path.join(tempfile.mkdtemp(), "long.csv"); df.to_csv(path)
t = time.time(); pg = Prograph(path); torch.cuda.synchronize(); print(f"Prograph(csv) N={N} L={L}: {time.time()-t:.2f} s")
t = time.time(); g = pg.build_graph(k=8); torch.cuda.synchronize(); print(f"build_graph(k=8): {time.time()-t:.2f} s")
t = time.time(); e = pg.build_graph(eps=3); torch.cuda.synchronize(); print(f"build_graph(eps=3): {time.time()-t:.2f} s")
rows = [0, 17, N - 1]
for r in rows:
    d = O.hamming(tok.astype(np.int64), tok[r].astype(np.int64).reshape(1, -1)).numpy()[0]
    order = np.argsort(d, kind="stable")[1:9]
    assert np.array_equal(g[r][0], order) and np.array_equal(g[r][1], d[order]), r
    cols = np.where((d <= 3) & (d > 0))[0]
    assert np.array_equal(e[r][0], cols) and np.array_equal(e[r][1], d[cols]), r
print("sampled rows match the oracle")
from prograph_amd.distance import hamming
t = time.time(); g2 = pg._build_graph_generic(None, 8, None, 8, False, "Tokenized", hamming, None); torch.cuda.synchronize()
print(f"generic batch loop, k=8: {time.time()-t:.2f} s;  same graph: {all(np.array_equal(a[0], b[0]) for a, b in zip(g, g2))}")
t = time.time(); gc = pg.build_graph(k=8, output="csr"); torch.cuda.synchronize(); print(f"build_graph(k=8, output='csr') (no tuples): {time.time()-t:.3f} s")
