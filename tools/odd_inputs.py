"""Unusual inputs through the public surface on the GPU: variable lengths, unknown letters, a
40-letter alphabet beyond 128 positions (8 planes do not fit: operator path), tiny datasets."""
import os, sys, tempfile, string
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from prograph_amd import Prograph
from oracle import prograph_oracle as O

tmp = tempfile.mkdtemp()
def make(seqs, name, **kw):
    f = os.path.join(tmp, name + ".csv")
    pd.DataFrame({"Sequence": seqs, "Fitness": np.arange(len(seqs), dtype=float)}).to_csv(f)
    return Prograph(f, **kw)

rng = np.random.RandomState(0)
AA = "ACDEFGHIKLMNPQRSTVWY"
# 1. variable lengths + unknown letters
seqs = ["".join(rng.choice(list(AA + "XZ"), size=rng.randint(5, 40))) for _ in range(500)]
seqs[0] = "".join(rng.choice(list(AA), size=40))        # the seed (row 0) must be the longest, as in the reference
pg = make(seqs, "varlen")
tok = pg.tokenized.astype(np.int64)
g = pg.build_graph(k=4); e = pg.build_graph(eps=30)
for r in (0, 123, 499):
    d = O.hamming(tok, tok[r:r + 1]).numpy()[0]
    o = np.argsort(d, kind="stable")[1:5]
    assert np.array_equal(g[r][0], o) and np.array_equal(g[r][1], d[o])
    c = np.nonzero((d <= 30) & (d > 0))[0]
    assert np.array_equal(e[r][0], c) and np.array_equal(e[r][1], d[c])
print("variable length ok")
# 2. 40-letter alphabet, 150 positions
alpha = (string.ascii_uppercase + string.ascii_lowercase)[:40]
seqs = ["".join(rng.choice(list(alpha), size=150)) for _ in range(300)]
for i in range(1, 300, 3):
    s = list(seqs[i - 1]); s[rng.randint(150)] = alpha[rng.randint(40)]; seqs[i] = "".join(s)
pg = make(seqs, "wide", amino_acids=alpha)
tok = pg.tokenized.astype(np.int64)
g = pg.build_graph(k=3); e = pg.build_graph(eps=2)
for r in (0, 1, 299):
    d = O.hamming(tok, tok[r:r + 1]).numpy()[0]
    o = np.argsort(d, kind="stable")[1:4]
    assert np.array_equal(g[r][0], o) and np.array_equal(g[r][1], d[o]), (r, g[r], o)
    c = np.nonzero((d <= 2) & (d > 0))[0]
    assert np.array_equal(e[r][0], c)
print("wide alphabet ok", pg.indexing(distances=[1])[:5])
# 3. tiny datasets
for n in (1, 2, 3):
    pg = make(["ACDEF", "ACDEG", "WCDEG"][:n], f"tiny{n}")
    print("tiny", n, [tuple(map(list, t)) for t in pg.build_graph(k=2)], [tuple(map(list, t)) for t in pg.build_graph(eps=1)])
