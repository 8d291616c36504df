"""End-to-end wall clock of the user-facing calls at scale, by stage:
ingest (read_csv, tokenize, reverse map), plane packing, the eps=1 graph of the constructor (device
CSR), materialisation of the reference's N tuples, kNN, and saving / reloading the graphs as flat arrays.
usage: tools/e2e_bench.py [N] [L]       (run on the GPU box; prints a table, one line per stage)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from prograph_amd import synth, _native
from prograph_amd.prograph import Prograph
from prograph_amd.utils import save

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows = []


def timed(label, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    rows.append((label, time.perf_counter() - t)); print(f"{label:62s} {rows[-1][1]:9.3f} s", flush=True)
    return r


print(f"N={N} L={L}  (clustered synthetic sequences, 256 per cluster)")
tok = synth.clustered_tokens(N, L)
seqs = synth.tokens_to_strings(tok)
df = pd.DataFrame({"Sequence": seqs, "Fitness": np.random.RandomState(0).rand(N)})
tmp = tempfile.mkdtemp(); path = os.path.join(tmp, "synthetic.csv"); df.to_csv(path)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()

# the constructor's stages one by one (what Prograph.__init__ does, prograph/prograph.py:96-144 of the reference)
frame = timed("pd.read_csv", lambda: pd.read_csv(path, index_col=0))
probe = Prograph.__new__(Prograph)
probe.amino_acids = "ACDEFGHIKLMNPQRSTVWY"
probe.tokens = {aa.encode("utf-8"): i for i, aa in enumerate(probe.amino_acids, start=1)}
timed("[round-2 path, for comparison] host tokenize: table lookup over the byte view", lambda: probe.tokenize(frame["Sequence"]))
raw, table = timed("byte view of the strings (np.array(..., dtype=bytes))", lambda: probe._byte_view(frame["Sequence"]))
planes, tokdev = timed("H2D + pg_pack_bytes (letter table + bit slicing + signatures on the device)",
                       lambda: _native.pack_bytes(raw, table.astype(np.uint8), bits=5))
tokens = timed("token matrix back to the host (uint8 D2H + widen to int)", lambda: tokdev.cpu().numpy().astype(int))
timed("seq_idxs reverse map (dict of N strings)", lambda: dict(zip(frame["Sequence"], range(len(frame)))))
csr = timed("eps<=1 graph: slots + scan + compact (device CSR)", lambda: _native.eps_graph(planes, planes, _native.CMP_LE, 1))
print(f"    nnz = {csr[1].numel()}")
from prograph_amd.graph import CSRGraph
g = CSRGraph(*csr, planes.n)
tup = timed("CSRGraph.to_tuples (the reference's N (idx, w) tuples)", g.to_tuples)
timed("DataFrame column assignment of the N tuples", lambda: frame.__setitem__("Neighbours", tup))
del frame, tup, g, csr, planes

pg = timed("Prograph(csv)  [all of the above + summary print]", lambda: Prograph(path))
timed("build_graph(k=16, output='csr')", lambda: pg.build_graph(k=16, output="csr"))
timed("build_graph(k=16) -> tuples", lambda: pg.build_graph(k=16))
timed("build_graph(eps=2, output='csr')", lambda: pg.build_graph(eps=2, output="csr"))
timed("degree() from the device CSR", lambda: pg.degree())
timed("neighbourhood(seed, eps=4)  [1xN Hamming + select]", lambda: pg.neighbourhood(pg.seed.Sequence, 4))
timed("calc_neighbours(seed, eps=2, comp=le)", lambda: pg.calc_neighbours(pg.seed.Sequence, eps=2, comp=__import__("operator").le))
timed("save(graphs='csr')  [frame pickle + flat .npz side-car]", lambda: save(pg, name="flat", directory=tmp + "/", graphs="csr"))
timed("save()  [reference format: pickle with N tuples]", lambda: save(pg, name="tuples", directory=tmp + "/"))
timed("Prograph('flat.pkl')  [graphs restored, no N^2 build]", lambda: Prograph(os.path.join(tmp, "flat.pkl")))
timed("Prograph('tuples.pkl')  [reference format]", lambda: Prograph(os.path.join(tmp, "tuples.pkl")))
for f in ("flat.pkl", "flat.graphs.npz", "tuples.pkl"):
    print(f"    {f}: {os.path.getsize(os.path.join(tmp, f)) / 1e6:.1f} MB")
