"""End-to-end wall clock of the user-facing calls at scale (ingest -> graph -> host tuples), by stage.
usage: tools/e2e_bench.py [N] [L]"""
import os, sys, time, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from prograph_amd import synth
from prograph_amd.prograph import Prograph

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tok = synth.clustered_tokens(N, L)
t0 = time.time(); seqs = synth.tokens_to_strings(tok); print(f"strings {time.time()-t0:.2f}s", flush=True)
df = pd.DataFrame({"Sequence": seqs, "Fitness": np.random.RandomState(0).rand(N)})
tmp = tempfile.mkdtemp(); path = os.path.join(tmp, "synthetic.csv"); df.to_csv(path)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()

def timed(label, f):
    torch.cuda.synchronize(); t = time.time(); r = f(); torch.cuda.synchronize(); print(f"{label:46s} {time.time()-t:8.3f} s", flush=True); return r

pr = cProfile.Profile(); pr.enable()
pg = timed("Prograph(csv)  [ingest + eps=1 graph + print]", lambda: Prograph(path))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
timed("build_graph(k=16) -> tuples", lambda: pg.build_graph(k=16))
timed("build_graph(k=16, output='csr')", lambda: pg.build_graph(k=16, output="csr"))
timed("build_graph(eps=2) -> tuples", lambda: pg.build_graph(eps=2))
timed("build_graph(eps=2, output='csr')", lambda: pg.build_graph(eps=2, output="csr"))
for _ in range(2):
    timed("build_graph(k=16, output='csr') again", lambda: pg.build_graph(k=16, output="csr"))
pr = cProfile.Profile(); pr.enable()
timed("build_graph(k=16, output='csr') profiled", lambda: pg.build_graph(k=16, output="csr"))
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
pr = cProfile.Profile(); pr.enable()
timed("build_graph(eps=2) -> tuples profiled", lambda: pg.build_graph(eps=2))
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
timed("indexing(positions=[3, 7])", lambda: pg.indexing(positions=[3, 7]))
timed("degree()", lambda: pg.degree())
