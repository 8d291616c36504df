"""cfg3's sequences in lexicographic order (cluster mates adjacent): the same candidates per row, but in 2-3 super-tiles per
pass instead of 256 - what the scattered mates cost the kNN kernel (scan exits / restarts) as opposed to the candidates themselves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
os.environ["PG_ENGINE"] = "mfma"
def timeit(f, iters=9):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
tok = synth.clustered_tokens(200000, 64)
order = np.lexsort(tok.T[::-1])
rng = np.random.RandomState(3)
for name, t in (("generator order (mates 781 columns apart)", tok), ("lexicographic (mates adjacent)", tok[order]), ("random permutation", tok[rng.permutation(len(tok))])):
    p = nat.pack(torch.from_numpy(np.ascontiguousarray(t)), bits=5)
    out = (torch.empty((200000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((200000, 16), dtype=torch.uint8, device=p.buf.device))
    print(f"{name}: kNN16 {timeit(lambda: nat.knn_graph(p, p, 16, out=out)):.3f} ms", flush=True)
