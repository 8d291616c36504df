import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from prograph_amd import _native as nat, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
tok, lens = synth.clustered_varlen_tokens(N)
t = torch.from_numpy(tok).cuda()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, d, st = nat.levenshtein_knn(t, 8, band=8, cap=512, return_stats=True)
    torch.cuda.synchronize(); print("total ms", (time.perf_counter() - t0) * 1e3, st, flush=True)
print("dist histogram of ranks:", np.bincount(d.cpu().numpy().ravel(), minlength=10)[:10])
