import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from prograph_amd import _native as nat, synth
N, L, M = 200000, 64, 8192
tok = synth.clustered_tokens(N, L)
xp = nat.pack(torch.from_numpy(tok), bits=5)
yp = nat.pack(torch.from_numpy(tok[:M].copy()), bits=5)
for ob in (8, 4, 1):
    out = nat.hamming_dense(xp, yp, out_bytes=ob); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = nat.hamming_dense(xp, yp, out_bytes=ob); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = np.median(ts) * 1e-3
    wb = M * N * ob
    print(f"dense (M={M}, N={N}) out={ob}B: {t*1e3:.2f} ms  {M*N/t:.3e} pairs/s  write {wb/t/1e12:.2f} TB/s ({wb/t/8e12*100:.0f}% of 8 TB/s)", flush=True)
    del out
