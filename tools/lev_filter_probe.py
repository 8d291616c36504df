import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from prograph_amd import _native as nat, synth
N = 200000
tok, lens = synth.clustered_varlen_tokens(N)
t = torch.from_numpy(tok).cuda()
L = nat.lib(); dev = t.device
np_ = nat.npad(N)
prof = torch.empty(3 * np_ * 16, dtype=torch.uint8, device=dev); ln = torch.empty(N, dtype=torch.int32, device=dev); fl = torch.zeros(1, dtype=torch.int32, device=dev)
nat._check(L.pg_lev_profile(nat._ptr(t), N, 128, t.stride(0), nat._ptr(prof), np_, nat._ptr(ln), nat._ptr(fl), nat._stream()), "p")
cap = 512
si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev); cnt = torch.empty(N, dtype=torch.int32, device=dev)
for band in (0, 2, 8):
    for mode in ("0", "2"):
        os.environ["PG_LB_FILTER"] = mode
        ts = []
        for _ in range(4):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            nat._check(L.pg_lev_candidates(nat._ptr(prof), np_, N, 0, N, band, cap, nat._ptr(si), nat._ptr(sw), nat._ptr(cnt), nat._stream()), "c")
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(f"band={band} filter={mode}: {np.median(ts):.2f} ms, candidates {int(cnt.to(torch.int64).sum())}", flush=True)
