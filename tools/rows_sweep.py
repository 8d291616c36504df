"""Rows per pass of the MFMA engine (PG_ROWS_PER_WAVE) against kernel time: kNN 16 and eps<=2 on several shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
os.environ["PG_ENGINE"] = "mfma"
def t(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
CASES = (("cfg3 200k/64", lambda: synth.clustered_tokens(200000, 64)), ("dense 200k/64", lambda: synth.clustered_tokens(200000, 64, members=200000)),
         ("100k/128", lambda: synth.clustered_tokens(100000, 128)), ("cfg2 50k/32", lambda: synth.clustered_tokens(50000, 32)), ("150k/64", lambda: synth.clustered_tokens(150000, 64)),
         ("300k/64", lambda: synth.clustered_tokens(300000, 64)))
for name, mk in CASES:
    tok = mk(); N = tok.shape[0]
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
    cap = 256
    si = torch.empty(N * cap, dtype=torch.int32, device=p.buf.device); sw = torch.empty(N * cap, dtype=torch.uint8, device=p.buf.device)
    cnt = torch.empty(N, dtype=torch.int32, device=p.buf.device)
    line = f"{name:14s}"
    for rpw in (32, 30, 28, 26, 24, 20, 16):
        os.environ["PG_ROWS_PER_WAVE"] = str(rpw)
        k = t(lambda: nat.knn_graph(p, p, 16, out=out))
        e = t(lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)) if "dense" not in name else float("nan")
        line += f"  r{rpw}: {k:6.3f}/{e:6.3f}"
    print(line + "   (kNN16 / eps2 slots, ms)", flush=True)
