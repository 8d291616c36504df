import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from prograph_amd import _native as nat, synth
def run(N, L, mode, wpcs, iters=7):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    cap = 256; dev = p.buf.device
    si = torch.empty(N*cap, dtype=torch.int32, device=dev); sw = torch.empty(N*cap, dtype=torch.uint8, device=dev); cnt = torch.empty(N, dtype=torch.int32, device=dev)
    out = (torch.empty((N,16), dtype=torch.int32, device=dev), torch.empty((N,16), dtype=torch.uint8, device=dev))
    f = (lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)) if mode == "eps" else (lambda: nat.knn_graph(p, p, 16, out=out))
    res = []
    for w in wpcs:
        os.environ["PG_WAVES_PER_CU"] = str(w)
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res.append((w, float(np.median(ts))))
    print(f"N={N} L={L} {mode}: " + "  ".join(f"wpc{w}={t:.3f}ms({N*N/t/1e9:.0f}e12)" for w, t in res), flush=True)
wp = [12, 16, 20, 24, 32, 40, 48, 64, 96]
run(200000, 64, "knn", wp); run(200000, 64, "eps", wp); run(50000, 32, "eps", wp); run(50000, 32, "knn", wp)
run(1000000, 64, "knn", [16, 24, 32], iters=2) if len(sys.argv) > 1 else None
