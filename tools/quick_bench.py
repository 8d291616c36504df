import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from prograph_amd import _native as nat, synth
def run(N, L, bits, mode, iters=5):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=bits)
    cap = 256
    dev = p.buf.device
    si = torch.empty(N*cap, dtype=torch.int32, device=dev); sw = torch.empty(N*cap, dtype=torch.uint8, device=dev); cnt = torch.empty(N, dtype=torch.int32, device=dev)
    out = (torch.empty((N,16), dtype=torch.int32, device=dev), torch.empty((N,16), dtype=torch.uint8, device=dev))
    f = (lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)) if mode == "eps" else (lambda: nat.knn_graph(p, p, 16, out=out))
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = np.median(ts) / 1e3
    print(f"N={N} L={L} bits={bits} {mode} wpc={os.environ.get("PG_WAVES_PER_CU","32")}: {t*1e3:.2f} ms  {N*N/t:.3e} pairs/s  alg {N*N*L/t/1e12:.2f} TB/s", flush=True)
for wpc in sys.argv[1:] or ["8"]:
    os.environ["PG_WAVES_PER_CU"] = wpc
    for bits in (5, 8):
        run(50000, 32, bits, "eps"); run(50000, 32, bits, "knn")
        run(200000, 64, bits, "eps"); run(200000, 64, bits, "knn")
