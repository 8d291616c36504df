"""MFMA engine (pg_mm.h) vs VALU engine (pg_nsq.h): eps slots and kNN on random / clustered / dense
data, and the optimistic kNN cap on the new engine.   usage: mm_probe.py [landscape|guess]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth

def timeit(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

what = sys.argv[1] if len(sys.argv) > 1 else "landscape"
rng = np.random.RandomState(1)
if what == "landscape":
    cases = [("cfg3 clustered N=200k L=64", synth.clustered_tokens(200000, 64)),
             ("random N=200k L=64", rng.randint(1, 21, size=(200000, 64)).astype(np.uint8)),
             ("dense(one cluster) N=200k L=64", synth.clustered_tokens(200000, 64, members=200000)),
             ("clusters of 64 N=200k L=64", synth.clustered_tokens(200000, 64, members=64)),
             ("cfg2 clustered N=50k L=32", synth.clustered_tokens(50000, 32)),
             ("clustered N=100k L=128", synth.clustered_tokens(100000, 128)),
             ("clustered N=20k L=32", synth.clustered_tokens(20000, 32))]
    for name, tok in cases:
        N = tok.shape[0]
        p = nat.pack(torch.from_numpy(tok), bits=5)
        dev = p.buf.device
        cap = 256
        si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
        cnt = torch.empty(N, dtype=torch.int32, device=dev); cl = torch.empty(N, dtype=torch.int32, device=dev)
        out = (torch.empty((N, 16), dtype=torch.int32, device=dev), torch.empty((N, 16), dtype=torch.uint8, device=dev))
        L_ = nat.lib()
        def sym():
            nat._check(L_.pg_eps_slots_sym(nat._ptr(p.buf), p.npad, p.n, p.g * 32, p.bits, nat.CMP_LE, 2.0, cap, nat._ptr(si), nat._ptr(sw),
                                           nat._ptr(cnt), nat._ptr(cl), nat._ptr(nat.workspace(N, dev)), nat._stream()), "sym")
        line = [name]
        for eng in ("valu", "mfma"):
            os.environ["PG_ENGINE"] = eng
            te = timeit(lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)); nnz = int(cnt.to(torch.int64).sum())
            ts = timeit(sym)
            tk = timeit(lambda: nat.knn_graph(p, p, 16, out=out))
            line.append(f"{eng}: eps2 {te:.3f} (nnz {nnz}) sym {ts:.3f} knn16 {tk:.3f}")
        print(" | ".join(line), flush=True)
else:
    os.environ["PG_ENGINE"] = "mfma"
    for N, L, members in ((200000, 64, 256), (200000, 64, 64), (200000, 64, 32), (100000, 128, 256), (50000, 32, 256)):
        tok = synth.clustered_tokens(N, L, members=members)
        p = nat.pack(torch.from_numpy(tok), bits=5)
        out = (torch.empty((N, 16), dtype=torch.int32, device=p.buf.device), torch.empty((N, 16), dtype=torch.uint8, device=p.buf.device))
        res = []
        for g in ("0", "3", "4", "5", "6", "7", "8", "10"):
            os.environ["PG_KNN_GUESS"] = g
            res.append((g, timeit(lambda: nat.knn_graph(p, p, 16, out=out))))
        print(f"N={N} L={L} members={members}: " + "  ".join(f"G{g}={t:.3f}" for g, t in res), flush=True)
