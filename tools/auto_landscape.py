"""What the library picks by itself (engine, eps path: size rules + the device-side data probe) on seven shapes, timed
beside every forced combination in the same process (profiles/r03_engine_landscape.txt): auto/best is the cost of the
library's own choice, the probe and the gated launches included."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat, synth

def timeit(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

for k in ("PG_ENGINE", "PG_ENGINE_MIN_ROWS", "PG_GATE_FORCE", "PG_PROBE"):
    os.environ.pop(k, None)
rng = np.random.RandomState(1)
cases = [("cfg3 clustered N=200k L=64", synth.clustered_tokens(200000, 64)),
         ("random N=200k L=64", rng.randint(1, 21, size=(200000, 64)).astype(np.uint8)),
         ("dense(one cluster) N=200k L=64", synth.clustered_tokens(200000, 64, members=200000)),
         ("clusters of 64 N=200k L=64", synth.clustered_tokens(200000, 64, members=64)),
         ("cfg2 clustered N=50k L=32", synth.clustered_tokens(50000, 32)),
         ("clustered N=100k L=128", synth.clustered_tokens(100000, 128)),
         ("clustered N=20k L=32", synth.clustered_tokens(20000, 32))]
for name, tok in cases:
    N = tok.shape[0]
    p = nat.pack(torch.from_numpy(tok), bits=5); dev = p.buf.device; cap = 256
    si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev); cl = torch.empty(N, dtype=torch.int32, device=dev)
    out = (torch.empty((N, 16), dtype=torch.int32, device=dev), torch.empty((N, 16), dtype=torch.uint8, device=dev))
    L_ = nat.lib()
    def sym():
        nat._check(L_.pg_eps_slots_sym(nat._ptr(p.buf), p.npad, p.n, p.g * 32, p.bits, nat.CMP_LE, 2.0, cap, nat._ptr(si), nat._ptr(sw),
                                       nat._ptr(cnt), nat._ptr(cl), nat._ptr(nat.workspace(N, dev)), nat._stream()), "sym")
    use_sym = N >= 32768          # _native.eps_graph's rule for the entry point
    rect = lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)
    knn = lambda: nat.knn_graph(p, p, 16, out=out)
    def forced(env, f):
        os.environ.update(env); t = timeit(f)
        for k_ in env: os.environ.pop(k_)
        return t
    # every variant in the same process on the same buffers, two interleaved rounds, the better median of each
    # (the first timing after allocating the buffers is slow whichever variant it is)
    ve = {"AUTO": ({}, sym if use_sym else rect), "valu rect": ({"PG_ENGINE": "valu"}, rect), "mfma rect": ({"PG_ENGINE": "mfma"}, rect)}
    if use_sym:
        ve["valu sym"] = ({"PG_ENGINE": "valu"}, sym); ve["mfma sym"] = ({"PG_ENGINE": "mfma"}, sym)
    vk = {"AUTO": ({}, knn), "valu": ({"PG_ENGINE": "valu"}, knn), "mfma R=1": ({"PG_ENGINE": "mfma", "PG_MM_R": "1"}, knn),
          "mfma R=2": ({"PG_ENGINE": "mfma", "PG_MM_R": "2"}, knn)}
    timeit(sym if use_sym else rect); timeit(knn)
    fe, fk = {}, {}
    for rnd in range(2):
        for k_, (env, f) in ve.items(): fe[k_] = min(fe.get(k_, 1e9), forced(env, f))
        for k_, (env, f) in vk.items(): fk[k_] = min(fk.get(k_, 1e9), forced(env, f))
    te, tk = fe.pop("AUTO"), fk.pop("AUTO")
    be, bk = min(fe, key=fe.get), min(fk, key=fk.get)
    print(f"{name}\n  eps2 slots: AUTO {te:.3f}  best forced {be} {fe[be]:.3f}  auto/best {te / fe[be]:.3f}   ("
          + ", ".join(f"{k_} {v:.3f}" for k_, v in fe.items()) + ")\n"
          f"  knn16:      AUTO {tk:.3f}  best forced {bk} {fk[bk]:.3f}  auto/best {tk / fk[bk]:.3f}   ("
          + ", ".join(f"{k_} {v:.3f}" for k_, v in fk.items()) + ")", flush=True)
