"""Event counters of the MFMA engine (debug build of the library with -DPG_MM_STATS:
   make -C prograph_amd/csrc BUILD=build_stats OUT=../libprograph_hip_stats.so EXTRA=-DPG_MM_STATS).
   usage: mm_stats.py [case ...]   cases: cfg3 random dense c64 cfg2 cfg3sorted cfg3shuffled"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PROGRAPH_HIP_LIB"] = os.path.join(ROOT, "prograph_amd", "libprograph_hip_stats.so")
os.environ["PG_ENGINE"] = "mfma"
import numpy as np, torch
from prograph_amd import _native as nat, synth

NAMES = ["L1 super-tiles", "with candidates", "dense tiles, every distance", "dense runs", "dense super-tiles", "dense row-steps past the bound",
         "candidates queued", "flushes", "insertions/matches", "resweep super-tiles", "passes", "dense tiles, folded bound",
         "x64 cycles in flush", "x64 cycles in kNN insertion loops", "x64 cycles in folded tiles (incl. their flushes)", "x64 cycles in passes",
         "x64 cycles in scan (hot loop)", "x64 cycles in slow_mfma (incl. its flushes)", "scan calls", "bias refreshes", "queueing turns",
         "x64 cycles of pass setup", "-", "-"]
NST = 24
lib = nat.lib()
def stats(reset=True):
    buf = (ctypes.c_ulonglong * NST)()
    lib.pg_debug_stats(buf, 1 if reset else 0)
    return list(buf)
def timed(f):
    f(); torch.cuda.synchronize(); stats()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), stats()
rng = np.random.RandomState(1)
CASES = {"cfg3": lambda: synth.clustered_tokens(200000, 64), "random": lambda: rng.randint(1, 21, size=(200000, 64)).astype(np.uint8),
         "dense": lambda: synth.clustered_tokens(200000, 64, members=200000), "c64": lambda: synth.clustered_tokens(200000, 64, members=64),
         "cfg2": lambda: synth.clustered_tokens(50000, 32), "dense50k": lambda: synth.clustered_tokens(50000, 64, members=50000),
         # cfg3's sequences in another order: cluster mates adjacent / scattered without the generator's regular stride
         "cfg3sorted": lambda: (lambda t: np.ascontiguousarray(t[np.lexsort(t.T[::-1])]))(synth.clustered_tokens(200000, 64)),
         "cfg3shuffled": lambda: (lambda t: np.ascontiguousarray(t[np.random.RandomState(3).permutation(len(t))]))(synth.clustered_tokens(200000, 64))}
for name in (sys.argv[1:] or ["cfg3", "dense", "random"]):
    tok = CASES[name]()
    N = tok.shape[0]
    p = nat.pack(torch.from_numpy(tok), bits=5)
    dev = p.buf.device
    cap = 256
    si = torch.empty(N * cap, dtype=torch.int32, device=dev); sw = torch.empty(N * cap, dtype=torch.uint8, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev)
    out = (torch.empty((N, 16), dtype=torch.int32, device=dev), torch.empty((N, 16), dtype=torch.uint8, device=dev))
    whats = os.environ.get("MM_STATS_WHAT", "knn16,eps2").split(",")
    for what, f in (("knn16", lambda: nat.knn_graph(p, p, 16, out=out)), ("eps2", lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt))):
        if what not in whats:
            continue
        ms, st = timed(f)
        print(f"== {name} {what}: {ms:.3f} ms")
        for n, v in zip(NAMES, st):
            if n != "-":
                print(f"   {n:28s} {v:>14d}   per pass {v / max(st[10], 1):10.1f}")
    sys.stdout.flush()
