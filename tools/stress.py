"""Randomised stress of the all-pairs engine against the C oracle (not part of the test-suite: minutes).
usage: tools/stress.py [iterations] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prograph_amd import _native as nat
from oracle import c_oracle as C

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"          # larger problems (the symmetric eps path switches on by itself)
t0 = time.time()
for it in range(iters):
    # big: sizes around the kNN plan's round boundaries (column pieces: 68 000 / 100 000 rows with 32-row passes, 135 000 /
    # 139 000 with 64-row passes), the probe and its gated launches (engine "auto", >= 65 536 columns)
    N = int(rng.choice([40000, 68000, 100000, 135000, 139000])) if BIG else int(rng.choice([300, 1000, 2500, 6000, 12000, 20000]))
    amax = int(rng.choice([2, 4, 20, 31, 200]))
    L = int(rng.randint(4, 255 if amax <= 31 else 129))
    ncl = max(1, N // int(rng.choice([8, 24, 100, 400, 3000] + ([N] if BIG else []))))   # (big: sometimes ONE cluster - the probe's third outcome)
    base = rng.randint(0, amax + 1, size=(ncl, L))
    tok = base[rng.randint(0, ncl, size=N)].copy()
    nm = rng.randint(0, int(rng.choice([2, 4, 12])), size=N)
    for t in range(int(nm.max())):
        act = np.nonzero(nm > t)[0]
        tok[act, rng.randint(0, L, size=len(act))] = rng.randint(0, amax + 1, size=len(act))
    nloose = int(N * rng.choice([0, 0, 0.1, 0.5]))
    if nloose:
        tok[rng.choice(N, nloose, replace=False)] = rng.randint(0, amax + 1, size=(nloose, L))
    order = rng.rand()
    tok = tok[rng.permutation(N)].astype(np.uint8) if order < 0.4 else (np.ascontiguousarray(tok[np.lexsort(tok.T[::-1])]).astype(np.uint8) if order < 0.6 else tok.astype(np.uint8))
    bits = 5 if (amax <= 31 and (L > 128 or rng.rand() < 0.7)) else 8
    os.environ["PG_KNN_GUESS"] = str(rng.choice([0, 2, 5, 8, 8, 8, 20]))
    if BIG and rng.rand() < 0.6: os.environ.pop("PG_KNN_GUESS")
    os.environ["PG_LB_FILTER"] = str(rng.choice([0, 1, 1, 1, 2]))
    os.environ["PG_EPS_SYM"] = "auto" if BIG else str(rng.choice([0, 1]))
    # both engines; on the MFMA engine also the density rules of its filter hierarchy (level-2 runs and direct
    # runs start early / late / never) and the fill pass for overflowed rows (always / rarely)
    os.environ["PG_ENGINE"] = str(rng.choice(["valu", "mfma", "mfma", "mfma"]))
    if BIG and rng.rand() < 0.5: os.environ.pop("PG_ENGINE")      # the library's own choice: probe + gated launches
    os.environ["PG_MM_PIECES"] = str(rng.choice([0, 0, 2, 5, 8]))  # (0: the library's own count)
    os.environ["PG_EPS_ORDERED"] = str(rng.choice([0, 0, 1]))
    os.environ["PG_MM_L1"] = str(rng.choice([1, 16, 96, 96, 200, 257]))
    os.environ["PG_MM_L2"] = str(rng.choice([1, 8, 48, 48, 65]))
    os.environ["PG_MM_RUN"] = str(rng.choice([1, 2, 8, 8, 64]))
    os.environ["PG_FILL_MIN_ROWS"] = str(rng.choice([0, 8, 8, 1000000]))
    k = int(rng.choice([1, 5, 16, 16, 19, 40] if BIG else [1, 5, 16, 40, 63, 90]))
    lo = int(rng.randint(0, N // 2)); nr = int(rng.randint(1, N - lo + 1)) if rng.rand() < 0.5 else None
    if nr is None: lo = 0
    if os.environ.get("STRESS_ONLY") and int(os.environ["STRESS_ONLY"]) != it:      # replay one iteration of a run
        rng.choice([1, 2, 3, 6, L // 2]); (rng.choice([0, 0, 1, 2, 3, 4]) if N <= 2500 else None); rng.choice([4, 64, 512])
        continue
    p = nat.pack(torch.from_numpy(tok), bits=bits)
    if os.environ.get("STRESS_ONLY"):
        print(f"replaying it={it}: N={N} L={L} bits={bits} k={k} row0={lo} nrows={nr} engine={os.environ.get('PG_ENGINE', 'auto')}", flush=True)
    idx, d = nat.knn_graph(p, p, k, row0=lo, nrows=nr)
    ridx, rd = C.knn(tok, k, row0=lo, nrows=(N - lo if nr is None else nr), fast=BIG)
    ok1 = np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d.cpu().numpy(), rd)
    eps = int(rng.choice([1, 2, 3, 6, L // 2]))
    cmp = int(rng.choice([nat.CMP_LE, nat.CMP_LE, nat.CMP_LT, nat.CMP_EQ, nat.CMP_GE, nat.CMP_GT])) if N <= 2500 else nat.CMP_LE
    cap = int(rng.choice([4, 64, 512]))
    ip, ix, w = nat.eps_graph(p, p, cmp, eps, row0=lo, nrows=nr, cap=cap)
    rip, rix, rw = C.eps_csr(tok, cmp, eps, row0=lo, nrows=(N - lo if nr is None else nr), fast=BIG)
    ok2 = np.array_equal(ip.cpu().numpy(), rip) and np.array_equal(ix.cpu().numpy(), rix) and np.array_equal(w.cpu().numpy(), rw)
    if not (ok1 and ok2):
        print(f"MISMATCH it={it} N={N} L={L} amax={amax} bits={bits} k={k} eps={eps} cmp={cmp} cap={cap} row0={lo} nrows={nr} "
              f"guess={os.environ.get('PG_KNN_GUESS', 'auto')} filter={os.environ['PG_LB_FILTER']} sym={os.environ['PG_EPS_SYM']} engine={os.environ.get('PG_ENGINE', 'auto')} pieces={os.environ['PG_MM_PIECES']} ordered={os.environ['PG_EPS_ORDERED']} "
              f"L1={os.environ['PG_MM_L1']} L2={os.environ['PG_MM_L2']} run={os.environ['PG_MM_RUN']} fill={os.environ['PG_FILL_MIN_ROWS']} knn_ok={ok1} eps_ok={ok2}", flush=True)
        sys.exit(1)
    if it % (2 if BIG else 20) == 0:
        print(f"it {it} ok ({time.time() - t0:.0f}s) N={N} L={L} bits={bits} k={k}", flush=True)
print(f"all {iters} iterations ok in {time.time() - t0:.0f}s")
