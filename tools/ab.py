"""A/B timing of several builds of libprograph_hip.so on one box: tools/ab.py libA.so libB.so[:ENV=VAL,ENV=VAL] ...
Each build runs in its own child process (PROGRAPH_HIP_LIB), the rounds are interleaved."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from prograph_amd import _native as nat, synth
CASES = eval(os.environ.get("AB_CASES", "None"))
def run(N, L, bits, mode, iters=7):
    tok = synth.clustered_tokens(N, L)
    p = nat.pack(torch.from_numpy(tok), bits=bits)
    cap = 256; dev = p.buf.device
    si = torch.empty(N*cap, dtype=torch.int32, device=dev); sw = torch.empty(N*cap, dtype=torch.uint8, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev)
    out = (torch.empty((N,16), dtype=torch.int32, device=dev), torch.empty((N,16), dtype=torch.uint8, device=dev))
    f = (lambda: nat.eps_slots_only(p, p, nat.CMP_LE, 2, 0, N, cap, si, sw, cnt)) if mode == "eps" else (lambda: nat.knn_graph(p, p, 16, out=out))
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
cases = CASES or [(200000, 64, 5, "knn"), (200000, 64, 5, "eps"), (50000, 32, 5, "knn"), (50000, 32, 5, "eps"), (200000, 64, 8, "knn"), (100000, 128, 5, "knn")]
print(" ".join("%%s%%d/%%d/%%d=%%.3f" %% (m, N, L, b, run(N, L, b, m)) for N, L, b, m in cases), flush=True)
""" % ROOT

libs = sys.argv[1:]
for rnd in range(2):
    for spec in libs:
        lib, _, extra = spec.partition(":")
        env = dict(os.environ, PROGRAPH_HIP_LIB=os.path.abspath(lib))
        for kv in filter(None, extra.split(",")):
            env[kv.split("=")[0]] = kv.split("=", 1)[1]
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        print(f"[{os.path.basename(spec)}] {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
