"""A/B of library builds on the dense workload (N=200k one cluster, L=64, kNN 16) and cfg3:
usage: ab_dense.py lib1.so [lib2.so ...]   (each in a child process: the library is bound at import)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from prograph_amd import _native as nat, synth
def t(f, iters=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for name, tok in (("cfg3", synth.clustered_tokens(200000, 64)), ("dense", synth.clustered_tokens(200000, 64, members=200000))):
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((200000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((200000, 16), dtype=torch.uint8, device=p.buf.device))
    print(f"  {name:6s} kNN16 {t(lambda: nat.knn_graph(p, p, 16, out=out)):8.3f} ms", flush=True)
''' % ROOT
for lib in sys.argv[1:]:
    print(lib, flush=True)
    env = dict(os.environ, PROGRAPH_HIP_LIB=os.path.abspath(lib), PG_ENGINE="mfma")
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
