import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PROGRAPH_HIP_LIB"] = os.path.join(ROOT, "prograph_amd", "libprograph_hip_stats.so")
os.environ["PG_ENGINE"] = "mfma"
import numpy as np, torch
from prograph_amd import _native as nat
lib = nat.lib()
g = np.load(os.path.join(ROOT, "tests/golden/synth_n2085_l64.npz"))
tok = g["tokens"]
p = nat.pack(torch.from_numpy(np.ascontiguousarray(tok)), bits=5)
lib.pg_debug_stats(None, 1)
indptr, idx, w = [x.cpu().numpy() for x in nat.eps_graph(p, p, nat.CMP_LE, 1, cap=256)]
buf = (ctypes.c_ulonglong * 24)()
lib.pg_debug_stats(buf, 1)
print("eps: super-tiles", buf[0], "ring mismatches tile0", buf[22], "tiles1-3", buf[23], "counts ok:", np.array_equal(indptr, g["eps1_indptr"]))
