"""One GPU's block of BASELINE configs[3] (125 000 rows x 1 000 000 columns, kNN 16) a few times - for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prograph_amd import _native as nat, synth
p = nat.pack(torch.from_numpy(synth.clustered_tokens(1_000_000, 64)), bits=5)
out = (torch.empty((125_000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((125_000, 16), dtype=torch.uint8, device=p.buf.device))
for _ in range(4):
    nat.knn_graph(p, p, 16, row0=375_000, nrows=125_000, out=out)
torch.cuda.synchronize()
