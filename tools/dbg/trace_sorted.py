import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
tok = synth.clustered_tokens(200000, 64)
mode = os.environ.get("TR_ORDER", "sorted")
tok = tok[np.lexsort(tok.T[::-1])] if mode == "sorted" else tok[np.random.RandomState(3).permutation(len(tok))]
p = nat.pack(torch.from_numpy(np.ascontiguousarray(tok)), bits=5)
out = (torch.empty((200000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((200000, 16), dtype=torch.uint8, device=p.buf.device))
for _ in range(3): nat.knn_graph(p, p, 16, out=out)
torch.cuda.synchronize()
ws = nat.workspace(200000, p.buf.device)
