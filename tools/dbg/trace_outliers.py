import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prograph_amd import _native as nat, synth
share = float(os.environ.get("TR_SHARE", "0.001"))
tok = synth.clustered_tokens(200000, 64); rng = np.random.RandomState(5); n = int(share * len(tok))
tok[rng.choice(len(tok), n, replace=False)] = rng.randint(1, 21, size=(n, 64)).astype(np.uint8)
p = nat.pack(torch.from_numpy(tok), bits=5)
out = (torch.empty((200000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((200000, 16), dtype=torch.uint8, device=p.buf.device))
for _ in range(3): nat.knn_graph(p, p, 16, out=out)
torch.cuda.synchronize()
