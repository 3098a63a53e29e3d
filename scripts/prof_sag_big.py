#!/usr/bin/env python3
"""SAGPool step at a large batch (IMDB-B x 8192) for rocprofv3 --kernel-trace --stats"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import sag_layers as S, synthetic
dev = torch.device("cuda"); torch.manual_seed(0)
B = int(os.environ.get("B", 8192))
hb4 = synthetic.host_batch(3, B, "IMDB-BINARY", 136)
sizes = hb4["sizes"]; rp = hb4["rowptr"][: int(sizes.sum()) + 1]; col = hb4["col"]
dst = np.repeat(np.arange(int(sizes.sum())), np.diff(rp))
ei = torch.from_numpy(np.stack([col.astype(np.int64), dst.astype(np.int64)])).to(dev)
class D: pass
d = D(); d.x = torch.ones(int(sizes.sum()), 1, device=dev); d.edge_index = ei
d.batch = torch.repeat_interleave(torch.arange(B), torch.from_numpy(sizes)).to(dev)
lab4 = torch.from_numpy(hb4["label"]).to(dev)
net = S.Net(1, 128, 2, 0.5, 0.5, use_batch=True).to(dev).train()
for _ in range(10):
    net.zero_grad(set_to_none=True); torch.nn.functional.nll_loss(net(d), lab4).backward()
torch.cuda.synchronize()
