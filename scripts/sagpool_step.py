#!/usr/bin/env python3
"""BASELINE config 4 (IMDB-B SAGPool ratio 0.5, h = 128, batch 128): fwd + bwd step replayed from a hipGraph, with the launch
names of one step (for scripts/replay_trace.py: the in-graph timeline)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import sag_layers as S, synthetic, message_passing as mp, _native as nat
dev = torch.device("cuda"); torch.manual_seed(0)
St = torch.cuda.Stream(); torch.cuda.set_stream(St)
hb4 = synthetic.host_batch(3, 128, "IMDB-BINARY", 136)
sizes = hb4["sizes"]; rp = hb4["rowptr"][: int(sizes.sum()) + 1]; col = hb4["col"]
dst = np.repeat(np.arange(int(sizes.sum())), np.diff(rp))
ei = torch.from_numpy(np.stack([col.astype(np.int64), dst.astype(np.int64)])).to(dev)
class D: pass
d = D(); d.x = torch.ones(int(sizes.sum()), 1, device=dev); d.edge_index = ei
d.batch = torch.repeat_interleave(torch.arange(128), torch.from_numpy(sizes)).to(dev)
lab4 = torch.from_numpy(hb4["label"]).to(dev)
net = S.Net(1, 128, 2, 0.5, 0.5, use_batch=True).to(dev).train()
def step():
    # the library's nll_loss, deferred into the head's backward (valid after it), the backward seeded with d loss = 1: what
    # FlatTrainer(defer_loss=True) does for its steps; TORCH_LOSS=1: F.nll_loss + a default backward (four more torch launches)
    net.zero_grad(set_to_none=True)
    if os.environ.get("TORCH_LOSS") == "1":
        torch.nn.functional.nll_loss(net(d), lab4).backward()
    else:
        with mp.deferred_loss():
            mp.nll_loss(net(d), lab4).backward(gradient=mp.unit_seed(dev))
for _ in range(3): step()
torch.cuda.synchronize(); mp.check_device_errors()
nat.trace = []; step(); names = [t[2] or t[0] for t in nat.trace]; ent = [t[0] for t in nat.trace]; nat.trace = None
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=St):
    step()
gr.replay(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(St)
for _ in range(50): gr.replay()
e1.record(St); e1.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
mp.check_device_errors()
print("cfg4 IMDB-B SAGPool(0.5) h128 b128: %.0f us/step from one hipGraph -> %.0f graphs/s ; %d library launches per step" % (us, 128 / us * 1e6, len(ent)))
print("   " + ", ".join(ent))
