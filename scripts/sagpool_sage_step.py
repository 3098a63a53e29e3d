#!/usr/bin/env python3
"""BASELINE config 4 (IMDB-B SAGPool ratio 0.5, h = 128, batch 128) as a FlatTrainer optimiser step replayed from one hipGraph, for
rocprofv3 / replay_trace.py:   python scripts/sagpool_sage_step.py [gcn|sage] [replays]
gcn = the reference's network (Code/sag/network.py); sage = the same network with SAGEConv convs (config 4 as BASELINE words it)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import sag_layers as S, synthetic, message_passing as mp
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep


class D:
    pass


conv = sys.argv[1] if len(sys.argv) > 1 else "sage"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda")
hb = synthetic.host_batch(3, 128, "IMDB-BINARY", 136)
sizes = hb["sizes"]; n = int(sizes.sum()); rp = hb["rowptr"][: n + 1]; col = hb["col"]
dst = np.repeat(np.arange(n), np.diff(rp))
d = D()
d.edge_index = torch.from_numpy(np.stack([col.astype(np.int64), dst.astype(np.int64)])).to(dev)
d.x = torch.ones(n, 1, device=dev)
d.batch = torch.repeat_interleave(torch.arange(128), torch.from_numpy(sizes)).to(dev)
lab = torch.from_numpy(hb["label"]).to(dev)
torch.manual_seed(0)
net = S.Net(1, 128, 2, 0.5, 0.0, use_batch=True, conv=conv).to(dev).train()
tr = FlatTrainer(net, lr=1e-3, clip=2.0, defer_loss=True)
gs = GraphedStep(tr, lambda: mp.nll_loss(net(d), lab), warmup=3)
for _ in range(20):
    gs.step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(gs.stream)
for _ in range(reps):
    gs.step()
e1.record(gs.stream); e1.synchronize()
print("IMDB-B b128 SAGPool(0.5) + %s conv: %.1f us/step, loss %.5f, %s" % (conv, e0.elapsed_time(e1) / reps * 1e3, gs.loss_value(), gs.describe()))
