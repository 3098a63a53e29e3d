#!/usr/bin/env python3
"""the surface-(B) GAT optimiser step alone (pyg.GatNet, DD b32, 2 layers x 4 heads x 64; one hipGraph), for rocprofv3 / replay_trace.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_stage_gnn_amd import pyg, synthetic
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep


class D:
    pass


dev = torch.device("cuda")
hb = synthetic.host_batch(2, 32, "DD", 1000)
d = D()
d.x, d.edge_index, d.batch, lab = synthetic.to_pyg(hb, dev)
torch.manual_seed(0)
net = pyg.GatNet(89, 64, 2, heads=4, num_layers=2).to(dev).train()
tr = FlatTrainer(net, lr=5e-4, clip=2.0, defer_loss=True)
gs = GraphedStep(tr, lambda: net.loss(d, lab), warmup=3)
for _ in range(20):
    gs.step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(gs.stream)
for _ in range(50):
    gs.step()
e1.record(gs.stream); e1.synchronize()
print("GatNet DD b32: %.1f us/step, loss %.5f, %s" % (e0.elapsed_time(e1) / 50 * 1e3, gs.loss_value(), gs.describe()))
