#!/usr/bin/env python3
"""Soak: many replays of the DiffPool (config 5: device-wide barriers with a bounded spin inside dense_stack_*), GAT (config 3) and headline
steps from their hipGraphs, the device error word checked every 1,000 replays and the loss at the end.   python3 scripts/soak.py [replays=20000]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, gat_encoders as G, synthetic, message_passing as mp
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep
dev = torch.device("cuda"); torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
class A: bias = True
def soak(tag, model, loss_fn, B, lr):
    tr = FlatTrainer(model, lr=lr, clip=2.0, defer_loss=True)
    gs = GraphedStep(tr, loss_fn, warmup=3)
    t0 = time.perf_counter()
    for k in range(N):
        gs.step()
        if (k + 1) % 1000 == 0:
            mp.check_device_errors()                       # synchronises; raises if a bounded barrier gave up
    loss = gs.loss_value()
    dt = time.perf_counter() - t0
    assert loss == loss and abs(loss) < 1e6, loss
    print("%s: %d replayed optimiser steps, %.1f s (%.1f us/step incl. the checks), final loss %.5f, no device error" % (tag, N, dt, dt / N * 1e6, loss))
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
dpm = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                              assign_input_dim=89, final_dim="number_classes").to(dev)
soak("cfg5 DiffPool b16", dpm, lambda: dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5), 16, 1e-4)
hb3 = synthetic.host_batch(2, 32, "DD", 1000)
x3d, adj3 = synthetic.to_dense(hb3)
gat = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes", per_graph_features=True).to(dev)
x3, g3 = gat.packed_batch(x3d.to(dev), adj3.to(dev), hb3["sizes"])
lab3 = torch.from_numpy(hb3["label"]).to(dev)
soak("cfg3 GAT b32", gat, lambda: gat.loss(gat(x3, g3)[1], lab3), 32, 1e-4)
hb = synthetic.host_batch(0, 32, "DD", 1000)
g, x, lab = synthetic.to_device(hb, dev)
m = E.GcnEncoderGraph(89, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
soak("headline SAGE b32", m, lambda: m.loss(m(x, g)[1], lab), 32, 1e-4)
# round 4: config 4 (the reference's SAGPool network and the same with SAGEConv convs) and surface (B)'s SageNet
import numpy as np
from two_stage_gnn_amd import sag_layers as S, pyg
hb4 = synthetic.host_batch(3, 128, "IMDB-BINARY", 136)
sizes4 = hb4["sizes"]; n4 = int(sizes4.sum()); rp4 = hb4["rowptr"][: n4 + 1]
class D: pass
d4 = D()
d4.edge_index = torch.from_numpy(np.stack([hb4["col"].astype(np.int64), np.repeat(np.arange(n4), np.diff(rp4)).astype(np.int64)])).to(dev)
d4.x = torch.ones(n4, 1, device=dev)
d4.batch = torch.repeat_interleave(torch.arange(128), torch.from_numpy(sizes4)).to(dev)
lab4 = torch.from_numpy(hb4["label"]).to(dev)
for conv in ("gcn", "sage"):
    net4 = S.Net(1, 128, 2, 0.5, 0.0, use_batch=True, conv=conv).to(dev).train()
    soak("cfg4 SAGPool + %s conv b128" % conv, net4, lambda: mp.nll_loss(net4(d4), lab4), 128, 1e-4)
dB = D()
dB.x, dB.edge_index, dB.batch, labB = synthetic.to_pyg(hb, dev)
netB = pyg.SageNet(89, 128, 2, num_layers=3).to(dev).train()
soak("surface B SageNet DD b32", netB, lambda: mp.nll_loss(netB(dB), labB), 32, 1e-4)
