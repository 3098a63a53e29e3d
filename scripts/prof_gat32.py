#!/usr/bin/env python3
"""cfg 3 batched (32 DD graphs, per-graph features) eager steps for rocprofv3 --kernel-trace --stats"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import gat_encoders as G, synthetic
from two_stage_gnn_amd.graph import GraphBatch
dev = torch.device("cuda"); torch.manual_seed(0)
hb32 = synthetic.host_batch(2, 32, "DD", 1000)
x32, adj32 = synthetic.to_dense(hb32)
gat32 = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes", per_graph_features=True).to(dev)
adj32d = adj32.to(dev)
x32, g32 = gat32.packed_batch(x32.to(dev), adj32d, hb32["sizes"])
lab32 = torch.from_numpy(hb32["label"]).to(dev)
for _ in range(20):
    gat32.zero_grad(set_to_none=True); gat32.loss(gat32(x32, g32)[1], lab32).backward()
torch.cuda.synchronize()
