#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import gat_encoders as G, synthetic
from two_stage_gnn_amd.graph import GraphBatch
dev = torch.device("cuda"); torch.manual_seed(0)
hb1 = synthetic.host_batch(2, 1, "DD", 1000)
x1, adj1 = synthetic.to_dense(hb1)
gat = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes").to(dev)
x1, adj1 = x1.to(dev), adj1.to(dev)
gpad = GraphBatch.from_dense(adj1, layout="padded"); gpad.transpose_map()
lab1 = torch.tensor([1], device=dev)
for _ in range(20):
    gat.zero_grad(set_to_none=True); gat.loss(gat(x1, gpad)[1], lab1).backward()
torch.cuda.synchronize()
