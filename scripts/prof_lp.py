#!/usr/bin/env python3
"""f4 link-prediction loss kernels on the cfg-5 batch, for rocprofv3 --kernel-trace --stats"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import synthetic, diffpool as dp
dev = torch.device("cuda"); torch.manual_seed(0)
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
Sm = torch.softmax(torch.randn(g5.total_rows, 64, device=dev), -1); Sm[g5.n_rows:] = 0
Sl = Sm.clone().requires_grad_(True)
for _ in range(20):
    Sl.grad = None; dp.link_pred_loss(Sl, g5).backward()
torch.cuda.synchronize()
