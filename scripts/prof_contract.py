#!/usr/bin/env python3
"""DiffPool level-1 contraction X' = S^T Z, A' = S^T A S on the cfg-5 batch (DD b16, 512 -> 64), 30 eager calls, for
rocprofv3 --pmc / --kernel-trace (MFMA-busy counters of the ragged slab kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import synthetic, diffpool as dp
dev = torch.device("cuda"); torch.manual_seed(0)
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
Sm = torch.softmax(torch.randn(g5.total_rows, 64, device=dev), -1); Sm[g5.n_rows:] = 0
Z = torch.randn(g5.total_rows, 192, device=dev)
for _ in range(30):
    dp.diffpool_contract_rows(Sm, Z, g5)
torch.cuda.synchronize()
