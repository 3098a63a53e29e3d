#!/usr/bin/env python3
"""Row-panel product without the row epilogue, K = N = 256 (GAT projection): us per launch over row counts.
TSGNN_ROWGEMM_COLSPLIT_PANELS=0 disables the column blocks, =100000 forces them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import _native as nat
from kbench import burst_us
for rows in (1000, 8608, 32768, 131072, 555341):
    a = torch.randn(rows, 256, device="cuda"); w = torch.randn(256, 256, device="cuda") * 0.05; c = torch.empty(rows, 256, device="cuda")
    for tb in (0, 1):
        us = burst_us(lambda: nat.call("rowgemm_f32", a, 256, w, 256, tb, None, c, 256, None, rows, 256, 256, 0, 0), iters=50)
        print("rows %7d trans_b %d: %8.2f us  %6.1f TF" % (rows, tb, us, 2.0 * rows * 256 * 256 / us / 1e6))
