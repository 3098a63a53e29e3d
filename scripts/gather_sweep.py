#!/usr/bin/env python3
"""The fused gather kernel (tsgnn_gather_rowgemm_f32, K = N = 128) over batch sizes; run with TSGNN_ROWGEMM_KS2=0 / 1 to compare
the one-group and the two-group (split-K) row-panel kernels at and beyond one block per CU."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from two_stage_gnn_amd import synthetic
dev = torch.device("cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for B in (8, 32, 36, 64, 256, 2048):
        hb = synthetic.host_batch(100 + B, B, "DD", 1000)
        g, _, _ = synthetic.to_device(hb, dev)
        ms, fl, nb = bench.fused_layer_probe(g, 128, iters=200 if B <= 256 else 20)
        print("gather_rowgemm KS2=%s B=%5d rows=%8d panels=%6d: %9.2f us  %6.1f TF (%.3f of 157.3)  %6.0f GB/s" % (
            os.environ.get("TSGNN_ROWGEMM_KS2", "1"), B, g.n_rows, (g.n_rows + 31) // 32, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3, nb / ms / 1e6))
