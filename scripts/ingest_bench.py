#!/usr/bin/env python3
"""f1: dense padded adjacency -> CSR on the GPU (what the drop-in pays once per batch when handed the reference's
adj[B,Nmax,Nmax]); HBM-bound: two passes over 4*B*n_b*n_b bytes (count, fill)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import synthetic
from two_stage_gnn_amd.graph import GraphBatch

hb = synthetic.host_batch(0, 32, "DD", 1000)
x, adj = synthetic.to_dense(hb)
adj = adj.cuda()
for layout, sizes in (("packed", hb["sizes"]), ("padded", None)):
    for _ in range(3):
        g = GraphBatch.from_dense(adj, sizes=sizes, layout=layout)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g = GraphBatch.from_dense(adj, sizes=sizes, layout=layout)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    scanned = 2 * 4 * float((hb["sizes"].astype(float) ** 2).sum()) if layout == "packed" else 2 * 4 * 32 * 1000 * 1000
    print("ingest dense->CSR %s: %.1f us per batch (incl. 1 host sync for nnz), scans %.1f MB -> %.0f GB/s; nnz=%d"
          % (layout, dt * 1e6, scanned / 1e6, scanned / dt / 1e9, g.nnz))
