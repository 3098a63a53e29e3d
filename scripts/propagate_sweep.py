#!/usr/bin/env python3
"""GCN-normalised aggregation (tsgnn_gcn_propagate_f32, F = 128: the GCNConv propagate of the SAGPool path) over
batch sizes of DD- and IMDB-B-shaped graphs: us per launch (hipGraph-replayed burst, HIP events) and the fraction of the 8 TB/s
HBM spec on §8(d)'s algorithmic bytes (8NF + 4E + 4(N+1), + 8N for the two per-row coefficient arrays that replace the 4E
per-edge weights)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import synthetic, sag_stack as SS, _native as nat
from two_stage_gnn_amd.synthetic import aggregation_bytes

F = 128
dev = torch.device("cuda")


def burst_us(fn, iters):
    s = torch.cuda.current_stream()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); e1.record(s); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for shape, nmax, batches in (("DD", 1000, (32, 256, 2048, 16384)), ("IMDB-BINARY", 136, (128, 2048, 32768))):
        for B in batches:
            hb = synthetic.host_batch(100 + B, B, shape, nmax)
            g, _, _ = synthetic.to_device(hb, dev)
            n = g.n_rows
            dinv, self_w = SS.gcn_coef(g)
            x = torch.randn(g.total_rows, F, device=dev); y = torch.empty_like(x)
            us = burst_us(lambda: nat.call("gcn_propagate_f32", g.rowptr, g.col, dinv, self_w, x, F, 0, None, None, None, y, F, None,
                                           n, F), 200 if n < 100000 else 20)
            nbytes = aggregation_bytes(n, g.nnz, F) + 8 * n
            print("gcn_propagate %-11s B=%6d rows=%8d nnz=%9d: %9.2f us  %6.0f GB/s (%.2f of 8000)" % (
                shape, B, n, g.nnz, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000))
