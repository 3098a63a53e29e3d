#!/usr/bin/env python3
"""Row-panel product (tsgnn_rowgemm_f32, K = N = 128, bias + L2 normalise) and its gather-fused variant over batch sizes:
us per launch (hipGraph-replayed burst, HIP events), fp32-MFMA and HBM fractions."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import synthetic, _native as nat

H = 128
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
    for B in (32, 256, 2048, 8192):
        hb = synthetic.tiled_batch(100 + B, B, "DD", 1000)
        g = synthetic.structure_to_device(hb, dev)
        R, n = g.total_rows, g.n_rows
        X = torch.randn(R, H, device=dev); V = torch.empty_like(X); rinv = torch.empty(R, device=dev)
        W = torch.randn(H, H, device=dev) * 0.1; b = torch.randn(H, device=dev)
        us = burst_us(lambda: nat.call("rowgemm_f32", X, H, W, H, 0, b, V, H, rinv, n, H, H, 1, 0), 100 if B <= 256 else 20)
        flops = 2.0 * n * H * H
        nbytes = 2 * 4 * n * H + 4 * H * H + 4 * n
        print("rowgemm   B=%5d rows=%8d: %9.2f us  %6.1f TF (%.2f of 157.3)  %6.0f GB/s (%.2f of 8000)  %s" % (
            B, n, us, flops / us / 1e6, flops / us / 1e6 / 157.3, nbytes / us / 1e3, nbytes / us / 1e3 / 8000, nat.last_kernel()))
        Wt = W.t().contiguous()
        us = burst_us(lambda: nat.call("rowgemm_f32", X, H, Wt, H, 1, None, V, H, None, n, H, H, 0, 0), 100 if B <= 256 else 20)
        print("  dZ=dU.W^T          rows=%8d: %9.2f us  %6.1f TF (%.2f of 157.3)  %s" % (n, us, flops / us / 1e6, flops / us / 1e6 / 157.3,
                                                                                      nat.last_kernel()))
