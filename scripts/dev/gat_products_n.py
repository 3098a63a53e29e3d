#!/usr/bin/env python3
"""the five dense products of the GAT b32 step (8,518 rows): this library's kernels vs torch.mm (rocBLAS / hipBLASLt), hipGraph bursts"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from two_stage_gnn_amd import _native as nat, gat_fused as gf
dev = torch.device("cuda"); torch.manual_seed(0)
S = torch.cuda.Stream(); torch.cuda.set_stream(S)
R = int(os.environ.get("ROWS", "8518"))
def burst(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=S):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(S); g.replay(); e1.record(S); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (K, N) in [(92, 256), (256, 256), (92, 264), (256, 264)]:
    x = torch.randn(R, K, device=dev); w = torch.randn(K, N, device=dev); hp = torch.empty(R, N, device=dev)
    dhp = torch.randn(R, N, device=dev); dx = torch.empty(R, K, device=dev)
    fl = 2.0 * R * K * N
    t = burst(lambda: nat.call("rowgemm_f32", x, K, w, N, 0, None, hp, N, None, R, K, N, 0, 0))
    tt = burst(lambda: torch.mm(x, w, out=hp))
    print("fwd  [%d x %d] . [%d x %d]: lib %.1f us (%.1f TF)  torch.mm %.1f us (%.1f TF)" % (R, K, K, N, t, fl / t / 1e6, tt, fl / tt / 1e6))
    t = burst(lambda: nat.call("rowgemm_f32", dhp, N, w, N, 1, None, dx, K, None, R, N, K, 0, 0))
    tt = burst(lambda: torch.mm(dhp, w.t(), out=dx))
    print("dx   [%d x %d] . [%d x %d]^T: lib %.1f us (%.1f TF)  torch.mm %.1f us (%.1f TF)" % (R, N, K, N, t, fl / t / 1e6, tt, fl / tt / 1e6))
    t = burst(lambda: gf.wgrad_blocks(x, K, dhp))
    dw = torch.empty(K, N, device=dev)
    tt = burst(lambda: torch.mm(x.t(), dhp, out=dw))
    print("dW   [%d x %d]^T . [%d x %d]: lib %.1f us (%.1f TF, 2 launches)  torch.mm %.1f us (%.1f TF)" % (R, K, R, N, t, fl / t / 1e6, tt, fl / tt / 1e6))
