#!/usr/bin/env python3
"""timing of the fused SAGPool head kernels (B=128, 256->128->64->2)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import _native as nat
dev = torch.device("cuda"); torch.manual_seed(0)
B, D0, D1, D2, C = 128, 256, 128, 64, 2
x = torch.randn(B, D0, device=dev); w1 = torch.randn(D1, D0, device=dev) * .1; w2 = torch.randn(D2, D1, device=dev) * .1; w3 = torch.randn(C, D2, device=dev)
b1 = torch.zeros(D1, device=dev); b2 = torch.zeros(D2, device=dev); b3 = torch.zeros(C, device=dev)
a1 = torch.empty(B, D1, device=dev); a2 = torch.empty(B, D2, device=dev); logp = torch.empty(B, C, device=dev)
dlogp = torch.randn(B, C, device=dev)
dw1 = torch.empty(D1, D0, device=dev); db1 = torch.empty(D1, device=dev); dw2 = torch.empty(D2, D1, device=dev); db2 = torch.empty(D2, device=dev)
dw3 = torch.empty(C, D2, device=dev); db3 = torch.empty(C, device=dev); dx = torch.empty(B, D0, device=dev)
def fwd(): nat.call("mlp3_fwd_f32", x, D0, w1, b1, None, 1.0, w2, b2, w3, b3, B, D0, D1, D2, C, a1, a2, logp)
def bwd(dxt): nat.call("mlp3_bwd_f32", x, D0, w1, w2, w3, a1, a2, logp, dlogp, 1.0, B, D0, D1, D2, C, dw1, db1, dw2, db2, dw3, db3, dxt, D0)
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n // 20): g.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ws = torch.empty(B * (C + D2 + D1), device=dev)
def bwd2(dxt): nat.call("mlp3_bwd2_f32", x, D0, w1, w2, w3, a1, a2, logp, dlogp, 1.0, B, D0, D1, D2, C, dw1, db1, dw2, db2, dw3, db3, dxt, D0, ws)
print("fwd %.1f us" % t(fwd))
print("bwd2 (two launches) full %.1f us per pair" % t(lambda: bwd2(dx)))
print("bwd full %.1f us" % t(lambda: bwd(dx)))
print("bwd no dx %.1f us" % t(lambda: bwd(None)))
