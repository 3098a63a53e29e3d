#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the DD b32 shapes (hipGraph-captured bursts, HIP events)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import message_passing as mp, synthetic, _native as nat
from two_stage_gnn_amd.graph import GraphBatch


def burst_us(fn, iters=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(iters):
                fn()
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s); gr.replay(); e1.record(s); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    B = int(os.environ.get("B", 32)); H = 128
    hb = synthetic.host_batch(0, B, "DD", 1000)
    g, x, label = synthetic.to_device(hb, torch.device("cuda"))
    R = g.total_rows
    print("rows", R, "nnz", g.nnz)
    X = torch.randn(R, H, device="cuda"); Y = torch.empty_like(X)
    W = torch.randn(H, H, device="cuda") * 0.1; b = torch.randn(H, device="cuda")
    rinv = torch.empty(R, device="cuda")
    res = {}
    res["empty(memset 4B)"] = burst_us(lambda: rinv[:1].zero_())
    res["spmm F=128"] = burst_us(lambda: mp.spmm_raw(g.rowptr, g.col, None, X, R, out=Y))
    g.ell()
    res["spmm ELL F=128"] = burst_us(lambda: mp.spmm_ell(g, X, out=Y))
    res["spmm F=92"] = burst_us(lambda: mp.spmm_raw(g.rowptr, g.col, None, x, R, out=torch.empty_like(x)))
    res["linear_l2norm 128x128"] = burst_us(lambda: nat.call("linear_l2norm_f32", X, H, W, H, b, Y, H, rinv, R, H, H, 1, 0))
    res["rowgemm fwd 128x128 norm"] = burst_us(lambda: nat.call("rowgemm_f32", X, H, W, H, 0, b, Y, H, rinv, R, H, H, 1, 0))
    dz = torch.empty_like(X)
    res["rowgemm dZ (trans)"] = burst_us(lambda: nat.call("rowgemm_f32", X, H, W, H, 1, None, dz, H, None, R, H, H, 0, 0))
    res["gemm dZ=dU.W^T"] = burst_us(lambda: mp.gemm(X, H, 1, W, 1, H, dz, H, 1, R, H, H))
    res["gemm dW splitk"] = burst_us(lambda: mp.gemm_tn_splitk(X, H, Y))
    res["linear_wgrad (dW+db)"] = burst_us(lambda: mp.linear_wgrad(X, H, Y, True))
    res["colsum"] = burst_us(lambda: mp.colsum(X))
    mean = torch.empty(g.nmax, device="cuda"); rstd = torch.empty(g.nmax, device="cuda")
    res["bn_slots fwd (stats+apply)"] = burst_us(lambda: nat.call("bn_slots_fwd_f32", g.graph_ptr, g.slot_count, g.row_slot, g.B, g.nmax, g.n_rows, g.n_ghost, X, H, H, 1, 1, mean, rstd, Y, H))
    m1 = torch.empty(g.nmax, device="cuda"); m2 = torch.empty(g.nmax, device="cuda")
    res["bn_slots bwd (stats+apply)"] = burst_us(lambda: nat.call("bn_slots_bwd_f32", g.graph_ptr, g.slot_count, g.row_slot, g.B, g.nmax, g.n_rows, g.n_ghost, X, H, Y, H, H, 1, 1, mean, rstd, m1, m2, dz, H))
    out = torch.empty(g.B, H, device="cuda"); arg = torch.empty(g.B, H, dtype=torch.int32, device="cuda"); ws = mp._readout_ws(g, H, torch.device("cuda"))
    res["readout fwd (one launch)"] = burst_us(lambda: nat.call("readout_max_fwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, X, H, H, 0, ws, out, H, arg))
    res["l2norm_bwd"] = burst_us(lambda: nat.call("l2norm_bwd_f32", X, H, Y, H, rinv, dz, H, R, H))
    res["torch.mm 9151x128x128 (rocBLAS ref)"] = burst_us(lambda: torch.mm(X, W, out=Y))
    res["torch.mm X^T.Y (rocBLAS ref)"] = burst_us(lambda: torch.mm(X.t(), Y))
    for k, v in res.items():
        print("%-40s %8.2f us" % (k, v))


if __name__ == "__main__" and not os.environ.get("EXTRA"):
    main()


def extra():
    """launch-overhead probes: alternating DIFFERENT trivial kernels, and the fused-stack kernels"""
    dev = torch.device("cuda")
    a = torch.zeros(256, device=dev); b = torch.zeros(256, device=dev)
    def alt():
        a.zero_(); nat.call("relu_fwd_f32", a, 256, b); b.add_(1.0); nat.call("relu_bwd_f32", a, b, 256, a)
    print("%-40s %8.2f us per kernel" % ("4 different trivial kernels alternating", burst_us(alt, 50) / 4))
    big = torch.zeros(9151 * 128, device=dev); big2 = torch.zeros_like(big)
    def alt_big():
        nat.call("relu_fwd_f32", big, big.numel(), big2); nat.call("relu_bwd_f32", big2, big, big.numel(), big)
    print("%-40s %8.2f us per kernel" % ("2 kernels alternating, 4.7 MB in/out each", burst_us(alt_big, 50) / 2))
    hb = synthetic.host_batch(0, 32, "DD", 1000)
    g, x, label = synthetic.to_device(hb, dev)
    R, H = g.total_rows, 128
    V = torch.randn(R, H, device=dev); Y = torch.empty_like(V); mean = torch.empty(g.nmax, device=dev); rstd = torch.empty(g.nmax, device=dev)
    print("%-40s %8.2f us" % ("slot_bn_fwd", burst_us(lambda: nat.call("slot_bn_fwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, V, H, H, 1, mean, rstd, Y, H, None, 0))))
    packed = torch.zeros(g.B * H, dtype=torch.int64, device=dev)
    print("%-40s %8.2f us" % ("readout_partial", burst_us(lambda: nat.call("readout_partial_f32", g.graph_ptr, g.B, g.nmax, g.n_rows, g.n_ghost, Y, H, H, packed))))
    out = torch.empty(g.B, H, device=dev); arg = torch.empty(g.B * H, dtype=torch.int32, device=dev)
    nat.call("readout_decode_layers_f32", packed, g.B, 1, H, H, out, H, arg)
    rinv = torch.ones(R, device=dev); du = torch.empty_like(V); dxs = torch.randn(R, H, device=dev); dout = torch.randn(g.B, H, device=dev)
    print("%-40s %8.2f us" % ("slot_post_bwd (bn)", burst_us(lambda: nat.call("slot_post_bwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, V, H, dxs, H, dout, H, arg, H, 1, 1, mean, rstd, rinv, du, H))))
    print("%-40s %8.2f us" % ("slot_post_bwd (last, no dxs)", burst_us(lambda: nat.call("slot_post_bwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, V, H, None, 0, dout, H, arg, H, 0, 0, None, None, rinv, du, H))))


if os.environ.get("EXTRA"):
    extra()
