#!/usr/bin/env python3
"""Secondary BASELINE.json configs (not the bench line): fwd+bwd time of each model family on its synthetic shape,
plus the DiffPool contraction's MFMA utilisation (flops / time / 157.3 TF fp32-MFMA peak)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from two_stage_gnn_amd import dense_encoders as E, gat_encoders as G, sag_layers as S, synthetic, diffpool as dp, message_passing as mp
from two_stage_gnn_amd.graph import GraphBatch


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for _ in range(iters):
        fn()
    e1.record(torch.cuda.current_stream()); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def graph_us(fn, iters=20):
    """the same step replayed from a hipGraph (no host overhead); None if the step cannot be captured"""
    st = torch.cuda.current_stream()      # the script's one side stream: autograd's AccumulateGrad nodes live on it too
    try:
        with torch.cuda.stream(st):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                fn()
            gr.replay(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(iters):
                gr.replay()
            e1.record(st); e1.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3
    except Exception as e:                                     # noqa
        torch.cuda.synchronize()
        return None


class A:
    bias = True

dev = torch.device("cuda")
torch.manual_seed(0)
# everything (eager timing and captures) runs on ONE side stream: capturing a step whose AccumulateGrad nodes were created on
# another stream ends in a segfault inside torch's capture_end (stream-mismatch warning of torch.autograd.graph)
_S = torch.cuda.Stream()
torch.cuda.set_stream(_S)

# config 2: PROTEINS SAGE 3-layer h=128 batch 64
hb = synthetic.host_batch(1, 64, "PROTEINS", 620)
g, x, label = synthetic.to_device(hb, dev)
m = E.GcnEncoderGraph(3, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
def step_sage():
    m.zero_grad(set_to_none=True)
    with mp.deferred_loss():          # the cross-entropy inside the head's backward, as FlatTrainer(defer_loss=True) runs it
        m.loss(m(x, g)[1], label).backward(gradient=mp.unit_seed(dev))
t = timeit(step_sage)
tg = graph_us(step_sage)
print("cfg2 PROTEINS SAGE-3L h128 b64 (Nmax 620): %.0f us/step eager, %s us/step hipGraph -> %.0f graphs/s" % (t, "%.0f" % tg if tg else "n/a", 64 / (tg or t) * 1e6))

# config 3: DD GAT 2-layer 4-head h=64, batch 32 graphs = 32 sequential B=1 forwards (the reference's GAT batch size is 1)
hb1 = synthetic.host_batch(2, 1, "DD", 1000)
x1, adj1 = synthetic.to_dense(hb1)
gat = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes").to(dev)
x1, adj1 = x1.to(dev), adj1.to(dev)
gpad = GraphBatch.from_dense(adj1, layout="padded"); gpad.transpose_map()
lab1 = torch.tensor([1], device=dev)
def step_gat():
    gat.zero_grad(set_to_none=True)
    with mp.deferred_loss():          # the cross-entropy inside the head's backward, as FlatTrainer(defer_loss=True) runs it
        gat.loss(gat(x1, adj1, hb1["sizes"])[1], lab1).backward(gradient=mp.unit_seed(dev))
t = timeit(step_gat)
tg = graph_us(step_gat)
print("cfg3 DD GAT-2L 4 heads h64, one graph per step (Nmax 1000): %.0f us/step eager, %s us/step hipGraph -> %.0f graphs/s" % (t, "%.0f" % tg if tg else "n/a", 1 / (tg or t) * 1e6))

# config 3, batched: 32 DD graphs in ONE block-diagonal step, every graph with its own features (= 32 reference B = 1 forwards)
hb32 = synthetic.host_batch(2, 32, "DD", 1000)
x32, adj32 = synthetic.to_dense(hb32)
gat32 = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes", per_graph_features=True).to(dev)
adj32d = adj32.to(dev)
x32, g32 = gat32.packed_batch(x32.to(dev), adj32d, hb32["sizes"])        # n_b rows + one ghost representative per graph
lab32 = torch.from_numpy(hb32["label"]).to(dev)
def step_gat32():
    gat32.zero_grad(set_to_none=True)
    with mp.deferred_loss():          # the cross-entropy inside the head's backward, as FlatTrainer(defer_loss=True) runs it
        gat32.loss(gat32(x32, g32)[1], lab32).backward(gradient=mp.unit_seed(dev))
t = timeit(step_gat32)
tg = graph_us(step_gat32)
print("cfg3 DD GAT-2L 4 heads h64, batch 32 in one block-diagonal step (per-graph features, packed rows + 1 ghost row per graph, Nmax 1000): %.0f us/step eager, %s us/step hipGraph -> %.0f graphs/s" % (t, "%.0f" % tg if tg else "n/a", 32 / (tg or t) * 1e6))
del adj32, adj32d

# config 4: IMDB-B SAGPool ratio .5 h=128 batch 128 (PyG per-graph semantics)
hb4 = synthetic.host_batch(3, 128, "IMDB-BINARY", 136)
sizes = hb4["sizes"]; rp = hb4["rowptr"][: int(sizes.sum()) + 1]; col = hb4["col"]
dst = np.repeat(np.arange(int(sizes.sum())), np.diff(rp))
ei = torch.from_numpy(np.stack([col.astype(np.int64), dst.astype(np.int64)])).to(dev)
class D: pass
d = D(); d.x = torch.ones(int(sizes.sum()), 1, device=dev); d.edge_index = ei
d.batch = torch.repeat_interleave(torch.arange(128), torch.from_numpy(sizes)).to(dev)
lab4 = torch.from_numpy(hb4["label"]).to(dev)
net_c = S.Net(1, 128, 2, 0.5, 0.5, use_batch=True, fused=False).to(dev).train()
def step_sag_c():
    net_c.zero_grad(set_to_none=True); torch.nn.functional.nll_loss(net_c(d), lab4).backward()
t = timeit(step_sag_c, iters=10, warm=3)
print("cfg4 IMDB-B SAGPool(0.5) h128 b128, drop-ins composed level by level: %.0f us/step eager (host syncs for k / E'), %.0f graphs/s" % (t, 128 / t * 1e6))
net = S.Net(1, 128, 2, 0.5, 0.5, use_batch=True).to(dev).train()
def step_sag():
    net.zero_grad(set_to_none=True)
    with mp.deferred_loss():                     # the library's nll_loss inside the head's backward (scripts/sagpool_step.py)
        mp.nll_loss(net(d), lab4).backward(gradient=mp.unit_seed(dev))
t = timeit(step_sag, iters=20, warm=3)
tg = graph_us(step_sag)
print("cfg4 IMDB-B SAGPool(0.5) h128 b128, sync-free fused levels: %.0f us/step eager, %s us/step hipGraph -> %.0f graphs/s" % (t, "%.0f" % tg if tg else "n/a", 128 / (tg or t) * 1e6))

# the same step as a data-parallel optimiser step (gradient bucket + clip + Adam; the RCCL all-reduce of the bucket sits between
# the two hipGraphs at N > 1): FlatTrainer / GraphedStep are model-agnostic
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep
tr4 = FlatTrainer(net, lr=5e-4, clip=2.0, defer_loss=True)
gs4 = GraphedStep(tr4, lambda: mp.nll_loss(net(d), lab4), warmup=3)
for _ in range(5):
    gs4.step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(gs4.stream)
for _ in range(50):
    gs4.step()
e1.record(gs4.stream); e1.synchronize()
t = e0.elapsed_time(e1) / 50 * 1e3
print("cfg4 full optimiser step (fwd + bwd + bucket + clip + Adam) from one hipGraph: %.0f us/step -> %.0f graphs/s per GPU, loss %.4f"
      % (t, 128 / t * 1e6, gs4.loss_value()))
torch.cuda.set_stream(_S)

def optimiser_step_us(model, loss_fn, B, tag, iters=50):
    """fwd + bwd + bucket + clip + Adam from one hipGraph (FlatTrainer / GraphedStep are model-agnostic)"""
    tr = FlatTrainer(model, lr=5e-4, clip=2.0, defer_loss=True)
    gs = GraphedStep(tr, loss_fn, warmup=3)
    for _ in range(5):
        gs.step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(gs.stream)
    for _ in range(iters):
        gs.step()
    e1.record(gs.stream); e1.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e3
    print("%s full optimiser step (fwd + bwd + bucket + clip + Adam) from one hipGraph: %.0f us/step -> %.0f graphs/s per GPU" % (tag, t, B / t * 1e6))
    torch.cuda.set_stream(_S)


optimiser_step_us(gat32, lambda: gat32.loss(gat32(x32, g32)[1], lab32), 32, "cfg3 b32")

# config 5: DD DiffPool 64 -> 8, h=64, batch 16, Nmax 512
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
dpm = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                              assign_input_dim=89, final_dim="number_classes").to(dev)
def step_dp():
    dpm.zero_grad(set_to_none=True)
    with mp.deferred_loss():          # the cross-entropy inside the head's backward, as FlatTrainer(defer_loss=True) runs it
        dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5).backward(gradient=mp.unit_seed(dev))
t = timeit(step_dp, iters=10, warm=3)
tg = graph_us(step_dp)
print("cfg5 DD DiffPool 512->64->8 h64 b16: %.0f us/step eager, %s us/step hipGraph -> %.0f graphs/s" % (t, "%.0f" % tg if tg else "n/a", 16 / (tg or t) * 1e6))
optimiser_step_us(dpm, lambda: dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5), 16, "cfg5")
# the contraction alone, level 1 (S [rows,64], Z [rows,192]) + level 2 dense
Sm = torch.softmax(torch.randn(g5.total_rows, 64, device=dev), -1); Sm[g5.n_rows:] = 0
Z = torch.randn(g5.total_rows, 192, device=dev)
gr = torch.cuda.CUDAGraph(); st = torch.cuda.current_stream()
with torch.cuda.stream(st):
    for _ in range(3):
        dp.diffpool_contract_rows(Sm, Z, g5)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(20):
            xo, ao = dp.diffpool_contract_rows(Sm, Z, g5)
    gr.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st); gr.replay(); e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
# f4: link-prediction side loss (value + dS) on the level-1 assignment of the same batch
Sl = Sm.clone().requires_grad_(True)
def step_lp():
    Sl.grad = None; dp.link_pred_loss(Sl, g5).backward()
tl = graph_us(step_lp) or timeit(step_lp)
pairs = float((hb5["sizes"].astype(np.float64) ** 2).sum())
print("f4 link-pred loss + gradient, DD b16 K=64 (%.2f M row pairs, no [B,N,N] tensor): %.1f us -> %.1f GFLOP/s of pair products"
      % (pairs / 1e6, tl, 4 * pairs * 64 / tl / 1e3))
nb = hb5["sizes"].astype(np.float64)
dense_flops = 16 * (2 * 512 * 512 * 64 + 2 * 64 * 512 * 64 + 2 * 64 * 512 * 192)          # reference's padded bmm's, level 1
sparse_flops = 2 * g5.nnz * 64 + float((2 * 64 * nb * 64 + 2 * 64 * nb * 192).sum())    # what is executed: SpMM + ragged GEMMs
print("cfg5 level-1 contraction X'=S^T Z, A'=S^T A S (2 launches: SpMM + both ragged products): %.1f us ; executed %.1f MFLOP -> %.1f TF (%.1f%% of 157.3 TF fp32 MFMA); "
      "reference-equivalent dense work %.1f MFLOP -> %.1f TF-equivalent (%.0f%%)"
      % (us, sparse_flops / 1e6, sparse_flops / us / 1e6, sparse_flops / us / 1e6 / 157.3 * 100, dense_flops / 1e6,
         dense_flops / us / 1e6, dense_flops / us / 1e6 / 157.3 * 100))
