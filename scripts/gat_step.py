#!/usr/bin/env python3
"""BASELINE config 3, batched (DD GAT 2 layers x 4 heads x 64, 32 graphs in one block-diagonal step, per-graph features, packed
rows + one ghost representative per graph): fwd + bwd replayed from a hipGraph, with the launch inventory of one step
(TSGNN_GAT_FUSED=0 selects the per-op path of attention.py for comparison).  Under rocprofv3: eager steps only (GAT_EAGER=n)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import gat_encoders as G, synthetic, _native as nat, message_passing as mp
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda"); torch.manual_seed(0)
S = torch.cuda.Stream(); torch.cuda.set_stream(S)
hb32 = synthetic.host_batch(2, 32, "DD", 1000)
x32, adj32 = synthetic.to_dense(hb32)
gat32 = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes", per_graph_features=True).to(dev)
adj32d = adj32.to(dev)
x32, g32 = gat32.packed_batch(x32.to(dev), adj32d, hb32["sizes"])
lab32 = torch.from_numpy(hb32["label"]).to(dev)
def step():
    gat32.zero_grad(set_to_none=True)
    with mp.deferred_loss():          # the cross-entropy inside the head's backward, as FlatTrainer(defer_loss=True) runs it
        gat32.loss(gat32(x32, g32)[1], lab32).backward(gradient=mp.unit_seed(dev))
for _ in range(3): step()
torch.cuda.synchronize()
eager = int(os.environ.get("GAT_EAGER", "0"))
if eager:
    for _ in range(eager): step()
    torch.cuda.synchronize(); sys.exit(0)
nat.trace = []; step(); names = [t[2] or t[0] for t in nat.trace]; nat.trace = None
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
kern = [e for e in prof.events() if e.device_type.name == "CUDA" and "emcpy" not in e.name and "emset" not in e.name]
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=S):
    step()
gr.replay(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(S)
for _ in range(50): gr.replay()
e1.record(S); e1.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print("cfg3 DD GAT-2L 4 heads h64 b32 (%d rows, %d entries): %.0f us/step from one hipGraph -> %.0f graphs/s ; %d library launches, %d device kernels per step"
      % (g32.n_rows, g32.nnz, us, 32 / us * 1e6, len(names), len(kern)))
from collections import Counter
print("   library: " + ", ".join("%s x%d" % kv for kv in Counter(n.split("<")[0] for n in names).most_common(20)))
print("   other device kernels: " + ", ".join("%s x%d" % (k[:60], v) for k, v in Counter(e.name for e in kern if "anonymous namespace" not in e.name or "at::" in e.name).most_common(12)))
