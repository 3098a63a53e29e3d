#!/usr/bin/env python3
"""BASELINE config 5 (DD DiffPool 512 -> 64 -> 8, h = 64, batch 16): fwd + bwd step, eager and replayed from a hipGraph, with
the launch count of one step (TSGNN_DENSE_ONE_LAUNCH=0 selects the layer-by-layer pooled levels for comparison)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic, message_passing as mp, _native as nat
dev = torch.device("cuda"); torch.manual_seed(0)
class A: bias = True
S = torch.cuda.Stream(); torch.cuda.set_stream(S)
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
dpm = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                              assign_input_dim=89, final_dim="number_classes").to(dev)
def step():
    dpm.zero_grad(set_to_none=True)
    with mp.deferred_loss():          # the cross-entropy inside the head's backward, as FlatTrainer(defer_loss=True) runs it
        dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5).backward(gradient=mp.unit_seed(dev))
for _ in range(3): step()
torch.cuda.synchronize(); mp.check_device_errors()
nat.trace = []; step(); n_lib = len(nat.trace); names = [t[2] or t[0] for t in nat.trace]; nat.trace = None
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=S):
    step()
gr.replay(); torch.cuda.synchronize(); mp.check_device_errors()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(S)
for _ in range(50): gr.replay()
e1.record(S); e1.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
mp.check_device_errors()
print("cfg5 DD DiffPool 512->64->8 h64 b16: %.0f us/step from one hipGraph -> %.0f graphs/s ; %d library launches per step (+ torch element-wise)"
      % (us, 16 / us * 1e6, n_lib))
from collections import Counter
print("   " + ", ".join("%s x%d" % kv for kv in Counter(n.split("<")[0] for n in names).most_common(14)))
