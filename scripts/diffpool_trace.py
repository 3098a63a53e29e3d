#!/usr/bin/env python3
"""DiffPool cfg5: the ordered library launches of one fwd + bwd step (entry point, device kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic, message_passing as mp, _native as nat
dev = torch.device("cuda"); torch.manual_seed(0)
class A: bias = True
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
dpm = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                              assign_input_dim=89, final_dim="number_classes").to(dev)
def step():
    dpm.zero_grad(set_to_none=True); dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5).backward(gradient=mp.unit_seed(dev))
for _ in range(3): step()
torch.cuda.synchronize()
nat.trace = []; step(); tr = nat.trace; nat.trace = None
for i, t in enumerate(tr):
    print(i, t[0], "|", t[2])
