#!/usr/bin/env python3
"""which torch operators (not library launches) does one DiffPool step (BASELINE config 5) put on the device, and from where?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic, message_passing as mp
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
seen = set()
def chain(e):
    out = []
    while e is not None and len(out) < 4:
        out.append(e.name); e = e.cpu_parent
    return " <- ".join(out)
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::") or not e.kernels:
        continue
    if any(c.kernels for c in e.cpu_children):
        continue                                     # report the innermost operator only
    print("%-60s %-44s %s" % (chain(e)[:60], str(e.input_shapes)[:44], ",".join(k.name[:40] for k in e.kernels)))
