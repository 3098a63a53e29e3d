"""which torch (non-library) kernels does one DiffPool step launch, and from where?"""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic
dev = torch.device("cuda"); torch.manual_seed(0)
class A: bias = True
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
dpm = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                              assign_input_dim=89, final_dim="number_classes").to(dev)
def step():
    dpm.zero_grad(set_to_none=True); dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5).backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::cat", "aten::mul", "aten::sum", "aten::clone", "aten::zeros", "aten::index", "aten::_to_copy"):
        st = [f for f in ev.stack if "two-stage-gnn_amd" in f or "two_stage_gnn_amd" in f]
        cnt[(ev.name, st[0].split("two-stage-gnn_amd/")[-1] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for k, v in cnt.most_common(40):
    print(v, k)
