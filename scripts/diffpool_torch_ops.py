"""which torch (non-library) kernels does one DiffPool step launch, and from which line of the package?  (wraps the torch entry
points that launch copy / fill kernels and records the nearest caller inside two-stage-gnn_amd)"""
import os, sys, collections, traceback
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
cnt = collections.Counter()
def where():
    for f in reversed(traceback.extract_stack(limit=14)[:-2]):
        if "two-stage-gnn_amd" in f.filename or "two_stage_gnn_amd" in f.filename:
            return "%s:%d" % (os.path.basename(f.filename), f.lineno)
    return "(autograd engine / torch)"
def wrap(obj, name, cond=lambda *a, **k: True):
    orig = getattr(obj, name)
    def w(*a, **k):
        if cond(*a, **k):
            cnt[(name, where())] += 1
        return orig(*a, **k)
    setattr(obj, name, w)
wrap(torch, "zeros"); wrap(torch, "cat"); wrap(torch, "zeros_like"); wrap(torch, "ones")
wrap(torch.Tensor, "contiguous", lambda t, *a, **k: t.is_cuda and not t.is_contiguous())
wrap(torch.Tensor, "zero_", lambda t, *a, **k: t.is_cuda); wrap(torch.Tensor, "copy_", lambda t, *a, **k: t.is_cuda)
wrap(torch.Tensor, "clone", lambda t, *a, **k: t.is_cuda); wrap(torch.Tensor, "add_", lambda t, *a, **k: t.is_cuda)
wrap(torch.Tensor, "fill_", lambda t, *a, **k: t.is_cuda); wrap(torch.Tensor, "float", lambda t, *a, **k: t.is_cuda and t.dtype != torch.float32)
wrap(torch.Tensor, "reshape", lambda t, *a, **k: t.is_cuda and not t.is_contiguous())
step()
torch.cuda.synchronize()
for k, v in cnt.most_common(50):
    print(v, k)
print("---- autograd / aten ops of one step (torch profiler, CPU side)")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step()
torch.cuda.synchronize()
c2 = collections.Counter(ev.name for ev in prof.events())
for k, v in c2.most_common(60):
    if any(t in k for t in ("Backward", "aten::copy_", "aten::fill_", "aten::zero", "aten::add", "aten::cat", "aten::clone", "AccumulateGrad", "aten::contiguous", "aten::mul", "aten::sum", "aten::select", "aten::slice", "aten::index", "aten::empty_like", "aten::narrow")):
        print(v, k)
