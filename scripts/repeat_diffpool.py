#!/usr/bin/env python3
"""is the DiffPool step (BASELINE config 5) bitwise repeatable?  N eager steps on the same batch and parameters: distinct (loss,
gradient) results, and where two of them differ."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic, message_passing as mp
dev = torch.device("cuda"); torch.manual_seed(0)
class A: bias = True
S = torch.cuda.Stream(); torch.cuda.set_stream(S)         # (everything on the stream the capture below uses)
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
dpm = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                              assign_input_dim=89, final_dim="number_classes").to(dev)
names = [k for k, p in dpm.named_parameters()]
res = []
for it in range(int(os.environ.get("N", "60"))):
    dpm.zero_grad(set_to_none=True)
    loss = dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5)
    loss.backward(gradient=mp.unit_seed(dev))
    res.append((loss.detach().clone(), {k: p.grad.clone() for k, p in dpm.named_parameters() if p.grad is not None}))
torch.cuda.synchronize(); mp.check_device_errors()
keys = []
for l, g in res:
    keys.append((float(l),) + tuple(float(g[k].double().sum()) for k in sorted(g)))
distinct = sorted(set(keys))
print("steps", len(res), "distinct results", len(distinct), "losses", sorted(set(k[0] for k in keys))[:5])
l0, g0 = res[0]
for i in (1, 2, len(res) - 1):
    li, gi = res[i]
    diff = [(k, float((gi[k] - g0[k]).abs().max()), float(g0[k].abs().max())) for k in sorted(g0) if not torch.equal(gi[k], g0[k])]
    print("step", i, "vs 0: loss equal", bool(torch.equal(li, l0)), "; differing tensors", len(diff), diff[:6])

# ---- the same under hipGraph replay (kernels back to back: the barriers of the pooled-level stacks at full speed)
del res
def step():
    dpm.zero_grad(set_to_none=True)
    l = dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5)
    l.backward(gradient=mp.unit_seed(dev))
    return l
for _ in range(3): step()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=S):
    lg = step()
grads = [p.grad for p in dpm.parameters() if p.grad is not None]
NR = int(os.environ.get("REPLAYS", "1000"))
seen = set()
for it in range(NR):
    gr.replay()
    if it % 10 == 9 or it < 5:                       # (read back every tenth replay: the rest run back to back)
        torch.cuda.synchronize()
        seen.add((float(lg),) + tuple(float(t.double().sum()) for t in grads))
torch.cuda.synchronize(); mp.check_device_errors()
print("hipGraph replays", NR, ": distinct (loss, gradient sums) among the sampled replays:", len(seen), "; equal to the eager result:",
      (float(lg),) + tuple(float(g0[k].double().sum()) for k in [n for n, p in dpm.named_parameters() if p.grad is not None]) in seen)
