import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda"); torch.manual_seed(0)
class A:
    bias = True
hb5 = synthetic.host_batch(4, 16, "DD", 512)
g5, x5, lab5 = synthetic.to_device(hb5, dev)
dpm = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                              assign_input_dim=89, final_dim="number_classes").to(dev)
def step():
    dpm.zero_grad(set_to_none=True); dpm.loss(dpm(x5, g5, hb5["sizes"], assign_x=x5)[1], lab5).backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.device_time_total > 0 and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.count)
for e in rows:
    print("%-28s x%-3d %6.1f us  | %s" % (e.key, e.count, e.self_device_time_total, " <- ".join(s.split("/")[-1] for s in e.stack[:6])))
