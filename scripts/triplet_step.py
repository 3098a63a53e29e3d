#!/usr/bin/env python3
"""the 2stg triplet step (tripletnet.py:16-45; ONE triplet per optimiser step) on a DD-shaped triplet resident in HBM —
forward + margin loss + backward + clip + Adam replayed from one hipGraph, with the launch inventory of one step.
TRIPLET_CRITERION=torch uses torch.nn.MarginRankingLoss (as the reference's loop builds it) instead of the drop-in one."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic, _native as nat
from two_stage_gnn_amd.triplet import tripletnet, MarginRankingLoss
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep
from torch.profiler import profile, ProfilerActivity
from collections import Counter
dev = torch.device("cuda"); torch.manual_seed(5)
hb = synthetic.host_batch(11, 3, "DD", 1000)
class A: bias = True
m = E.GcnEncoderGraph(hb["fin"], 128, 128, 2, 3, bn=True, args=A(), final_dim="output_dim").to(dev)
net = tripletnet(m)
g3, x3, _ = synthetic.to_device(hb, dev)
crit = torch.nn.MarginRankingLoss(margin=1.0) if os.environ.get("TRIPLET_CRITERION") == "torch" else MarginRankingLoss(margin=1.0)
tgt = torch.full((1,), -1.0, device=dev)
def loss_fn():
    dp, dn = net._embed(x3, g3, hb["sizes"], x3)[:2]
    return crit(dp, dn, tgt)
tr = FlatTrainer(m, lr=1e-3, clip=2.0)
gs = GraphedStep(tr, loss_fn, warmup=3)
for _ in range(100): gs.step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(gs.stream)
for _ in range(200): gs.step()
e1.record(gs.stream); e1.synchronize()
us = e0.elapsed_time(e1) / 200 * 1e3
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    gs.step(); torch.cuda.synchronize()
kern = [e for e in prof.events() if e.device_type.name == "CUDA"]
print("f3 DD-shaped triplet (%s nodes, Nmax 1000), GraphSage 3L h128, %s: %.1f us/step from one hipGraph -> %.0f triplets/s ; %d device kernels per step"
      % ("/".join(str(int(v)) for v in hb["sizes"]), type(crit).__module__.split(".")[0] + ".MarginRankingLoss", us, 1e6 / us, len(kern)))
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    return n.split("(")[0].split("<")[0][:40] or n[:40]
print("   " + ", ".join("%s x%d" % kv for kv in Counter(short(e.name) for e in kern).most_common(30)))
print("   last kernel of the step: " + short(sorted(kern, key=lambda e: e.time_range.start)[-1].name))
