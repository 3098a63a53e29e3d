#!/usr/bin/env python3
"""the surface-(B) optimiser step alone (pyg.SageNet on a DD / PROTEINS batch, one hipGraph), for rocprofv3 / replay_trace.py:
   python scripts/pyg_step.py [DD|PROTEINS|MUTAG] [seed] [replays]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_stage_gnn_amd import message_passing as mp, pyg, synthetic
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep


class D:
    pass


shape = sys.argv[1] if len(sys.argv) > 1 else "DD"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
B, nmax, L, hid = {"DD": (32, 1000, 3, 128), "PROTEINS": (64, 620, 3, 128), "MUTAG": (32, 40, 2, 64)}[shape]
dev = torch.device("cuda")
hb = synthetic.host_batch(seed=seed, B=B, shape=shape, nmax=nmax)
d = D()
d.x, d.edge_index, d.batch, lab = synthetic.to_pyg(hb, dev)
torch.manual_seed(1234)
net = pyg.SageNet(synthetic.SHAPES[shape][2], hid, 2, num_layers=L).to(dev).train()
tr = FlatTrainer(net, lr=1e-3, clip=2.0, defer_loss=True)
if os.environ.get("PYG_EAGER") == "1":                    # eager steps (PMC counters are attributed per dispatch outside a hipGraph)
    for _ in range(reps):
        tr.step(lambda: mp.nll_loss(net(d), lab))
    torch.cuda.synchronize()
    sys.exit(0)
gs = GraphedStep(tr, lambda: mp.nll_loss(net(d), lab), warmup=3)
for _ in range(20):
    gs.step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(gs.stream)
for _ in range(reps):
    gs.step()
e1.record(gs.stream); e1.synchronize()
print("%s b%d seed %d: %.1f us/step, loss %.5f, %s" % (shape, B, seed, e0.elapsed_time(e1) / reps * 1e3, gs.loss_value(), gs.describe()))
