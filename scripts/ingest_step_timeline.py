#!/usr/bin/env python3
"""The capacity-padded ingest step against the exact resident batch of the drawn batches' mean size, for scripts/replay_trace.py:
MODE=cap replays slot 0 of an IngestPipeline last, MODE=exact the exact batch last (the trace's last replay is the one analysed)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, ingest, synthetic
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep
dev = torch.device("cuda"); torch.manual_seed(1234)
class A: bias = True
B, nmax = 32, 1000
model = E.GcnEncoderGraph(synthetic.SHAPES["DD"][2], 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
trainer = FlatTrainer(model, lr=1e-3, clip=2.0, defer_loss=True)
ds = ingest.synthetic_dataset(seed=4242, n_graphs=512, shape="DD", nmax=nmax)
rng = np.random.default_rng(77)
sched = [rng.choice(len(ds), size=B, replace=False) for _ in range(60)]
pipe = ingest.IngestPipeline(model, trainer, ds, B, nmax, dev, sched)
rows = [int(ds.sizes[ids].sum()) for ids in sched]
k_mean = int(np.argmin(np.abs(np.asarray(rows) - np.mean(rows))))
g_m, x_m, y_m = ds.collate(sched[k_mean], nmax, ds.features("node-label"), dev)
gs_m = GraphedStep(trainer, lambda: model.loss(model(x_m, g_m)[1], y_m), warmup=2)
mode = os.environ.get("MODE", "cap")
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(pipe.compute if fn is cap else gs_m.stream)
    for _ in range(n): fn()
    (e1.record(pipe.compute if fn is cap else gs_m.stream)); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cap = pipe.steps[0].step
ex = gs_m.step
if mode == "cap":
    print("exact %d rows: %.1f us" % (rows[k_mean], t(ex))); print("capacity %d rows (ghost slots %d): %.1f us" % (pipe.row_cap, pipe.slots[0].g.ghost_slots_fixed, t(cap)))
else:
    print("capacity %d rows: %.1f us" % (pipe.row_cap, t(cap))); print("exact %d rows: %.1f us" % (rows[k_mean], t(ex)))
torch.cuda.synchronize()
