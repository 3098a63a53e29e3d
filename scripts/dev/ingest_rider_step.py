#!/usr/bin/env python3
"""the ingest slot's captured step alone and the pipeline with a new batch every step, for the rider-placement sweep
(scripts/dev/ingest_rider_sweep.sh: TSGNN_INGEST_PULL_PARTS / TSGNN_INGEST_PULL_SKIP)"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from two_stage_gnn_amd import dense_encoders as E, ingest
from two_stage_gnn_amd.data_parallel import FlatTrainer
dev = torch.device("cuda")
class Args: bias = True
torch.manual_seed(1234)
model = E.GcnEncoderGraph(89, 128, 128, 2, 3, bn=True, args=Args(), final_dim="number_classes").to(dev)
tr = FlatTrainer(model, lr=1e-3, clip=2.0, defer_loss=True)
ds = ingest.synthetic_dataset(seed=4242, n_graphs=512, shape="DD", nmax=1000)
rng = np.random.default_rng(77)
sched = [rng.choice(len(ds), size=32, replace=False) for _ in range(220)]
pipe = ingest.IngestPipeline(model, tr, ds, 32, 1000, dev, sched)
gs = pipe.steps[0]
for _ in range(30): gs.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): gs.step()
torch.cuda.synchronize()
print("slot step alone (capacity %d rows = %d panels): %.4f ms" % (pipe.row_cap, pipe.row_cap // 32, (time.perf_counter() - t0) / 200 * 1e3))
pipe.run(sched[:20]); torch.cuda.synchronize()
T = {}
t0 = time.perf_counter(); pipe.run(sched[20:], ticks=T); torch.cuda.synchronize()
print("a new batch every step: %.4f ms/step; host us/step %s" % ((time.perf_counter() - t0) / 200 * 1e3, {k: round(v / 200 * 1e6, 1) for k, v in T.items()}))
