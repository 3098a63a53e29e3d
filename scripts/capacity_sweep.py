"""step time (pull + expand + fwd + bwd + optimiser, one hipGraph) against the slot's row capacity"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, ingest
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep
dev = torch.device("cuda")
class A: bias = True
torch.manual_seed(0)
model = E.GcnEncoderGraph(89, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
tr = FlatTrainer(model, lr=1e-3, clip=2.0, defer_loss=True)
ds = ingest.synthetic_dataset(4242, 512, "DD", 1000)
order = np.argsort(ds.sizes)
ids = order[100:132]                       # 32 smallish graphs: fits every capacity below
n = int(ds.sizes[ids].sum())
print("batch rows", n)
cs = torch.cuda.Stream()
for cap in (8192, 8448, 8704, 9216, 9728, 10240):
    for ghost in (int(ds.sizes.max()) + 1,):
        s = ingest.CapacityBatch(32, 1000, cap, 60000, 89, dev, ghost_slots=ghost)
        s.collate(ds, ids)
        def loss(s=s):
            s.pull()
            return model.loss(model(s.x, s.g)[1], s.label)
        gs = GraphedStep(tr, loss, warmup=2, stream=cs)
        for _ in range(20): gs.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): gs.step()
        torch.cuda.synchronize()
        print("row_cap %5d (%3d panels) ghost slots %d: %.1f us/step" % (cap, cap // 32, ghost, (time.perf_counter() - t0) / 300 * 1e6))
