"""where does the time of one ingest iteration go? host phases (perf_counter) and GPU-side variants, 300 iterations each"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, ingest
from two_stage_gnn_amd.data_parallel import FlatTrainer
dev = torch.device("cuda")
class A: bias = True
torch.manual_seed(0)
model = E.GcnEncoderGraph(89, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
tr = FlatTrainer(model, lr=1e-3, clip=2.0, defer_loss=True)
ds = ingest.synthetic_dataset(4242, 512, "DD", 1000)
rng = np.random.default_rng(77)
sched = [rng.choice(len(ds), size=32, replace=False) for _ in range(320)]
pipe = ingest.IngestPipeline(model, tr, ds, 32, 1000, dev, sched)
pipe.run(sched[:20]); torch.cuda.synchronize()
for workers in (3, 2, 1, 0):
    T = {}
    S = sched[20:]
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    pipe.run(S, workers=workers, ticks=T)
    t_host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print("workers=%d: host loop %.1f us/iter, with GPU drain %.1f us/iter" % (workers, t_host / len(S) * 1e6, t_tot / len(S) * 1e6))
    print("   " + "  ".join("%s %.1f" % (k, v / len(S) * 1e6) for k, v in T.items()))

def timed(fn, n=400):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
depth = len(pipe.slots)
print("replay slot 0 only (pull + expand + step, same staged batch)  %.1f us" % timed(lambda k: pipe.steps[0].step()))
print("replay alternating slots                                      %.1f us" % timed(lambda k: pipe.steps[k % depth].step()))
with torch.cuda.stream(pipe.compute):
    print("pull + expand alone                                           %.1f us" % timed(lambda k: pipe.slots[0].pull()))
    print("expand alone                                                  %.1f us" % timed(lambda k: pipe.slots[0].expand()))
