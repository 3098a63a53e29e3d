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
    def tick(name, t0):
        t1 = time.perf_counter(); T[name] = T.get(name, 0.0) + (t1 - t0); return t1
    depth = len(pipe.slots)
    pool = pipe._get_pool(workers) if workers else None
    S = sched[20:]
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    if pool is not None:
        for k in range(depth):
            pipe.slots[k].collate_async(pool, ds, S[k])
    for k, ids in enumerate(S):
        s, gs = pipe.slots[k % depth], pipe.steps[k % depth]
        t = time.perf_counter()
        if pool is not None:
            s.collate_wait()
        else:
            s.collate(ds, ids)
        t = tick("collate/wait", t)
        gs.step(); t = tick("replay", t)
        s.mark_consumed(pipe.compute); t = tick("record", t)
        if pool is not None and k + depth < len(S):
            s.collate_async(pool, ds, S[k + depth])
        t = tick("submit", t)
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
