"""where does the host time of one ingest iteration go? (perf_counter around each phase, 300 iterations)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, ingest, synthetic
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
        pipe.copy.wait_event(s.consumed); t = tick("wait consumed", t)
        s.upload(pipe.copy); t = tick("upload", t)
        pipe.compute.wait_event(s.uploaded); t = tick("wait uploaded", t)
        gs.step(); t = tick("replay", t)
        s.consumed.record(pipe.compute); t = tick("record", t)
        if pool is not None and k + depth < len(S):
            s.collate_async(pool, ds, S[k + depth])
        t = tick("submit", t)
    t_host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print("workers=%d: host loop %.1f us/iter, with GPU drain %.1f us/iter" % (workers, t_host / len(S) * 1e6, t_tot / len(S) * 1e6))
    print("   " + "  ".join("%s %.1f" % (k, v / len(S) * 1e6) for k, v in T.items()))

# ---- which part of the per-step protocol costs GPU time?  (400 iterations each, wall clock incl. drain)
def timed(fn, n=400):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
depth = len(pipe.slots)
print("replay slot 0 only            %.1f us" % timed(lambda k: pipe.steps[0].step()))
print("replay alternating slots      %.1f us" % timed(lambda k: pipe.steps[k % depth].step()))
def with_events(k):
    s = pipe.slots[k % depth]
    pipe.copy.wait_event(s.consumed); s.uploaded.record(pipe.copy); pipe.compute.wait_event(s.uploaded)
    pipe.steps[k % depth].step(); s.consumed.record(pipe.compute)
print("+ event protocol, no upload   %.1f us" % timed(with_events))
def with_upload(k):
    s = pipe.slots[k % depth]
    pipe.copy.wait_event(s.consumed); s.upload(pipe.copy); pipe.compute.wait_event(s.uploaded)
    pipe.steps[k % depth].step(); s.consumed.record(pipe.compute)
print("+ upload of the same staging  %.1f us" % timed(with_upload))
def upload_only(k):
    s = pipe.slots[k % depth]
    s.upload(pipe.copy)
print("upload only (copy stream)     %.1f us" % timed(upload_only))
def same_stream(k):
    s = pipe.slots[k % depth]
    s.upload(pipe.compute)
    pipe.steps[k % depth].step()
print("upload on the compute stream, no events  %.1f us" % timed(same_stream))
import ctypes
def copy_only(k):
    s = pipe.slots[k % depth]
    with torch.cuda.stream(pipe.copy):
        s.dev.copy_(s.host, non_blocking=True)
print("H2D copy only (copy stream)   %.1f us  (%d bytes)" % (timed(copy_only), 4 * pipe.slots[0].words))
small = torch.zeros(65536, dtype=torch.int32).pin_memory(); smalld = torch.zeros(65536, dtype=torch.int32, device=dev)
def copy_small(k):
    with torch.cuda.stream(pipe.copy):
        smalld.copy_(small, non_blocking=True)
print("H2D copy of 256 KB            %.1f us" % timed(copy_small))
