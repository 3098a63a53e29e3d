#!/usr/bin/env python3
"""Un-profiled breakdown of the bench step: replay time of growing prefixes of the step (hipGraph, HIP events)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, synthetic, message_passing as mp
from two_stage_gnn_amd.data_parallel import FlatTrainer


def replay_us(fn, iters=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(iters):
                g.replay()
            e1.record(s); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


class A:
    bias = True

dev = torch.device("cuda")
torch.manual_seed(0)
model = E.GcnEncoderGraph(89, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
hb = synthetic.host_batch(0, 32, "DD", 1000)
g, x, label = synthetic.to_device(hb, dev)
tr = FlatTrainer(model, lr=1e-3, clip=2.0)

def stack():
    return model.readouts_rows(x, g)
def fwd():
    _, yp = model(x, g); return model.loss(yp, label)
def fwd_bwd():
    tr.zero_grad(); fwd().backward()
def fwd_bwd_gather():
    fwd_bwd(); tr.gather_grads()
def full():
    fwd_bwd_gather(); tr.apply()
def stack_fb():
    tr.zero_grad(); stack().sum().backward()

with torch.no_grad():
    t_stack_nograd = replay_us(stack)
print("stack forward (no grad)        %7.1f us" % t_stack_nograd)
print("stack forward (grad mode)      %7.1f us" % replay_us(stack))
print("stack fwd + stack bwd          %7.1f us" % replay_us(stack_fb))
print("full forward + loss            %7.1f us" % replay_us(fwd))
print("forward + backward             %7.1f us" % replay_us(fwd_bwd))
print("  + gradient bucket            %7.1f us" % replay_us(fwd_bwd_gather))
print("  + clip + Adam (whole step)   %7.1f us" % replay_us(full))
