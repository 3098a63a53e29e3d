"""does torch.cuda.CUDAGraph.debug_dump (hipGraphDebugDotPrint) list kernel names on this stack?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import _native as nat
x = torch.randn(1000, 128, device="cuda")
p = torch.zeros(1000, device="cuda"); m = torch.zeros_like(p); v = torch.zeros_like(p); g = torch.randn(1000, device="cuda")
state = torch.zeros(4, device="cuda"); ws = torch.zeros(264, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gr = torch.cuda.CUDAGraph()
    gr.enable_debug_mode()
    with torch.cuda.graph(gr, stream=s):
        nat.call("clip_adam_step_f32", p, g, m, v, 1000, 1e-2, 0.9, 0.999, 1e-8, 0.0, 0.7, 0.5, state, ws, None)
        y = x * 2
    os.makedirs("gpurun_out", exist_ok=True)
    gr.debug_dump("gpurun_out/graph_dump.dot")
print(open("gpurun_out/graph_dump.dot").read()[:3000])
