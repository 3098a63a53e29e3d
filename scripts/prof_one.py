#!/usr/bin/env python3
"""Launch one kernel family a few dozen times (eager) so rocprofv3 --pmc / --kernel-trace can be read per kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import message_passing as mp, synthetic, _native as nat

which = sys.argv[1] if len(sys.argv) > 1 else "rowgemm"
B = int(os.environ.get("B", 32)); H = 128
hb = synthetic.host_batch(0, B, "DD", 1000)
g, x, label = synthetic.to_device(hb, torch.device("cuda"))
R = g.total_rows
X = torch.randn(R, H, device="cuda"); Y = torch.empty_like(X)
W = torch.randn(H, H, device="cuda") * 0.1; b = torch.randn(H, device="cuda"); rinv = torch.empty(R, device="cuda")
for _ in range(int(os.environ.get("N", 30))):
    if which == "rowgemm":
        nat.call("rowgemm_f32", X, H, W, H, 0, b, Y, H, rinv, R, H, H, 1, 0)
    elif which == "spmm":
        mp.spmm_raw(g.rowptr, g.col, None, X, R, out=Y)
    elif which == "ell":
        mp.spmm_ell(g, X, out=Y)
    elif which == "gather":                 # aggregation fused into the product, as the step's forward layers launch it
        ell, ell_w, tail = g.ell()
        assert tail is None
        Z = getattr(sys.modules[__name__], "_Z", None)
        if Z is None:
            Z = sys.modules[__name__]._Z = torch.empty_like(X)
        nat.call("gather_rowgemm_f32", ell, ell_w, None, None, X, H, W, H, 0, b, Y, H, rinv, Z, H, g.n_rows, H, H, 1, g.n_ghost)
    elif which == "prop":                   # GCN-normalised aggregation of the SAGPool path (tsgnn_gcn_propagate_f32)
        from two_stage_gnn_amd import sag_stack as SS
        dinv, self_w = SS.gcn_coef(g)
        nat.call("gcn_propagate_f32", g.rowptr, g.col, dinv, self_w, X, H, 0, None, None, None, Y, H, None, g.n_rows, H)
torch.cuda.synchronize()
