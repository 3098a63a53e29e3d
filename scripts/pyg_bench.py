#!/usr/bin/env python3
"""Surface (B) beside surface (A): the full optimiser step (forward + nll / cross-entropy + backward + gradient bucket + clip + Adam, ONE
hipGraph, FlatTrainer + GraphedStep) of the PyG-named models (pyg.SageNet on fused SAGEConv launches, ...) next to the reference-surface
siblings (dense_encoders.GcnEncoderGraph, ...) on the same synthetic batches.  BASELINE.json configs 1-2 as worded + the headline shape."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_stage_gnn_amd import dense_encoders as E, message_passing as mp, pyg, synthetic
from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep


class A:
    bias = True


class D:
    pass


def step_us(model, loss_fn, iters=200, settle=64):
    tr = FlatTrainer(model, lr=1e-3, clip=2.0, defer_loss=True)
    gs = GraphedStep(tr, loss_fn, warmup=3)
    for _ in range(settle):
        gs.step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(gs.stream)
    for _ in range(iters):
        gs.step()
    e1.record(gs.stream); e1.synchronize()
    loss = gs.loss_value()
    return e0.elapsed_time(e1) / iters * 1e3, loss, gs.describe()


def main():
    dev = torch.device("cuda")
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = []
    for shape, B, nmax, L, hid, seed in (("DD", 32, 1000, 3, 128, 0), ("DD", 32, 1000, 3, 128, 6), ("PROTEINS", 64, 620, 3, 128, 1), ("MUTAG", 32, 40, 2, 64, 0)):
        if only and only != shape:
            continue
        hb = synthetic.host_batch(seed=seed, B=B, shape=shape, nmax=nmax)
        fin = synthetic.SHAPES[shape][2]
        # surface (A): the reference's own classes
        g, x, label = synthetic.to_device(hb, dev)
        torch.manual_seed(1234)
        ma = E.GcnEncoderGraph(fin, hid, hid, 2, L, bn=True, args=A(), final_dim="number_classes").to(dev)
        ta, la, da = step_us(ma, lambda: ma.loss(ma(x, g)[1], label))
        # surface (B): torch_geometric-named layers
        d = D()
        d.x, d.edge_index, d.batch, lab = synthetic.to_pyg(hb, dev)
        torch.manual_seed(1234)
        mb = pyg.SageNet(fin, hid, 2, num_layers=L).to(dev).train()
        tb, lb, db = step_us(mb, lambda: mb.loss(d, lab) if hasattr(mb, "loss") else mp.nll_loss(mb(d), lab))
        rows.append((shape, B, L, hid, seed, int(hb["sizes"].sum()), ta, tb))
        print("%-8s b%-3d %dL h%-3d seed %d (%5d rows): surface A GcnEncoderGraph %.1f us/step (%.0f graphs/s) | surface B SageNet[SAGEConv] %.1f us/step (%.0f graphs/s) | B / A = %.2f"
              % (shape, B, L, hid, seed, int(hb["sizes"].sum()), ta, B / ta * 1e6, tb, B / tb * 1e6, tb / ta), flush=True)


def gat():
    """config 3 as worded: DD GATConv 2 layers x 4 heads x 64, batch 32 — beside the reference-surface DGATEncoderGraph step"""
    from two_stage_gnn_amd import gat_encoders as G
    dev = torch.device("cuda")
    hb = synthetic.host_batch(2, 32, "DD", 1000)
    xd, adj = synthetic.to_dense(hb)
    torch.manual_seed(0)
    ga = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes", per_graph_features=True).to(dev)
    x32, g32 = ga.packed_batch(xd.to(dev), adj.to(dev), hb["sizes"])
    lab = torch.from_numpy(hb["label"]).to(dev)
    ta, _, _ = step_us(ga, lambda: ga.loss(ga(x32, g32)[1], lab), iters=100)
    d = D()
    d.x, d.edge_index, d.batch, lab2 = synthetic.to_pyg(hb, dev)
    torch.manual_seed(0)
    gb = pyg.GatNet(89, 64, 2, heads=4, num_layers=2).to(dev).train()
    tb, _, _ = step_us(gb, lambda: gb.loss(d, lab2), iters=100)
    print("DD GAT  b32  2L 4 heads h64 (%5d rows): surface A DGATEncoderGraph %.1f us/step (%.0f graphs/s) | surface B GatNet[GATConv] %.1f us/step (%.0f graphs/s) | B / A = %.2f"
          % (int(hb["sizes"].sum()), ta, 32 / ta * 1e6, tb, 32 / tb * 1e6, tb / ta), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "GAT":
        gat()
        sys.exit(0)
    main()
