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


def sagpool():
    """config 4 as worded: IMDB-B SAGPooling(0.5) + SAGEConv h = 128, batch 128 — composed per level (host round trips for the data-
    dependent sizes, like PyG) beside the reference-surface network (sag_layers.Net: SAGPool + GCNConv as one sync-free node)"""
    import numpy as np
    from two_stage_gnn_amd import sag_layers as S
    dev = torch.device("cuda")
    hb = synthetic.host_batch(3, 128, "IMDB-BINARY", 136)
    sizes = hb["sizes"]; n = int(sizes.sum())
    rp, col = hb["rowptr"][: n + 1], hb["col"]
    dst = np.repeat(np.arange(n), np.diff(rp))
    d = D()
    d.edge_index = torch.from_numpy(np.stack([col.astype(np.int64), dst.astype(np.int64)])).to(dev)
    d.x = torch.ones(n, 1, device=dev)
    d.batch = torch.repeat_interleave(torch.arange(128), torch.from_numpy(sizes)).to(dev)
    lab = torch.from_numpy(hb["label"]).to(dev)
    torch.manual_seed(0)
    na = S.Net(1, 128, 2, 0.5, 0.0, use_batch=True).to(dev).train()
    ta, _, _ = step_us(na, lambda: mp.nll_loss(na(d), lab), iters=100)
    torch.manual_seed(0)
    nc = S.Net(1, 128, 2, 0.5, 0.0, use_batch=True, conv="sage").to(dev).train()
    tc, _, _ = step_us(nc, lambda: mp.nll_loss(nc(d), lab), iters=100)
    print("IMDB-B   b128 h128 ratio .5 (%5d rows): surface A Net[SAGPool + GCNConv] %.1f us/step (%.0f graphs/s) | config 4 as worded, "
          "Net[SAGPool + SAGEConv] as one sync-free node %.1f us/step (%.0f graphs/s) | B / A = %.2f"
          % (n, ta, 128 / ta * 1e6, tc, 128 / tc * 1e6, tc / ta), flush=True)
    torch.manual_seed(0)
    nb = pyg.SagePoolNet(1, 128, 2, 0.5).to(dev).train()
    opt = torch.optim.Adam(nb.parameters(), lr=1e-3)

    def eager():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.nll_loss(nb(d), lab).backward()
        torch.nn.utils.clip_grad_norm_(nb.parameters(), 2.0)
        opt.step()
    for _ in range(5):
        eager()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        eager()
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / 20 * 1e6
    print("IMDB-B   b128 h128 ratio .5 (%5d rows): surface A Net[SAGPool + GCNConv] %.1f us/step (%.0f graphs/s, one hipGraph) | surface B "
          "SagePoolNet[SAGPooling + SAGEConv] %.0f us/step EAGER (%.0f graphs/s; three host round trips per level for k / E' / batch) | B / A = %.1f"
          % (n, ta, 128 / ta * 1e6, tb, 128 / tb * 1e6, tb / ta), flush=True)


def diffpool():
    """pyg.dense_diff_pool (operator forward + backward, eager and replayed from a hipGraph) at BASELINE config 5's first pooling level,
    beside the same operator with the link loss written as PyG writes it (the [B, N, N] product s s^T, a subtraction, a norm)"""
    import numpy as np
    from two_stage_gnn_amd.diffpool import bmm, diffpool_contract_dense, row_softmax
    dev = torch.device("cuda")
    B, N, K, Fd = 16, 512, 64, 64
    hb = synthetic.host_batch(seed=2, B=B, shape="DD", nmax=N)
    _, adj = synthetic.to_dense(hb)
    adj = adj.to(dev)
    mask = (torch.arange(N)[None, :] < torch.from_numpy(np.asarray(hb["sizes"]))[:, None]).to(dev)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, N, Fd, generator=g).to(dev).requires_grad_(True)
    s = torch.randn(B, N, K, generator=g).to(dev).requires_grad_(True)

    def fused():
        o = pyg.dense_diff_pool(x, adj, s, mask)
        return (o[0].sum() + o[1].sum()) * 1e-3 + o[2] + o[3]

    def as_written():
        ss = row_softmax(s.reshape(B * N, K)).reshape(B, N, K)
        m = mask.view(B, N, 1).to(x.dtype)
        xx, ss = x * m, ss * m
        out, out_adj = diffpool_contract_dense(ss, xx, adj)
        link = torch.norm(adj - bmm(ss, ss, trans_b=True), p=2) / adj.numel()
        ent = (-ss * torch.log(ss + 1e-15)).sum(dim=-1).mean()
        return (out.sum() + out_adj.sum()) * 1e-3 + link + ent

    St = torch.cuda.Stream()
    for name, fn in (("closed-form link loss, softmax + mask + entropy in one launch each way", fused),
                     ("as PyG writes it (s s^T [B, N, N], subtraction, norm; element-wise entropy) on the same kernels", as_written)):
        with torch.cuda.stream(St):
            def step():
                x.grad = s.grad = None
                fn().backward()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=St):
                step()
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(St)
            for _ in range(100):
                gr.replay()
            e1.record(St); e1.synchronize()
        print("dense_diff_pool b16 N512 -> K64 h64, forward + backward from one hipGraph: %6.1f us  (%s)" % (e0.elapsed_time(e1) / 100 * 1e3, name),
              flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "DIFFPOOL":
        diffpool()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "GAT":
        gat()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "SAGPOOL":
        sagpool()
        sys.exit(0)
    main()
