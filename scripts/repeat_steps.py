#!/usr/bin/env python3
"""are the secondary configs' steps bitwise repeatable from the FIRST step on a batch?  (GAT one graph / 32 graphs, SAGPool, PROTEINS
SAGE; the DiffPool step: scripts/repeat_diffpool.py).  20 eager steps each on the same batch and parameters."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_gnn_amd import dense_encoders as E, gat_encoders as G, sag_layers as S, synthetic, message_passing as mp
dev = torch.device("cuda")
class A: bias = True

def check(name, model, step):
    res = []
    for it in range(20):
        model.zero_grad(set_to_none=True)
        loss = step()
        res.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    torch.cuda.synchronize(); mp.check_device_errors()
    def same(i, j):
        return torch.equal(res[i][0], res[j][0]) and all(torch.equal(res[i][1][k], res[j][1][k]) for k in res[i][1])
    first = sum(not same(0, i) for i in range(1, 20))
    later = sum(not same(1, i) for i in range(2, 20))
    print("%-34s steps differing from step 0: %2d of 19 ; steps 2.. differing from step 1: %2d of 18" % (name, first, later))

torch.manual_seed(0)
hb = synthetic.host_batch(7, 64, "PROTEINS", 620)
g, x, label = synthetic.to_device(hb, dev)
m = E.GcnEncoderGraph(3, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
def s1():
    l = m.loss(m(x, g)[1], label); l.backward(gradient=mp.unit_seed(dev)); return l
check("cfg2 PROTEINS SAGE b64", m, s1)

hb1 = synthetic.host_batch(2, 1, "DD", 1000)
x1, adj1 = synthetic.to_dense(hb1)
gat = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes").to(dev)
x1, adj1 = x1.to(dev), adj1.to(dev)
lab1 = torch.tensor([1], device=dev)
def s2():
    l = gat.loss(gat(x1, adj1, hb1["sizes"])[1], lab1); l.backward(gradient=mp.unit_seed(dev)); return l
check("cfg3 GAT one graph", gat, s2)

hb32 = synthetic.host_batch(2, 32, "DD", 1000)
x32, adj32 = synthetic.to_dense(hb32)
gat32 = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes", per_graph_features=True).to(dev)
x32, g32 = gat32.packed_batch(x32.to(dev), adj32.to(dev), hb32["sizes"])
lab32 = torch.from_numpy(hb32["label"]).to(dev)
def s3():
    l = gat32.loss(gat32(x32, g32)[1], lab32); l.backward(gradient=mp.unit_seed(dev)); return l
check("cfg3 GAT 32 graphs", gat32, s3)
del adj32

hb4 = synthetic.host_batch(3, 128, "IMDB-BINARY", 136)
sizes = hb4["sizes"]; rp = hb4["rowptr"][: int(sizes.sum()) + 1]; col = hb4["col"]
dst = np.repeat(np.arange(int(sizes.sum())), np.diff(rp))
ei = torch.from_numpy(np.stack([col.astype(np.int64), dst.astype(np.int64)])).to(dev)
class D: pass
d = D(); d.x = torch.ones(int(sizes.sum()), 1, device=dev); d.edge_index = ei
d.batch = torch.repeat_interleave(torch.arange(128), torch.from_numpy(sizes)).to(dev)
lab4 = torch.from_numpy(hb4["label"]).to(dev)
net = S.Net(1, 128, 2, 0.5, 0.0, use_batch=True).to(dev).train()        # (dropout 0: the masks are random by design)
def s4():
    l = torch.nn.functional.nll_loss(net(d), lab4); l.backward(); return l
check("cfg4 SAGPool b128 (dropout 0)", net, s4)
