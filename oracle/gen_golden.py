#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE — never imported by the product path).

Runs in the BUILD container only: it imports the reference's dense encoders from
/root/reference/Code/sage+gat+diffpool (read-only, never copied), with the hard-coded
``.cuda()`` calls (encoders.py:24-26,132,137) turned into identities *inside this process*,
drives them with seeded synthetic inputs and stores inputs / parameters / outputs / gradients
as small ``.npz`` fixtures under tests/golden/.  The fixtures are data only; no reference
source text is written anywhere.

The GPU box has no /root/reference: tests read only the committed fixtures.

Usage:  python oracle/gen_golden.py            (rewrites tests/golden/*.npz)
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF_DIR = "/root/reference/Code/sage+gat+diffpool"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    # neutralise .cuda() in THIS process only
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF_DIR)
    warnings.filterwarnings("ignore")
    import encoders  # noqa
    import encoders_GAT  # noqa
    # DGATLayer.__init__ references an undefined name (encoders_GAT.py:65); give the module a
    # dummy class so the ctor can run.  No instance ever matches it, so nothing is re-initialised.
    encoders_GAT.DGATHead_V3 = type("DGATHead_V3", (), {})
    return encoders, encoders_GAT


def make_batch(gen, B, nmax, fin, sizes=None, p_edge=0.3, weighted=False):
    """Dense padded batch in the reference's layout (graph_sampler.py:102-114):
    adj[B,Nmax,Nmax] symmetric, zero diagonal, zero-padded; feats[B,Nmax,F] zero-padded."""
    if sizes is None:
        sizes = torch.randint(max(2, nmax // 3), nmax + 1, (B,), generator=gen).tolist()
    adj = torch.zeros(B, nmax, nmax)
    x = torch.zeros(B, nmax, fin)
    for b, n in enumerate(sizes):
        u = torch.rand(n, n, generator=gen)
        a = (torch.triu(u, 1) < p_edge).float() * (torch.triu(torch.ones(n, n), 1))
        if weighted:
            a = a * (0.5 + torch.rand(n, n, generator=gen))
        a = a + a.t()
        adj[b, :n, :n] = a
        x[b, :n] = torch.randn(n, fin, generator=gen)
    return x, adj, np.asarray(sizes, dtype=np.int64)


def randomise_(module, gen, scale=0.5):
    """Replace every parameter by seeded values (the reference leaves torch.FloatTensor
    storage uninitialised before Xavier; biases are 0 at init — we want non-zero biases so the
    ghost-row behaviour is exercised)."""
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * scale)


def sd_np(module, prefix="p."):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="g."):
    out = {}
    for k, p in module.named_parameters():
        out[prefix + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
    return out


def save(name, **arrs):
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_graphconv(enc, gen):
    for tag, add_self, norm, bias, weighted in [
        ("sum_norm_bias", False, True, True, False),
        ("self_nonorm_nobias", True, False, False, False),
        ("weighted_norm_bias", False, True, True, True),
    ]:
        B, nmax, fin, fout = 3, 12, 5, 7
        x, adj, sizes = make_batch(gen, B, nmax, fin, sizes=[12, 7, 4], weighted=weighted)
        m = enc.GraphConv(fin, fout, add_self=add_self, normalize_embedding=norm, bias=bias)
        randomise_(m, gen)
        x.requires_grad_(True)
        y = m(x, adj)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        save("graphconv_" + tag, x=x.detach(), adj=adj, sizes=sizes, gy=gy, y=y.detach(),
             gx=x.grad, add_self=int(add_self), normalize=int(norm), **sd_np(m), **grads_np(m))


def gen_apply_bn(enc, gen):
    # apply_bn (encoders.py:134-138): fresh BatchNorm1d(Nmax) per call, channel = node slot
    m = enc.GcnEncoderGraph(4, 4, 4, 2, 3)
    for tag, B in [("b5", 5), ("b1", 1)]:
        x = torch.randn(B, 9, 6, generator=gen, requires_grad=True)
        y = m.apply_bn(x)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        save("apply_bn_" + tag, x=x.detach(), y=y.detach(), gy=gy, gx=x.grad)


def gen_gcn_encoder(enc, gen):
    class A:  # args.bias (encoders.py:59-60)
        bias = True
    for tag, final_dim, bn, L, B, sizes in [
        ("cls_bn_l3", "number_classes", True, 3, 4, [16, 9, 5, 12]),
        ("emb_bn_l3", "output_dim", True, 3, 4, [14, 16, 3, 8]),
        ("pre_nobn_l2", "pretrain", False, 2, 3, [10, 4, 7]),
        ("cls_bn_l4_b1", "number_classes", True, 4, 1, [11]),
    ]:
        nmax, fin, hid, emb, lab = 16, 6, 8, 10, 2
        x, adj, sz = make_batch(gen, B, nmax, fin, sizes=sizes)
        m = enc.GcnEncoderGraph(fin, hid, emb, lab, L, bn=bn, args=A(), final_dim=final_dim)
        randomise_(m, gen, 0.4)
        a, b = m(x, adj, sz)
        label = torch.randint(0, 2, (B,), generator=gen)
        # CE on whichever of the two outputs is the class logits + a probe on the other
        ga = torch.randn(a.shape, generator=gen)
        gb = torch.randn(b.shape, generator=gen)
        loss = (a * ga).sum() + (b * gb).sum()
        loss.backward()
        save("gcn_encoder_" + tag, x=x, adj=adj, sizes=sz, label=label.numpy(), out_a=a.detach(),
             out_b=b.detach(), ga=ga, gb=gb, num_layers=L, bn=int(bn), final_dim=np.array(final_dim),
             dims=np.array([fin, hid, emb, lab]), **sd_np(m), **grads_np(m))


def gen_diffpool(enc, gen):
    class A:
        bias = True
    for tag, B, nmax, ratio, npool, sizes, final_dim in [
        ("p1", 3, 16, 0.25, 1, [16, 6, 11], "number_classes"),
        ("p2", 2, 32, 0.25, 2, [32, 13], "output_dim"),
        ("p1_nomask", 2, 16, 0.25, 1, [16, 9], "number_classes"),
    ]:
        fin, hid, emb, lab, L = 5, 6, 7, 2, 3
        x, adj, sz = make_batch(gen, B, nmax, fin, sizes=sizes)
        m = enc.SoftPoolingGcnEncoder(nmax, fin, hid, emb, lab, L, hid, assign_ratio=ratio,
                                      num_pooling=npool, bn=True, linkpred=False, args=A(),
                                      assign_input_dim=fin, final_dim=final_dim)
        randomise_(m, gen, 0.4)
        bnn = None if tag.endswith("nomask") else sz
        a, b = m(x, adj, bnn, assign_x=x)
        ga = torch.randn(a.shape, generator=gen)
        gb = torch.randn(b.shape, generator=gen)
        ((a * ga).sum() + (b * gb).sum()).backward()
        save("diffpool_" + tag, x=x, adj=adj, sizes=sz, masked=int(bnn is not None), out_a=a.detach(),
             out_b=b.detach(), ga=ga, gb=gb, assign_last=m.assign_tensor.detach(),
             cfg=np.array([nmax, fin, hid, emb, lab, L, npool]), ratio=ratio,
             final_dim=np.array(final_dim), **sd_np(m), **grads_np(m))
    # the inline contraction alone (encoders.py:374-375)
    B, n, k, f = 3, 12, 4, 5
    s = torch.softmax(torch.randn(B, n, k, generator=gen), -1).requires_grad_(True)
    z = torch.randn(B, n, f, generator=gen, requires_grad=True)
    _, adj, _ = make_batch(gen, B, n, 1, sizes=[12, 8, 5])
    adj.requires_grad_(True)
    xo = torch.matmul(torch.transpose(s, 1, 2), z)
    ao = torch.transpose(s, 1, 2) @ adj @ s
    gx = torch.randn(xo.shape, generator=gen)
    ga = torch.randn(ao.shape, generator=gen)
    ((xo * gx).sum() + (ao * ga).sum()).backward()
    save("diffpool_contract", s=s.detach(), z=z.detach(), adj=adj.detach(), x_out=xo.detach(),
         adj_out=ao.detach(), gx=gx, ga=ga, gs=s.grad, gz=z.grad, gadj=adj.grad)


def gen_gat(gat, gen):
    # single head: B=1 (the only batch size at which the reference is meaningful, T4) and B=2
    for tag, B, concat, sizes in [("b1_concat", 1, True, [10]), ("b1_raw", 1, False, [12]),
                                  ("b2_concat", 2, True, [12, 7])]:
        nmax, fin, fout = 12, 5, 6
        x, adj, sz = make_batch(gen, B, nmax, fin, sizes=sizes)
        m = gat.DGATHead(fin, fout, concat=concat)
        randomise_(m, gen)
        x.requires_grad_(True)
        y = m(x, adj)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        save("gat_head_" + tag, x=x.detach(), adj=adj, sizes=sz, y=y.detach(), gy=gy, gx=x.grad,
             concat=int(concat), **sd_np(m), **grads_np(m))
    for tag, concat, heads in [("concat_h3", True, 3), ("mean_h2", False, 2)]:
        B, nmax, fin, fout = 1, 14, 5, 4
        x, adj, sz = make_batch(gen, B, nmax, fin, sizes=[11])
        m = gat.DGATLayer(fin, fout, n_heads=heads, concat=concat)
        randomise_(m, gen)
        x.requires_grad_(True)
        y = m(x, adj)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        save("gat_layer_" + tag, x=x.detach(), adj=adj, sizes=sz, y=y.detach(), gy=gy, gx=x.grad,
             concat=int(concat), heads=heads, **sd_np(m), **grads_np(m))
    for tag, L, heads, final_dim in [("l2", 2, [2, 2], "number_classes"), ("l3", 3, [3, 2, 2], "output_dim")]:
        B, nmax, fin, hid, emb, lab = 1, 15, 6, 4, 5, 2
        x, adj, sz = make_batch(gen, B, nmax, fin, sizes=[13])
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            m = gat.DGATEncoderGraph(fin, hid, emb, lab, None, num_layers=L, num_heads=heads,
                                     neg_input_slopes=[0.2] * L, dropouts=[0.0] * L, final_dim=final_dim)
        randomise_(m, gen, 0.4)
        a, b = m(x, adj, sz)
        ga = torch.randn(a.shape, generator=gen)
        gb = torch.randn(b.shape, generator=gen)
        ((a * ga).sum() + (b * gb).sum()).backward()
        save("gat_encoder_" + tag, x=x, adj=adj, sizes=sz, out_a=a.detach(), out_b=b.detach(), ga=ga,
             gb=gb, num_layers=L, heads=np.array(heads), final_dim=np.array(final_dim),
             dims=np.array([fin, hid, emb, lab]), **sd_np(m), **grads_np(m))


def main():
    enc, gat = _import_reference()
    gen = torch.Generator().manual_seed(20261003)
    gen_graphconv(enc, gen)
    gen_apply_bn(enc, gen)
    gen_gcn_encoder(enc, gen)
    gen_diffpool(enc, gen)
    gen_gat(gat, gen)


if __name__ == "__main__":
    main()
