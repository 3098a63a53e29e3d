"""GPU parity of the torch_geometric-named operators and the SAGPool path against the CPU restatement of
the documented PyG formulas (oracle/pyg_ref.py — parity unpinned by the reference, see its header)."""
import numpy as np
import pytest
import torch

from oracle import pyg_ref as P

pytestmark = pytest.mark.gpu


def rand_graph(seed, n, e, sym=True, batch_sizes=None):
    g = torch.Generator().manual_seed(seed)
    if batch_sizes is None:
        src = torch.randint(0, n, (e,), generator=g); dst = torch.randint(0, n, (e,), generator=g)
    else:
        srcs, dsts, off = [], [], 0
        for nb in batch_sizes:
            eb = max(1, e * nb // n)
            srcs.append(torch.randint(0, nb, (eb,), generator=g) + off)
            dsts.append(torch.randint(0, nb, (eb,), generator=g) + off)
            off += nb
        src, dst = torch.cat(srcs), torch.cat(dsts)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if sym:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    code = torch.unique(src * n + dst)
    return torch.stack([code // n, code % n])


def tie_free(seed, *shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def grads(loss, params):
    gs = torch.autograd.grad(loss, params, allow_unused=True)
    return [g if g is not None else torch.zeros_like(p) for g, p in zip(gs, params)]


@pytest.mark.parametrize("sym", [True, False])
def test_gcn_conv(sym):
    from two_stage_gnn_amd import pyg
    n, fin, fout = 300, 13, 24
    ei = rand_graph(1, n, 900, sym)
    x = tie_free(2, n, fin)
    m = pyg.GCNConv(fin, fout).cuda()
    with torch.no_grad():
        m.bias.copy_(tie_free(3, fout))
    w, b = m.weight.detach().cpu().requires_grad_(True), m.bias.detach().cpu().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ref = P.gcn_conv(xr, ei, w, b)
    xg = x.cuda().requires_grad_(True)
    out = m(xg, ei.cuda())
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    gy = tie_free(4, n, fout)
    gr = grads((ref * gy).sum(), [xr, w, b])
    gg = grads((out * gy.cuda()).sum(), [xg, m.weight, m.bias])
    for a, c in zip(gg, gr):
        torch.testing.assert_close(a.cpu(), c, rtol=1e-4, atol=1e-4)


def test_topk_filter_pools():
    from two_stage_gnn_amd import pyg
    sizes = [17, 1, 40, 8, 33]
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    score = tie_free(5, n)
    for ratio in (0.5, 0.8, 0.25, 1.0):
        ref = P.topk(score, ratio, batch)
        got = pyg.topk(score.cuda(), ratio, batch.cuda())
        np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())
    # batch=None: the whole mini-batch as one graph (reference behaviour, trap T6)
    np.testing.assert_array_equal(pyg.topk(score.cuda(), 0.5, None).cpu().numpy(), P.topk(score, 0.5, torch.zeros(n, dtype=torch.long)).numpy())
    ei = rand_graph(6, n, 400, True, sizes)
    perm = P.topk(score, 0.5, batch)
    ref_ei = P.filter_adj(ei, perm, n)
    got_ei, _ = pyg.filter_adj(ei.cuda(), None, perm.cuda(), num_nodes=n)
    np.testing.assert_array_equal(got_ei.cpu().numpy(), ref_ei.numpy())
    x = tie_free(7, n, 19)
    xg = x.cuda().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    B = len(sizes)
    mx, mn = pyg.global_max_pool(xg, batch.cuda()), pyg.global_mean_pool(xg, batch.cuda())
    torch.testing.assert_close(mx.detach().cpu(), P.global_max_pool(xr, batch, B))
    torch.testing.assert_close(mn.detach().cpu(), P.global_mean_pool(xr, batch, B).detach(), rtol=1e-5, atol=1e-6)
    gy = tie_free(8, B, 19)
    (g1,) = grads(((mx + 2 * mn) * gy.cuda()).sum(), [xg])
    (g2,) = grads(((P.global_max_pool(xr, batch, B) + 2 * P.global_mean_pool(xr, batch, B)) * gy).sum(), [xr])
    torch.testing.assert_close(g1.cpu(), g2, rtol=1e-5, atol=1e-6)


def test_topk_large_segment_and_limits():
    from two_stage_gnn_amd import pyg
    n = 9000                                                # DD b32 pooled as one graph (T6): ~8.6k nodes
    score = tie_free(9, n)
    got = pyg.topk(score.cuda(), 0.5, None).cpu()
    ref = torch.argsort(score, descending=True, stable=True)[:4500]
    np.testing.assert_array_equal(got.numpy(), ref.numpy())
    with pytest.raises(RuntimeError, match="not supported"):
        pyg.topk(torch.randn(20000).cuda(), 0.5, None)


@pytest.mark.parametrize("use_batch", [False, True])
def test_sagpool_net_vs_oracle(use_batch):
    """Code/sag Net: 3 x [GCNConv -> ReLU -> SAGPool -> gmp||gap], IMDB-B-like mini-batch."""
    from two_stage_gnn_amd import sag_layers as S
    sizes = [20, 12, 31, 20, 9, 25]
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    ei = rand_graph(11, n, 700, True, sizes)
    x = tie_free(12, n, 5)
    torch.manual_seed(3)
    net = S.Net(5, 32, 2, 0.5, 0.5, use_batch=use_batch).cuda().eval()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if k.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.1)
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}

    class D:
        pass
    d = D(); d.x, d.edge_index, d.batch = x.cuda(), ei.cuda(), batch.cuda()
    out = net(d)
    ref = P.sag_net(p_ref, x, ei, 0.5, batch if use_batch else None)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    gy = tie_free(13, *ref.shape)
    (ref * gy).sum().backward()
    (out * gy.cuda()).sum().backward()
    for k, p in net.named_parameters():
        r = p_ref[k].grad
        if r is None:
            continue
        err = (p.grad.cpu() - r).abs().max().item()
        assert err <= 2e-3 * r.abs().max().item() + 1e-6, (k, err)


def test_sage_graph_gat_conv_sagpooling():
    from two_stage_gnn_amd import pyg
    n, fin, fout, H = 200, 11, 8, 4
    ei = rand_graph(21, n, 700, True)
    x = tie_free(22, n, fin)
    torch.manual_seed(5)
    # SAGEConv
    m = pyg.SAGEConv(fin, fout).cuda()
    xg = x.cuda().requires_grad_(True); xr = x.clone().requires_grad_(True)
    ps = [m.lin_l.weight, m.lin_l.bias, m.lin_r.weight]
    pr = [p.detach().cpu().requires_grad_(True) for p in ps]
    out, ref = m(xg, ei.cuda()), P.sage_conv(xr, ei, *pr)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    gy = tie_free(23, n, fout)
    for a, c in zip(grads((out * gy.cuda()).sum(), [xg] + ps), grads((ref * gy).sum(), [xr] + pr)):
        torch.testing.assert_close(a.cpu(), c, rtol=1e-4, atol=1e-4)
    # GATConv (per-target softmax), concat and mean
    for concat in (True, False):
        m = pyg.GATConv(fin, fout, heads=H, concat=concat).cuda()
        with torch.no_grad():
            m.bias.copy_(torch.randn_like(m.bias) * 0.1)
        ps = [m.lin_l.weight, m.att_l, m.att_r, m.bias]
        pr = [p.detach().cpu().requires_grad_(True) for p in ps]
        xg = x.cuda().requires_grad_(True); xr = x.clone().requires_grad_(True)
        out = m(xg, ei.cuda())
        ref = P.gat_conv(xr, ei, pr[0], pr[1], pr[2], pr[3], H, concat)
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
        gy = tie_free(24, *ref.shape)
        for a, c in zip(grads((out * gy.cuda()).sum(), [xg] + ps), grads((ref * gy).sum(), [xr] + pr)):
            torch.testing.assert_close(a.cpu(), c, rtol=1e-3, atol=1e-4)
    # SAGPooling (GraphConv scorer)
    sizes = [60, 90, 50]
    batch = torch.repeat_interleave(torch.arange(3), torch.tensor(sizes))
    ei = rand_graph(25, n, 700, True, sizes)
    m = pyg.SAGPooling(fin, ratio=0.5).cuda()
    ps = [m.gnn.lin_l.weight, m.gnn.lin_l.bias, m.gnn.lin_r.weight]
    pr = [p.detach().cpu().requires_grad_(True) for p in ps]
    xg = x.cuda().requires_grad_(True); xr = x.clone().requires_grad_(True)
    xo, eo, _, bo, perm, sc = m(xg, ei.cuda(), None, batch.cuda())
    rx, re, rb, rperm, rsc = P.sag_pooling(xr, ei, batch, 0.5, *pr)
    np.testing.assert_array_equal(perm.cpu().numpy(), rperm.numpy())
    np.testing.assert_array_equal(eo.cpu().numpy(), re.numpy())
    torch.testing.assert_close(xo.detach().cpu(), rx.detach(), rtol=1e-4, atol=1e-5)
    gy = tie_free(26, *rx.shape)
    for a, c in zip(grads((xo * gy.cuda()).sum(), [xg] + ps), grads((rx * gy).sum(), [xr] + pr)):
        torch.testing.assert_close(a.cpu(), c, rtol=1e-4, atol=1e-4)


def test_dense_diff_pool():
    from two_stage_gnn_amd import pyg
    B, N, K, Fd = 3, 40, 8, 12
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, N, Fd, generator=g); s = torch.randn(B, N, K, generator=g)
    adj = (torch.rand(B, N, N, generator=g) < 0.2).float(); adj = ((adj + adj.transpose(1, 2)) > 0).float()
    mask = torch.arange(N)[None, :] < torch.tensor([40, 25, 31])[:, None]
    xr, sr = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
    xg, sg = x.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    ro = P.dense_diff_pool(xr, adj, sr, mask)
    go = pyg.dense_diff_pool(xg, adj.cuda(), sg, mask.cuda())
    for a, c in zip(go, ro):
        torch.testing.assert_close(a.detach().cpu(), c.detach(), rtol=1e-4, atol=1e-5)
    lr = (ro[0] ** 2).sum() + (ro[1] ** 2).sum() + ro[2] + ro[3]
    lg = (go[0] ** 2).sum() + (go[1] ** 2).sum() + go[2] + go[3]
    for a, c in zip(grads(lg, [xg, sg]), grads(lr, [xr, sr])):
        torch.testing.assert_close(a.cpu(), c, rtol=1e-3, atol=1e-3)


def test_shim_namespace_matches_reference_imports():
    import os, sys
    import two_stage_gnn_amd
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "two-stage-gnn_amd", "shim")
    sys.path.insert(0, shim)
    try:
        from torch_geometric.nn import GCNConv, GraphConv, TopKPooling  # noqa  (network.py:2-3)
        from torch_geometric.nn import global_mean_pool as gap, global_max_pool as gmp  # noqa (network.py:4)
        from torch_geometric.nn.pool.topk_pool import topk, filter_adj  # noqa (layers.py:2)
    finally:
        sys.path.remove(shim)
        for k in [k for k in sys.modules if k.startswith("torch_geometric")]:
            del sys.modules[k]
