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


@pytest.mark.parametrize("improved,weighted,self_loops", [(True, False, False), (False, True, False), (True, True, True)])
def test_gcn_conv_edge_weight_and_improved(improved, weighted, self_loops):
    """PyG GCNConv's options the reference never passes (network.py:34): per-edge weights (existing self loops keep theirs) and
    improved = self loops of weight 2, forward and backward (x, W, b) against the oracle's gcn_norm"""
    from two_stage_gnn_amd import pyg
    n, fin, fout = 200, 12, 16
    ei = rand_graph(7, n, 700, False)
    if self_loops:
        loops = torch.arange(0, n, 7)
        ei = torch.cat([ei, torch.stack([loops, loops])], dim=1)
    ew = (0.25 + torch.rand(ei.size(1), generator=torch.Generator().manual_seed(8))) if weighted else None
    x = tie_free(9, n, fin)
    m = pyg.GCNConv(fin, fout, improved=improved).cuda()
    with torch.no_grad():
        m.bias.copy_(tie_free(10, fout))
    w, b = m.weight.detach().cpu().requires_grad_(True), m.bias.detach().cpu().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ref = P.gcn_conv(xr, ei, w, b, edge_weight=ew, improved=improved)
    xg = x.cuda().requires_grad_(True)
    out = m(xg, ei.cuda(), ew.cuda() if ew is not None else None)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    gy = tie_free(11, n, fout)
    gr = grads((ref * gy).sum(), [xr, w, b])
    gg = grads((out * gy.cuda()).sum(), [xg, m.weight, m.bias])
    for a, c in zip(gg, gr):
        torch.testing.assert_close(a.cpu(), c, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("improved,self_loops,sym", [(False, False, False), (True, True, False), (False, False, True)])
def test_gcn_conv_edge_weight_gradient(improved, self_loops, sym):
    """gradient with respect to edge_weight (PyG's gcn_norm is differentiable in the weights; never exercised by the reference,
    Code/sag/layers.py:18): through the normalisation's degrees and the aggregation's sampled product, against the oracle"""
    from two_stage_gnn_amd import pyg
    n, fin, fout = 150, 10, 20
    ei = rand_graph(21, n, 500, sym)
    if self_loops:
        loops = torch.arange(0, n, 5)
        ei = torch.cat([ei, torch.stack([loops, loops])], dim=1)
    ew = 0.25 + torch.rand(ei.size(1), generator=torch.Generator().manual_seed(22))
    x = tie_free(23, n, fin)
    m = pyg.GCNConv(fin, fout, improved=improved).cuda()
    w, b = m.weight.detach().cpu().requires_grad_(True), m.bias.detach().cpu().requires_grad_(True)
    xr, ewr = x.clone().requires_grad_(True), ew.clone().requires_grad_(True)
    ref = P.gcn_conv(xr, ei, w, b, edge_weight=ewr, improved=improved)
    xg, ewg = x.cuda().requires_grad_(True), ew.cuda().requires_grad_(True)
    out = m(xg, ei.cuda(), ewg)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    gy = tie_free(24, n, fout)
    gr = grads((ref * gy).sum(), [xr, w, b, ewr])
    gg = grads((out * gy.cuda()).sum(), [xg, m.weight, m.bias, ewg])
    assert gr[3].abs().max() > 1e-2
    for a, c in zip(gg, gr):
        torch.testing.assert_close(a.cpu(), c, rtol=1e-4, atol=1e-4)


def test_topk_pooling():
    """PyG TopKPooling (imported beside GraphConv, Code/sag/network.py:3, never called): selection, gated features, filtered
    edges and the gradients of x and the projection vector against the oracle"""
    from two_stage_gnn_amd import pyg
    sizes = [23, 1, 40, 9, 31]
    n, f = sum(sizes), 12
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    ei = rand_graph(31, n, 400, True, batch_sizes=sizes)
    x = tie_free(32, n, f)
    m = pyg.TopKPooling(f, ratio=0.5).cuda()
    w = m.weight.detach().cpu().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    rx, rei, rb, rperm, rscore = P.topk_pooling(xr, ei, batch, 0.5, w)
    xg = x.cuda().requires_grad_(True)
    ox, oei, _, ob, operm, oscore = m(xg, ei.cuda(), None, batch.cuda())
    assert torch.equal(operm.cpu(), rperm) and torch.equal(ob.cpu(), rb) and torch.equal(oei.cpu(), rei)
    torch.testing.assert_close(ox.detach().cpu(), rx.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(oscore.detach().cpu(), rscore.detach(), rtol=1e-5, atol=1e-6)
    gy = tie_free(33, rx.size(0), f)
    gs = tie_free(34, rx.size(0))
    gr = grads((rx * gy).sum() + (rscore * gs).sum(), [xr, w])
    gg = grads((ox * gy.cuda()).sum() + (oscore * gs.cuda()).sum(), [xg, m.weight])
    for a, c in zip(gg, gr):
        torch.testing.assert_close(a.cpu(), c, rtol=1e-4, atol=1e-5)
    # one graph, no batch vector, multiplier
    m2 = pyg.TopKPooling(f, ratio=0.3, multiplier=2.0).cuda()
    with torch.no_grad():
        m2.weight.copy_(m.weight)
    e1 = rand_graph(35, 40, 120, True)
    r1 = P.topk_pooling(x[:40], e1, None, 0.3, w.detach())
    o1 = m2(x[:40].cuda(), e1.cuda())
    assert torch.equal(o1[4].cpu(), r1[3])
    torch.testing.assert_close(o1[0].cpu(), 2.0 * r1[0], rtol=1e-5, atol=1e-6)


def test_topk_min_score():
    """PyG topk's threshold mode (never used by the reference): nodes above min(min_score, graph max - 1e-7), node order"""
    from two_stage_gnn_amd import pyg
    sizes = [17, 1, 40, 8, 33]
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    score = tie_free(5, n)
    for ms in (0.0, 0.5, 10.0, -10.0):
        ref = P.topk(score, 0.5, batch, min_score=ms)
        got = pyg.topk(score.cuda(), 0.5, batch.cuda(), min_score=ms)
        np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())
    # (a threshold above every score keeps at most each graph's best node: PyG subtracts its 1e-7 tolerance in float32, so a
    # maximum of magnitude >= 1 is not strictly above it and that graph keeps nothing — reproduced as is)
    counts = np.bincount(batch[P.topk(score, 0.5, batch, min_score=10.0)].numpy(), minlength=len(sizes))
    assert (counts <= 1).all()


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


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("sym", [True, False])
@pytest.mark.parametrize("use_batch", [False, True])
def test_sagpool_net_vs_oracle(use_batch, sym, fused):
    """Code/sag Net: 3 x [GCNConv -> ReLU -> SAGPool -> gmp||gap], IMDB-B-like mini-batch; the sync-free fused node and the
    level-by-level composition of the drop-ins, on symmetric and directed edge lists."""
    from two_stage_gnn_amd import sag_layers as S
    sizes = [20, 12, 31, 20, 9, 25]
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    ei = rand_graph(11, n, 700, sym, sizes)
    x = tie_free(12, n, 5)
    torch.manual_seed(3)
    net = S.Net(5, 32, 2, 0.5, 0.5, use_batch=use_batch, fused=fused).cuda().eval()
    assert net._fused_ok() == fused
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


@pytest.mark.parametrize("case", ["ratio1", "singletons", "isolated", "odd_width", "large_graph", "wide256", "narrow8", "width100"])
def test_sagpool_net_edge_cases(case):
    """edge cases of the sync-free SAGPool levels against the oracle: ratio 1.0 (nothing dropped, the filter is the identity
    up to the top-k order), one-node graphs (k = 1 at every level), nodes without any edge, and a hidden width the float4
    kernels do not take (the model then composes the drop-ins level by level)"""
    from two_stage_gnn_amd import sag_layers as S
    ratio, nhid = 0.5, 32
    sizes = [20, 12, 31, 20, 9, 25]
    if case == "ratio1":
        ratio = 1.0
    elif case == "singletons":
        sizes = [1, 7, 1, 1, 16, 2]
    elif case == "odd_width":
        nhid = 30
    elif case == "large_graph":                      # > 1,024 nodes: the per-graph kernel's bitonic branch (rank count below)
        sizes = [1500, 40, 9]
    elif case == "wide256":                          # 64 lanes per row (the other tests run 8 / 16 / 32)
        nhid = 256
    elif case == "narrow8":
        nhid = 8
    elif case == "width100":                         # 25 float4 columns in a 32-lane group: idle lanes in every group
        nhid = 100
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    ei = rand_graph(51, n, 500 if case != "large_graph" else 6000, True, sizes)
    if case == "singletons":                         # rand_graph gives one-node graphs a self pair that it then removes
        keep = (ei[0] != ei[1])
        ei = ei[:, keep]
    if case == "isolated":
        drop = torch.tensor([0, 5, 33, n - 1])
        keep = ~(torch.isin(ei[0], drop) | torch.isin(ei[1], drop))
        ei = ei[:, keep]
    x = tie_free(52, n, 6)
    torch.manual_seed(8)
    net = S.Net(6, nhid, 3, ratio, 0.5, use_batch=True).cuda().eval()
    assert net._fused_ok() == (case != "odd_width")
    with torch.no_grad():
        for k, p in net.named_parameters():
            if k.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.1)
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}

    class D:
        pass
    d = D(); d.x, d.edge_index, d.batch = x.cuda(), ei.cuda(), batch.cuda()
    out = net(d)
    ref = P.sag_net(p_ref, x, ei, ratio, batch)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    gy = tie_free(53, *ref.shape)
    (ref * gy).sum().backward()
    (out * gy.cuda()).sum().backward()
    for k, p in net.named_parameters():
        r = p_ref[k].grad
        if r is None:
            continue
        err = (p.grad.cpu() - r).abs().max().item()
        assert err <= 2e-3 * r.abs().max().item() + 1e-6, (case, k, err)


@pytest.mark.parametrize("next_prop", [True, False])
def test_sagpool_fused_equals_composed_on_a_large_batch(next_prop, monkeypatch):
    """600 graphs per batch: the sync-free node (per-graph kernels with 600 workgroups, 600 partial rows in the reductions, the
    fused head at B = 600) against the level-by-level composition of the drop-ins, same weights — outputs and gradients; with
    the next level's aggregation / gradient propagate inside the per-graph kernels, and as launches of their own"""
    from two_stage_gnn_amd import sag_layers as S, sag_stack as SS
    monkeypatch.setattr(SS, "FUSED_NEXT_PROPAGATE", next_prop)
    if not next_prop:
        monkeypatch.setattr(SS, "NARROW_FUSED_MAX_ROWS", 0)      # 3 input columns: level 0 as propagate + product launches
    gen = torch.Generator().manual_seed(77)
    sizes = torch.randint(5, 41, (600,), generator=gen).tolist()
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    ei = rand_graph(78, n, 3 * n, True, sizes)
    x = tie_free(79, n, 3)
    torch.manual_seed(5)
    fused = S.Net(3, 64, 2, 0.5, 0.0, use_batch=True).cuda().eval()
    comp = S.Net(3, 64, 2, 0.5, 0.0, use_batch=True, fused=False).cuda().eval()
    comp.load_state_dict(fused.state_dict())

    class D:
        pass
    d = D(); d.x, d.edge_index, d.batch = x.cuda(), ei.cuda(), batch.cuda()
    a, b = fused(d), comp(d)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    gy = tie_free(80, *a.shape).cuda()
    (a * gy).sum().backward()
    (b * gy).sum().backward()
    for (k, p), (_, q) in zip(fused.named_parameters(), comp.named_parameters()):
        err = (p.grad - q.grad).abs().max().item()
        assert err <= 2e-3 * q.grad.abs().max().item() + 1e-6, (k, err)


def _csr_rows(rowptr, col, n):
    rp = rowptr.cpu().numpy()
    c = col.cpu().numpy()
    return [c[rp[i]:rp[i + 1]].tolist() for i in range(n)]


def test_sag_level_kernels_vs_oracle():
    """the device-side pieces of one sync-free level against the PyG restatement: normalised propagation, top-k with the
    relabelling map, CSR filtering (= filter_adj), next-level coefficients, gated gather, max||mean readout"""
    from two_stage_gnn_amd import _native as nat, sag_stack as SS
    from two_stage_gnn_amd.graph import GraphBatch
    sizes = [17, 1, 40, 8, 33, 64]
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    ei = rand_graph(21, n, 500, True, sizes)
    g = GraphBatch.from_edge_index(ei.cuda(), n, ghosts=False)
    dinv, self_w = SS.gcn_coef(g)
    for F in (1, 7, 32, 100):
        x = tie_free(22 + F, n, F)
        ref = P.gcn_conv(x, ei, torch.eye(F), None)
        got, _ = SS.propagate(g.rowptr, g.col, dinv, self_w, x.cuda(), n)
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-6)
    # score-layer mode: (A^ relu(y)) . w + b
    y, w, b = tie_free(30, n, 32), tie_free(31, 32), tie_free(32, 1)
    ref = P.gcn_conv(torch.relu(y), ei, w.view(-1, 1), b).view(-1)
    _, got = SS.propagate(g.rowptr, g.col, dinv, self_w, y.cuda(), n, relu_in=True, w_dot=w.cuda(), dot_bias=b.cuda(), want_y=False)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-5)
    # pooling
    score = tie_free(33, n)
    plan = SS.SagPlan.get(sizes, 0.5, torch.device("cuda"), depth=2)
    L, Ln = plan.levels[0], plan.levels[1]
    K = Ln.N
    perm = torch.empty(K, dtype=torch.int32, device="cuda"); new_id = torch.empty(n, dtype=torch.int32, device="cuda")
    nat.call("topk_segments_f32", score.cuda(), L.gp, Ln.gp, L.B, L.max_seg, perm, new_id)
    ref_perm = P.topk(score, 0.5, batch)
    np.testing.assert_array_equal(perm.cpu().numpy(), ref_perm.numpy())
    inv = torch.full((n,), -1, dtype=torch.int64); inv[ref_perm] = torch.arange(K)
    np.testing.assert_array_equal(new_id.cpu().numpy(), inv.numpy())
    xp = torch.empty(K, 32, device="cuda"); cnt = torch.empty(K, dtype=torch.int32, device="cuda")
    nat.call("sag_pool_gather_f32", y.cuda(), 32, score.cuda(), perm, new_id, g.rowptr, g.col, K, 32, 1, xp, 32, cnt)
    torch.testing.assert_close(xp.cpu(), torch.relu(y)[ref_perm] * torch.tanh(score[ref_perm]).view(-1, 1), rtol=1e-6, atol=1e-6)
    rp_n = torch.empty(K + 1, dtype=torch.int32, device="cuda"); col_n = torch.full((ei.size(1),), -7, dtype=torch.int32, device="cuda")
    dinv_n = torch.empty(K, device="cuda"); sw_n = torch.empty(K, device="cuda")
    nat.call("scan_short_i32", cnt, K, rp_n)
    nat.call("csr_filter_fill", g.rowptr, g.col, perm, new_id, K, rp_n, col_n, dinv_n, sw_n)
    ref_ei = P.filter_adj(ei, ref_perm, n)
    assert int(rp_n[-1]) == ref_ei.size(1)
    g_ref = GraphBatch.from_edge_index(ref_ei.cuda(), K, ghosts=False)
    assert _csr_rows(rp_n, col_n, K) == _csr_rows(g_ref.rowptr, g_ref.col, K)      # filter_adj keeps the edge order
    d_ref, s_ref = SS.gcn_coef(g_ref)
    torch.testing.assert_close(dinv_n, d_ref, rtol=0, atol=0)
    torch.testing.assert_close(sw_n, s_ref, rtol=0, atol=0)
    out = torch.empty(L.B, 64, device="cuda"); arg = torch.empty(L.B, 32, dtype=torch.int32, device="cuda")
    nat.call("sag_readout_f32", xp, 32, Ln.gp, L.B, 32, 0, out, 64, arg)
    b2 = batch[ref_perm]
    ref_out = torch.cat([P.global_max_pool(xp.cpu(), b2, L.B), P.global_mean_pool(xp.cpu(), b2, L.B)], 1)
    torch.testing.assert_close(out.cpu(), ref_out, rtol=1e-6, atol=1e-6)
    nat.call("sag_readout_f32", xp, 32, Ln.gp, L.B, 32, 1, out, 64, arg)
    torch.testing.assert_close(out.cpu(), 2 * ref_out, rtol=1e-6, atol=1e-6)
    # the same level tail as ONE per-graph launch, CSR filter included (rows with explicit ends at the graphs' old bases)
    wsc, bsc = tie_free(34, 32), tie_free(35, 1)
    t_ref = P.gcn_conv(torch.relu(y), ei, wsc.view(-1, 1), bsc).view(-1)
    perm_r = P.topk(t_ref, 0.5, batch)
    sc2 = torch.empty(n, device="cuda"); perm2 = torch.empty(K, dtype=torch.int32, device="cuda"); nid2 = torch.empty(n, dtype=torch.int32, device="cuda")
    xp2 = torch.empty(K, 32, device="cuda"); cnt2 = torch.empty(K, dtype=torch.int32, device="cuda")
    out2 = torch.empty(L.B, 64, device="cuda"); arg2 = torch.empty(L.B, 32, dtype=torch.int32, device="cuda")
    rp2 = torch.empty(K, dtype=torch.int32, device="cuda"); re2 = torch.empty(K, dtype=torch.int32, device="cuda")
    col2 = torch.full((ei.size(1),), -7, dtype=torch.int32, device="cuda")
    d2 = torch.empty(K, device="cuda"); s2 = torch.empty(K, device="cuda"); aggn = torch.empty(K, 32, device="cuda")
    nat.call("sag_pool_graph_f32", y.cuda(), 32, g.rowptr, None, g.col, dinv, self_w, wsc.cuda(), bsc.cuda(), L.gp, Ln.gp, L.B, L.max_seg, 32,
             sc2, perm2, nid2, xp2, 32, cnt2, out2, 64, arg2, 0, rp2, re2, col2, d2, s2, aggn, 32)
    torch.testing.assert_close(sc2.cpu(), t_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(perm2.cpu().numpy(), perm_r.numpy())
    ei_r = P.filter_adj(ei, perm_r, n)
    g_r = GraphBatch.from_edge_index(ei_r.cuda(), K, ghosts=False)
    rp2c, re2c, col2c = rp2.cpu().numpy(), re2.cpu().numpy(), col2.cpu().numpy()
    assert [col2c[rp2c[i]:re2c[i]].tolist() for i in range(K)] == _csr_rows(g_r.rowptr, g_r.col, K)
    d_r, s_r = SS.gcn_coef(g_r)
    torch.testing.assert_close(d2, d_r, rtol=0, atol=0)
    torch.testing.assert_close(s2, s_r, rtol=0, atol=0)
    xr2 = torch.relu(y)[perm_r] * torch.tanh(t_ref[perm_r]).view(-1, 1)
    torch.testing.assert_close(xp2.cpu(), xr2, rtol=1e-5, atol=1e-5)
    b3 = batch[perm_r]
    torch.testing.assert_close(out2.cpu(), torch.cat([P.global_max_pool(xr2, b3, L.B), P.global_mean_pool(xr2, b3, L.B)], 1), rtol=1e-5, atol=1e-5)
    # ... and the next level's aggregation A^' xp from the same launch
    torch.testing.assert_close(aggn.cpu(), P.gcn_conv(xr2, ei_r, torch.eye(32), None), rtol=1e-5, atol=1e-5)
    # the filtered CSR with explicit row ends feeds the propagate kernel
    xk = tie_free(36, K, 32)
    got, _ = SS.propagate(rp2, col2, d2, s2, xk.cuda(), K, rowend=re2)
    torch.testing.assert_close(got.cpu(), P.gcn_conv(xk, ei_r, torch.eye(32), None), rtol=1e-5, atol=1e-6)
    # scan across the single-block tile boundary
    c = torch.randint(0, 9, (9001,), dtype=torch.int32)
    o = torch.empty(9002, dtype=torch.int32, device="cuda")
    nat.call("scan_short_i32", c.cuda(), 9001, o)
    np.testing.assert_array_equal(o.cpu().numpy(), np.concatenate([[0], np.cumsum(c.numpy())]))


def test_sag_level_properties_at_baseline_size():
    """the one-launch level tail on the full BASELINE config-4 batch (IMDB-B-shaped, 128 graphs, H = 128), checked through
    properties instead of the (slow) python oracle: k = ceil(ratio n) rows per graph, scores non-increasing along perm with ties
    towards the smaller node, new_id the inverse of perm, xp = relu(y)[perm] * tanh(score[perm]), readout = max || mean of the kept
    rows, and the filtered adjacency = exactly the kept-kept entries of the input adjacency, relabelled"""
    import numpy as np
    from two_stage_gnn_amd import _native as nat, sag_stack as SS, synthetic
    from two_stage_gnn_amd.graph import GraphBatch
    hb = synthetic.host_batch(3, 128, "IMDB-BINARY", 136)
    sizes = hb["sizes"]; n = int(sizes.sum())
    rp = torch.from_numpy(hb["rowptr"][: n + 1].astype(np.int32)).cuda(); col = torch.from_numpy(hb["col"].astype(np.int32)).cuda()
    dst = torch.repeat_interleave(torch.arange(n, device="cuda"), (rp[1:] - rp[:-1]).long())
    g = GraphBatch.from_edge_index(torch.stack([col.long(), dst]), n, ghosts=False)
    dinv, self_w = SS.gcn_coef(g)
    plan = SS.SagPlan.get(sizes, 0.5, torch.device("cuda"), depth=2)
    L, Ln = plan.levels[0], plan.levels[1]
    K, H = Ln.N, 128
    assert np.array_equal(Ln.sizes, np.ceil(0.5 * sizes).astype(np.int64))
    gen = torch.Generator(device="cuda").manual_seed(8)
    y = torch.randn(n, H, generator=gen, device="cuda"); ws = torch.randn(H, generator=gen, device="cuda"); bs = torch.randn(1, generator=gen, device="cuda")
    i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device="cuda")
    score = torch.empty(n, device="cuda"); perm, new_id, cnt = i32(K), i32(n), i32(K)
    xp = torch.empty(K, H, device="cuda"); out = torch.empty(L.B, 2 * H, device="cuda"); arg = i32(L.B, H)
    rp_n, re_n, col_n = i32(K), i32(K), torch.full((int(g.col.numel()),), -7, dtype=torch.int32, device="cuda")
    d_n, s_n = torch.empty(K, device="cuda"), torch.empty(K, device="cuda")
    nat.call("sag_pool_graph_f32", y, H, g.rowptr, None, g.col, dinv, self_w, ws, bs, L.gp, Ln.gp, L.B, L.max_seg, H, score, perm, new_id, xp, H,
             cnt, out, 2 * H, arg, 0, rp_n, re_n, col_n, d_n, s_n, None, 0)
    # score layer = A^ (relu(y) w) + b through the stand-alone propagate kernel
    _, t_ref = SS.propagate(g.rowptr, g.col, dinv, self_w, y, n, relu_in=True, w_dot=ws, dot_bias=bs, want_y=False)
    torch.testing.assert_close(score, t_ref, rtol=1e-5, atol=1e-5)
    permL, row_graph = perm.long(), L.row_graph.long()
    kept_graph = Ln.row_graph.long()
    assert torch.equal(row_graph[permL], kept_graph)                                          # kept rows stay in their graph
    sp = score[permL]
    same = kept_graph[1:] == kept_graph[:-1]
    assert bool(((sp[:-1] >= sp[1:]) | ~same).all())                                          # descending inside every graph
    ties = same & (sp[:-1] == sp[1:])
    assert bool((permL[:-1][ties] < permL[1:][ties]).all())
    inv = torch.full((n,), -1, dtype=torch.int32, device="cuda"); inv[permL] = torch.arange(K, dtype=torch.int32, device="cuda")
    assert torch.equal(new_id, inv)
    # the smallest kept score of a graph is >= every dropped score of that graph
    big = torch.full((L.B,), float("inf"), device="cuda").scatter_reduce(0, kept_graph, sp, "amin")
    dropped = new_id < 0
    assert bool((score[dropped] <= big[row_graph[dropped]]).all())
    xr = torch.relu(y)[permL] * torch.tanh(sp).unsqueeze(1)
    torch.testing.assert_close(xp, xr, rtol=1e-5, atol=1e-6)
    mx = torch.full((L.B, H), float("-inf"), device="cuda").scatter_reduce(0, kept_graph.unsqueeze(1).expand(K, H), xp, "amax")
    mean = torch.zeros(L.B, H, device="cuda").index_add_(0, kept_graph, xp) / torch.from_numpy(Ln.sizes).cuda().float().unsqueeze(1)
    torch.testing.assert_close(out, torch.cat([mx, mean], 1), rtol=1e-5, atol=1e-5)
    # filtered adjacency: the kept-kept entries, relabelled (as sets of (new row, new col) pairs)
    src_new, dst_new = new_id.long()[g.col.long()], new_id.long()[dst]
    keep = (src_new >= 0) & (dst_new >= 0)
    ref = torch.sort(dst_new[keep] * K + src_new[keep]).values
    lens = (re_n - rp_n).long()
    assert int(lens.sum()) == int(keep.sum())
    rows_new = torch.repeat_interleave(torch.arange(K, device="cuda"), lens)
    pos = rp_n.long().repeat_interleave(lens) + (torch.arange(int(lens.sum()), device="cuda") - (torch.cumsum(lens, 0) - lens).repeat_interleave(lens))
    got = torch.sort(rows_new * K + col_n.long()[pos]).values
    assert torch.equal(got, ref)
    assert torch.equal(cnt.long(), lens)


@pytest.mark.parametrize("F", [64, 128])
def test_gcn_propagate_row_batched_large(F):
    """>= 262,144 rows take the row-batched gather (4 rows per lane group): same result as an index_add formulation,
    plain, with relu on the gathered rows + bias, and in score (dot) mode; ragged tail of rows included"""
    from two_stage_gnn_amd import _native as nat
    n, deg = 262144 + 37, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    cnt = torch.randint(0, 2 * deg + 1, (n,), generator=g, device="cuda", dtype=torch.int32)
    cnt[-1] = 0; cnt[7] = 70                                           # an empty last row, a row longer than a lane group
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device="cuda"); rowptr[1:] = torch.cumsum(cnt, 0)
    nnz = int(rowptr[-1])
    col = torch.randint(0, n, (nnz,), generator=g, device="cuda", dtype=torch.int32)
    dinv = torch.rand(n, generator=g, device="cuda") + 0.5
    self_w = torch.rand(n, generator=g, device="cuda")
    x = torch.randn(n, F, generator=g, device="cuda")
    bias = torch.randn(F, generator=g, device="cuda"); w = torch.randn(F, generator=g, device="cuda"); b0 = torch.randn(1, generator=g, device="cuda")
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), cnt.long())

    def ref(relu):
        xs = torch.relu(x) if relu else x
        acc = torch.zeros(n, F, device="cuda").index_add_(0, rows, dinv[col.long()].unsqueeze(1) * xs[col.long()])
        return dinv.unsqueeze(1) * acc + self_w.unsqueeze(1) * xs
    y = torch.empty_like(x); t = torch.empty(n, device="cuda")
    nat.call("gcn_propagate_f32", rowptr, col, dinv, self_w, x, F, 0, None, None, None, y, F, None, n, F)
    torch.testing.assert_close(y, ref(False), rtol=1e-4, atol=1e-4)
    nat.call("gcn_propagate_f32", rowptr, col, dinv, self_w, x, F, 1, bias, w, b0, y, F, t, n, F)
    r = ref(True) + bias
    torch.testing.assert_close(y, r, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(t, r @ w + b0, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("F,ld", [(1, 1), (3, 3), (8, 12)])
def test_gcn_propagate_narrow_large(F, ld):
    """>= 32,768 rows of <= 8 columns run one thread per row (the degree / constant input column of a large IMDB batch):
    same result as the index_add formulation, with explicit row ends, relu + bias and the score (dot) mode"""
    from two_stage_gnn_amd import _native as nat
    n = 32768 + 11
    g = torch.Generator(device="cuda").manual_seed(9)
    cap = torch.randint(0, 9, (n,), generator=g, device="cuda", dtype=torch.int32)
    cap[-1] = 0; cap[5] = 150
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device="cuda"); rowptr[1:] = torch.cumsum(cap, 0)
    cnt = torch.minimum(cap, torch.randint(0, 9, (n,), generator=g, device="cuda", dtype=torch.int32)); cnt[5] = 150
    rowend = (rowptr[:-1] + cnt).contiguous()
    col = torch.randint(0, n, (int(rowptr[-1]),), generator=g, device="cuda", dtype=torch.int32)
    dinv = torch.rand(n, generator=g, device="cuda") + 0.5
    self_w = torch.rand(n, generator=g, device="cuda")
    xb = torch.randn(n, ld, generator=g, device="cuda"); x = xb[:, :F]
    bias = torch.randn(F, generator=g, device="cuda"); w = torch.randn(F, generator=g, device="cuda"); b0 = torch.randn(1, generator=g, device="cuda")
    for ends in (None, rowend):
        c = cap if ends is None else cnt
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), c.long())
        e = (rowptr[:-1].long().repeat_interleave(c.long()) +
             (torch.arange(int(c.sum()), device="cuda") - (torch.cumsum(c.long(), 0) - c.long()).repeat_interleave(c.long())))
        cj = col[e].long()

        def ref(relu):
            xs = torch.relu(x) if relu else x
            acc = torch.zeros(n, F, device="cuda").index_add_(0, rows, dinv[cj].unsqueeze(1) * xs[cj])
            return dinv.unsqueeze(1) * acc + self_w.unsqueeze(1) * xs
        y = torch.empty(n, F, device="cuda"); t = torch.empty(n, device="cuda")
        nat.call("gcn_propagate_re_f32", rowptr, ends, col, dinv, self_w, xb, ld, 0, None, None, None, y, F, None, n, F)
        torch.testing.assert_close(y, ref(False), rtol=1e-4, atol=1e-4)
        nat.call("gcn_propagate_re_f32", rowptr, ends, col, dinv, self_w, xb, ld, 1, bias, w, b0, y, F, t, n, F)
        r = ref(True) + bias
        torch.testing.assert_close(y, r, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(t, r @ w + b0, rtol=1e-4, atol=1e-3)


def test_sag_step_replays_from_a_hipgraph():
    """the fused SAGPool step has no host round trip: fwd + bwd captured once, replayed on new features"""
    from two_stage_gnn_amd import sag_layers as S
    sizes = [20, 12, 31, 20, 9, 25, 14, 40]
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).cuda()
    ei = rand_graph(41, n, 900, True, sizes).cuda()
    torch.manual_seed(5)
    net = S.Net(4, 64, 2, 0.5, 0.0, use_batch=True).cuda().train()
    lab = (torch.arange(len(sizes)) % 2).cuda()

    class D:
        pass
    d = D(); d.x, d.edge_index, d.batch = tie_free(42, n, 4).cuda(), ei, batch
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        def step():
            net.zero_grad(set_to_none=True)
            out = net(d)
            torch.nn.functional.nll_loss(out, lab).backward()
            return out
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            out_g = step()
        grads_g = {k: p.grad for k, p in net.named_parameters()}
        d.x.copy_(tie_free(43, n, 4).cuda())
        gr.replay()
        torch.cuda.synchronize()
        got_out = out_g.clone()
        got = {k: v.clone() for k, v in grads_g.items()}
        ref_out = step()
        torch.cuda.synchronize()
    torch.testing.assert_close(got_out, ref_out.detach(), rtol=0, atol=0)
    for k, p in net.named_parameters():
        torch.testing.assert_close(got[k], p.grad, rtol=0, atol=0)


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
