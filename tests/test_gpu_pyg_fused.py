"""GPU parity of the FUSED torch_geometric-named layers (csrc/sageconv.hip, pyg_sage.py) against oracle/pyg_ref.py.

PARITY UNPINNED: torch_geometric is absent from /root/reference and from this image (SAGEConv / GATConv / SAGPooling /
dense_diff_pool have no call site in the reference); the oracle restates PyG's documented formulas.  Gradients are arbitrated by an
fp64 run of the same oracle (the discipline of tests/test_gpu_encoders.py::assert_grads_arbitrated)."""
import numpy as np
import pytest
import torch

from oracle import pyg_ref as P
from test_gpu_pyg import grads, rand_graph, tie_free

pytestmark = pytest.mark.gpu


def assert_arbitrated(hip, c32, c64, what, handful=4):
    """|hip - fp64| <= max(10 |cpu32 - fp64|, 2e-5 max|fp64|) per tensor AND at most a handful of entries further than 1e-4 max|fp64|"""
    hip, c32, c64 = hip.detach().cpu().double(), c32.detach().double(), c64.detach().double()
    scale = float(c64.abs().max()) + 1e-30
    e_hip, e_cpu = (hip - c64).abs(), (c32 - c64).abs()
    assert float(e_hip.max()) <= max(10 * float(e_cpu.max()), 2e-5 * scale), (what, float(e_hip.max()), float(e_cpu.max()), scale)
    assert int((e_hip > 1e-4 * scale).sum()) <= max(handful, 4 * int((e_cpu > 1e-4 * scale).sum())), what


def _hub_graph(seed, n, e, hubs=3, hub_deg=40):
    """symmetric random graph with a few rows of > 16 neighbours (CSR tail of the neighbour table) and isolated nodes"""
    ei = rand_graph(seed, n - 5, e, True)                       # the last five nodes stay isolated
    g = torch.Generator().manual_seed(seed + 100)
    extra = []
    for hnode in range(hubs):
        nb = torch.randperm(n - 5, generator=g)[:hub_deg]
        nb = nb[nb != hnode]
        extra.append(torch.stack([nb, torch.full_like(nb, hnode)]))
        extra.append(torch.stack([torch.full_like(nb, hnode), nb]))
    ei = torch.cat([ei] + extra, dim=1)
    code = torch.unique(ei[0] * n + ei[1])
    return torch.stack([code // n, code % n])


@pytest.mark.parametrize("fin,fout,normalize,aggr", [
    (128, 128, False, "mean"), (89, 128, False, "mean"), (3, 128, False, "mean"), (7, 64, True, "mean"), (64, 32, False, "mean"),
    (20, 100, True, "mean"), (128, 128, False, "add"),
])
def test_sage_conv_fused_vs_oracle(fin, fout, normalize, aggr):
    """one fused launch forward, slabs + reduction + one fused launch backward; ragged panel (n % 32 != 0), hub rows, isolated nodes"""
    from two_stage_gnn_amd import pyg
    n = 1000 + 13
    ei = _hub_graph(3, n, 2600)
    x = tie_free(4, n, fin)
    torch.manual_seed(7)
    m = (pyg.SAGEConv(fin, fout, normalize=normalize) if aggr == "mean" else pyg.GraphConv(fin, fout)).cuda()
    ps = [m.lin_l.weight, m.lin_l.bias, m.lin_r.weight]
    gy = tie_free(5, n, fout)

    def oracle(dtype):
        xr = x.to(dtype).requires_grad_(True)
        pr = [p.detach().cpu().to(dtype).requires_grad_(True) for p in ps]
        fn = P.sage_conv if aggr == "mean" else P.graph_conv
        out = fn(xr, ei, *pr)
        if normalize:
            out = torch.nn.functional.normalize(out, p=2.0, dim=-1)
        return out, grads((out * gy.to(dtype)).sum(), [xr] + pr)

    r32, g32 = oracle(torch.float32)
    r64, g64 = oracle(torch.float64)
    xg = x.cuda().requires_grad_(True)
    out = m(xg, ei.cuda())
    assert out.shape == (n, fout)
    assert_arbitrated(out, r32, r64, "out")
    torch.testing.assert_close(out.detach().cpu(), r32.detach(), rtol=1e-4, atol=1e-4)
    gg = grads((out * gy.cuda()).sum(), [xg] + ps)
    for name, a, b, c in zip(["dx", "dW_l", "db_l", "dW_r"], gg, g32, g64):
        assert_arbitrated(a, b, c, name)


def test_sage_conv_directed_edge_list():
    """A^T != A: the input gradient gathers through the transposed table"""
    from two_stage_gnn_amd import pyg
    n, fin, fout = 500, 16, 32
    ei = rand_graph(11, n, 2000, False)
    x = tie_free(12, n, fin)
    torch.manual_seed(3)
    m = pyg.SAGEConv(fin, fout).cuda()
    ps = [m.lin_l.weight, m.lin_l.bias, m.lin_r.weight]
    gy = tie_free(13, n, fout)
    xr = x.double().requires_grad_(True)
    pr = [p.detach().cpu().double().requires_grad_(True) for p in ps]
    ref = P.sage_conv(xr, ei, *pr)
    xg = x.cuda().requires_grad_(True)
    out = m(xg, ei.cuda())
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-5)
    for a, c in zip(grads((out * gy.cuda()).sum(), [xg] + ps), grads((ref * gy.double()).sum(), [xr] + pr)):
        torch.testing.assert_close(a.cpu().double(), c, rtol=1e-4, atol=1e-4)


class _D:
    pass


def _batch(seed, sizes, e_per_node, fin):
    n = int(sum(sizes))
    ei = rand_graph(seed, n, int(e_per_node * n), True, list(sizes))
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(list(sizes)))
    x = tie_free(seed + 1, n, fin)
    return x, ei, batch


@pytest.mark.parametrize("fin,hid,L,sizes", [(7, 64, 2, (18, 30, 9, 41, 17)), (3, 128, 3, (39, 120, 8, 64, 33, 250)), (128, 128, 4, (70, 300))])
def test_sage_net_fused_vs_oracle_and_composed(fin, hid, L, sizes):
    """SageNet (conv stack as ONE node, readouts in the layers' epilogues, fused head) against the oracle and against the same
    modules composed op by op (fused=False)"""
    from two_stage_gnn_amd import pyg
    x, ei, batch = _batch(31, sizes, 2.2, fin)
    lab = torch.arange(len(sizes)) % 2
    torch.manual_seed(9)
    net = pyg.SageNet(fin, hid, 2, num_layers=L).cuda().eval()
    d = _D(); d.x, d.edge_index, d.batch = x.cuda(), ei.cuda(), batch.cuda()
    from two_stage_gnn_amd import pyg_sage as ps
    assert ps.stack_ok(net.graph(d), list(net.convs), d.x)
    names = [k for k, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]

    def oracle(dtype):
        p = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in net.state_dict().items()}
        y = P.sage_net(p, x.to(dtype), ei, batch, L)
        return y, grads(torch.nn.functional.nll_loss(y, lab), [p[k] for k in names])

    y32, g32 = oracle(torch.float32)
    y64, g64 = oracle(torch.float64)
    y = net(d)
    assert_arbitrated(y, y32, y64, "log-probabilities")
    gg = grads(torch.nn.functional.nll_loss(y, lab.cuda()), params)
    for k, a, b, c in zip(names, gg, g32, g64):
        assert_arbitrated(a, b, c, k)
    # the same modules composed launch by launch
    net.fused = False
    y2 = net(d)
    g2 = grads(torch.nn.functional.nll_loss(y2, lab.cuda()), params)
    torch.testing.assert_close(y2, y, rtol=1e-5, atol=1e-5)
    for k, a, b in zip(names, gg, g2):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5, msg=k)
    # a second forward on the same batch starts from clean readout accumulators
    net.fused = True
    torch.testing.assert_close(net(d), y, rtol=0, atol=0)


# ------------------------------------------------------------------------------------------------ GATConv / GatNet
def _edge_order_mult(g, mult):
    """the dropout multiplier [nnz, H] of a GATConv forward (CSR entry order) in the order of the self-looped edge list"""
    out = torch.empty_like(mult)
    out[g.eid[: g.nnz].long()] = mult[: g.nnz]
    return out.cpu()


@pytest.mark.parametrize("dropout", [0.0, 0.4])
def test_gat_net_vs_oracle(dropout):
    """pyg.GatNet (per-target edge softmax, self loops, heads concatenated then averaged, attention dropout with the SAME mask handed
    to the oracle) on a small batch: log-probabilities and every parameter gradient, fp64-arbitrated"""
    from two_stage_gnn_amd import pyg
    sizes = (40, 90, 23, 64)
    x, ei, batch = _batch(41, sizes, 2.5, 12)
    lab = torch.arange(len(sizes)) % 2
    torch.manual_seed(11)
    net = pyg.GatNet(12, 16, 2, heads=4, num_layers=2, dropout=dropout).cuda().train()
    with torch.no_grad():
        for c in net.convs:
            c.bias.copy_(0.1 * torch.randn_like(c.bias))
    d = _D(); d.x, d.edge_index, d.batch = x.cuda(), ei.cuda(), batch.cuda()
    names = [k for k, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    y = net(d)
    g = net.graph(d)
    mults = [_edge_order_mult(g, c.last_drop_mult) for c in net.convs] if dropout > 0 else None

    def oracle(dtype):
        p = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in net.state_dict().items()}
        mm = None if mults is None else [m.to(dtype) for m in mults]
        yy = P.gat_net(p, x.to(dtype), ei, batch, 2, 4, mm)
        return yy, grads(torch.nn.functional.nll_loss(yy, lab), [p[k] for k in names])

    y32, g32 = oracle(torch.float32)
    y64, g64 = oracle(torch.float64)
    assert_arbitrated(y, y32, y64, "log-probabilities")
    gg = grads(torch.nn.functional.nll_loss(y, lab.cuda()), params)
    for k, a, b, c in zip(names, gg, g32, g64):
        assert_arbitrated(a, b, c, k)


# ------------------------------------------------------------------------------------------------ the SAGPool + SAGEConv stack's helpers
@pytest.mark.parametrize("feat,ld", [(1, 8), (3, 4), (128, 256), (64, 64), (20, 24)])
@pytest.mark.parametrize("with_rowend", [False, True])
def test_propagate_mean_both_forms(feat, ld, with_rowend):
    """tsgnn_propagate_mean_f32: y = D^-1 A x (+ xself) and its adjoint A D^-1 x (+ xself) on a symmetric list, coefficients from the row
    lengths; rowend form = rows with slack after them (what the per-graph pooling kernel leaves); isolated rows divide by max(len, 1)"""
    from two_stage_gnn_amd import _native as nat
    n = 700
    ei = _hub_graph(5, n, 1800)
    A = torch.zeros(n, n, dtype=torch.float64)
    A[ei[1], ei[0]] = 1.0
    deg = A.sum(1)
    order = torch.argsort(ei[1] * n + ei[0])
    dst, src = ei[1][order], ei[0][order]
    counts = torch.bincount(dst, minlength=n)
    if with_rowend:                                                   # three slack entries after every row
        rowptr = torch.cumsum(counts + 3, 0) - (counts + 3)
        rowend = rowptr + counts
        col = torch.full((int((counts + 3).sum()),), -7, dtype=torch.int64)
        pos = rowptr[dst] + (torch.arange(dst.numel()) - (torch.cumsum(counts, 0) - counts)[dst])
        col[pos] = src
        rp, re = rowptr.int().cuda(), rowend.int().cuda()
    else:
        rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
        col = src
        rp, re = rowptr.int().cuda(), None
    col = col.int().cuda()
    x = torch.zeros(n, ld)
    x[:, :feat] = tie_free(6, n, feat)
    xs = torch.zeros(n, ld)
    xs[:, :feat] = tie_free(7, n, feat)
    xd, xsd = x.cuda(), xs.cuda()
    inv = 1.0 / deg.clamp(min=1.0)
    for transpose, with_self in ((0, False), (1, True), (0, True), (1, False)):
        y = torch.full((n, ld), 3.0, device="cuda")
        nat.call("propagate_mean_f32", rp, re, col, transpose, xd, ld, xsd if with_self else None, ld if with_self else 0, y, ld, n, feat)
        torch.cuda.synchronize()
        xx = x[:, :feat].double()
        ref = (inv.view(-1, 1) * (A @ xx)) if transpose == 0 else (A @ (inv.view(-1, 1) * xx))
        if with_self:
            ref = ref + xs[:, :feat].double()
        torch.testing.assert_close(y[:, :feat].cpu().double(), ref, rtol=1e-5, atol=1e-5)
        if ld > feat and feat % 4:
            assert float((y[:, feat:].cpu() - 3.0).abs().max()) == 0.0          # columns beyond feat are left alone


def test_copy2d_multi_and_score_rows_in_the_reduction():
    """tsgnn_copy2d_multi_f32 (zero-padded placement of several matrices in one launch) and the K = 0 sets of
    tsgnn_sage_wgrad_reduce_oi_f32 (column sums of the score layer's partial rows beside the weight slabs, with |grad|^2 shares)"""
    from two_stage_gnn_amd import _native as nat, pyg_sage as ps
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    srcs = [torch.randn(128, 1, generator=g), torch.randn(128, 128, generator=g), torch.randn(64, 7, generator=g)]
    outs, words = [], [len(srcs)]
    dsrc = [s.to(dev) for s in srcs]
    for s in dsrc:
        Kp = (s.size(1) + 3) // 4 * 4
        o = torch.full((s.size(0), 2 * Kp), 9.0, device=dev)
        words += [s.data_ptr(), s.stride(0), s.size(0), s.size(1), o.data_ptr() + 4 * Kp, o.stride(0), Kp]
        outs.append(o)
    nat.call("copy2d_multi_f32", np.asarray(words, dtype=np.int64).ctypes.data)
    torch.cuda.synchronize()
    for s, o in zip(srcs, outs):
        Kp = o.size(1) // 2
        assert float((o[:, :Kp].cpu() - 9.0).abs().max()) == 0.0
        assert torch.equal(o[:, Kp:Kp + s.size(1)].cpu(), s)
        assert int(o[:, Kp + s.size(1):].count_nonzero()) == 0
    # a weight set and a score-row set in one reduction
    nslab, K, N, nb, F = 5, 6, 8, 37, 128
    ws = torch.randn(nslab, K + 1, N, generator=g)
    part = torch.randn(nb, F + 4, generator=g)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    dws, dbs = torch.zeros(F + 3, device=dev), torch.zeros(2, device=dev)
    wsd, partd = ws.to(dev), part.to(dev)
    sets = [(wsd, nslab, K, N, dw, db), (partd, nb, 0, F + 4, None, dws, F, dbs)]
    ps.reduce_oi(sets)
    torch.cuda.synchronize()
    tot = ws.double().sum(0)
    torch.testing.assert_close(dw.cpu().double(), tot[:K].t(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db.cpu().double(), tot[K], rtol=1e-5, atol=1e-5)
    col = part.double().sum(0)
    torch.testing.assert_close(dws[:F].cpu().double(), col[:F], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dbs[:1].cpu().double(), col[F:F + 1], rtol=1e-5, atol=1e-5)
    assert float(dws[F:].abs().max()) == 0.0 and float(dbs[1]) == 0.0                 # nothing written past the two outputs


def test_sagpool_sage_net_fused_vs_composed_small():
    """sag_layers.Net(conv="sage") as one node against the same modules composed op by op (its own SAGPool / pyg.SAGEConv classes) and
    against oracle/pyg_ref.sag_net(conv="sage"), on a small batch with a 7-column input (K % 4 != 0: padded halves of the concatenation)"""
    from two_stage_gnn_amd import sag_layers as S
    sizes = (18, 30, 9, 41, 17, 26)
    x, ei, batch = _batch(51, sizes, 2.2, 7)
    lab = torch.arange(len(sizes)) % 2
    torch.manual_seed(5)
    net = S.Net(7, 64, 2, 0.5, 0.0, use_batch=True, conv="sage").cuda().train()
    d = _D(); d.x, d.edge_index, d.batch = x.cuda(), ei.cuda(), batch.cuda()
    assert net._fused_ok()
    names = [k for k, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    y = net(d)
    gg = grads(torch.nn.functional.nll_loss(y, lab.cuda()), params)
    p64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in net.state_dict().items()}
    y64 = P.sag_net(p64, x.double(), ei, 0.5, batch, conv="sage")
    g64 = grads(torch.nn.functional.nll_loss(y64, lab), [p64[k] for k in names])
    torch.testing.assert_close(y.detach().cpu().double(), y64.detach(), rtol=1e-4, atol=1e-5)
    for k, a, c in zip(names, gg, g64):
        torch.testing.assert_close(a.cpu().double(), c, rtol=2e-4, atol=1e-5, msg=k)
    net.fused = False
    y2 = net(d)
    g2 = grads(torch.nn.functional.nll_loss(y2, lab.cuda()), params)
    torch.testing.assert_close(y2, y, rtol=1e-5, atol=1e-5)
    for k, a, b in zip(names, gg, g2):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5, msg=k)
