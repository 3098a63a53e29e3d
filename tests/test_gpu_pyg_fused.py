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
