"""Edge cases of the hot path: empty / single-node / isolated-node graphs, B = 1, odd feature widths, maximum sizes,
self loops, duplicate edges — compared with the CPU oracle or with direct dense math."""
import numpy as np
import pytest
import torch

from oracle import dense_ref as R
from oracle import pyg_ref as P

pytestmark = pytest.mark.gpu


def encoder(fin, hid, emb, L, bn=True):
    from two_stage_gnn_amd import dense_encoders as E

    class A:
        bias = True
    torch.manual_seed(0)
    m = E.GcnEncoderGraph(fin, hid, emb, 2, L, bn=bn, args=A(), final_dim="output_dim")
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "conv" in k and k.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.3)
    return m.cuda()


def check_encoder(m, x, adj, sizes, tol=1e-4):
    p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    a_ref, b_ref = R.gcn_encoder(p, x, adj, bn=m.bn, final_dim="output_dim")
    for layout_sizes in (sizes, None):                         # packed (+ghost rows) and padded
        a, b = m(x.cuda(), adj.cuda(), layout_sizes)
        torch.testing.assert_close(a.cpu(), a_ref, rtol=tol, atol=tol)
        torch.testing.assert_close(b.cpu(), b_ref, rtol=tol, atol=tol)


def test_graph_without_edges_and_single_node_graphs():
    B, nmax, fin = 5, 12, 6
    sizes = np.array([1, 12, 3, 1, 7])
    gen = torch.Generator().manual_seed(1)
    x = torch.zeros(B, nmax, fin); adj = torch.zeros(B, nmax, nmax)
    for b, n in enumerate(sizes):
        x[b, :n] = torch.randn(n, fin, generator=gen)
    adj[1, 0, 1] = adj[1, 1, 0] = 1.0                           # one edge in the whole batch; graphs 0,2,3,4 have none
    adj[4, 2, 5] = adj[4, 5, 2] = 1.0
    check_encoder(encoder(fin, 8, 8, 3), x, adj, sizes)


def test_full_size_graphs_no_ghost_candidates_and_b1():
    # every graph fills all Nmax slots (no ghost copy is ever used) ; then B = 1 (triplet setting, tripletnet.py:36)
    from util_graphs import dense_batch
    x, adj, sizes = dense_batch(3, 3, 10, 5, sizes=[10, 10, 10])
    check_encoder(encoder(5, 8, 12, 3), x, adj, sizes)
    x, adj, sizes = dense_batch(4, 1, 20, 5, sizes=[13])
    check_encoder(encoder(5, 8, 8, 4), x, adj, sizes)


def test_odd_widths_fall_back_to_generic_kernels():
    # hidden = 10 (not a multiple of 4): no float4 / fused-stack path anywhere
    from util_graphs import dense_batch
    x, adj, sizes = dense_batch(5, 4, 18, 7, sizes=[18, 5, 9, 14])
    check_encoder(encoder(7, 10, 6, 3), x, adj, sizes)
    check_encoder(encoder(7, 10, 6, 2, bn=False), x, adj, sizes)


def test_many_graphs_beyond_fused_limit():
    # B = 150 > 128: the fused slot kernels do not apply; generic slot kernels must give the same answer
    from util_graphs import dense_batch
    x, adj, sizes = dense_batch(6, 150, 16, 4, p_edge=0.3)
    check_encoder(encoder(4, 8, 8, 3), x, adj, sizes)


def test_self_loops_duplicates_and_isolated_nodes_pyg():
    from two_stage_gnn_amd import pyg
    n = 9
    ei = torch.tensor([[0, 1, 1, 2, 2, 4, 4, 7], [1, 0, 0, 2, 3, 4, 5, 8]])      # duplicate (1,0), self loops on 2 and 4, 6 isolated
    x = torch.randn(n, 5, generator=torch.Generator().manual_seed(2))
    m = pyg.GCNConv(5, 4).cuda()
    ref = P.gcn_conv(x, ei, m.weight.detach().cpu(), m.bias.detach().cpu())
    torch.testing.assert_close(m(x.cuda(), ei.cuda()).detach().cpu(), ref, rtol=1e-5, atol=1e-5)
    s = pyg.SAGEConv(5, 4).cuda()
    ref = P.sage_conv(x, ei, s.lin_l.weight.detach().cpu(), s.lin_l.bias.detach().cpu(), s.lin_r.weight.detach().cpu())
    torch.testing.assert_close(s(x.cuda(), ei.cuda()).detach().cpu(), ref, rtol=1e-5, atol=1e-5)
    # graph with no edges at all
    e0 = torch.zeros(2, 0, dtype=torch.long)
    ref = P.gcn_conv(x, e0, m.weight.detach().cpu(), m.bias.detach().cpu())
    torch.testing.assert_close(m(x.cuda(), e0.cuda()).detach().cpu(), ref, rtol=1e-5, atol=1e-5)


def test_topk_ratio_edge_values_and_tiny_graphs():
    from two_stage_gnn_amd import pyg
    sizes = [1, 2, 3, 1, 5]
    batch = torch.repeat_interleave(torch.arange(5), torch.tensor(sizes))
    score = torch.randn(sum(sizes), generator=torch.Generator().manual_seed(3))
    for ratio in (0.01, 0.34, 0.5, 0.999, 1.0):
        np.testing.assert_array_equal(pyg.topk(score.cuda(), ratio, batch.cuda()).cpu().numpy(), P.topk(score, ratio, batch).numpy())
    # filter_adj that removes every edge
    ei = torch.tensor([[0, 1], [1, 0]])
    out, _ = pyg.filter_adj(ei.cuda(), None, torch.tensor([2]).cuda(), num_nodes=3)
    assert out.shape == (2, 0)


def test_max_nodes_1000_single_dd_graph_gat_and_sage():
    """reference defaults: --max_nodes 1000 (train.py:475), batch_size 1 (train.py:480)"""
    from util_graphs import dense_batch
    x, adj, sizes = dense_batch(8, 1, 1000, 16, sizes=[743], p_edge=0.01)
    check_encoder(encoder(16, 32, 32, 3), x, adj, sizes)
    from two_stage_gnn_amd import gat_encoders as G
    torch.manual_seed(1)
    m = G.DGATEncoderGraph(16, 8, 8, 2, None, num_layers=2, num_heads=[2, 2]).cuda()
    p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    a_ref, b_ref = R.gat_encoder(p, x, adj)
    a, b = m(x.cuda(), adj.cuda(), sizes)
    torch.testing.assert_close(a.cpu(), a_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.cpu(), b_ref, rtol=1e-4, atol=1e-4)


def test_weighted_normalised_adjacency_dropin():
    """GraphSampler(normalize=True) ships D^-1/2 A D^-1/2 (graph_sampler.py:29-31): values must survive the CSR ingest"""
    from util_graphs import dense_batch
    from two_stage_gnn_amd import dense_encoders as E
    x, adj, sizes = dense_batch(9, 3, 150, 6, sizes=[150, 90, 120], p_edge=0.05)
    deg = adj.sum(-1).clamp(min=1)
    adjn = adj / deg.sqrt().unsqueeze(-1) / deg.sqrt().unsqueeze(-2)
    m = E.GraphConv(6, 8, normalize_embedding=True).cuda()
    ref = R.graph_conv(x, adjn, m.weight.detach().cpu(), m.bias.detach().cpu(), normalize=True)
    torch.testing.assert_close(m(x.cuda(), adjn.cuda()).detach().cpu(), ref, rtol=1e-4, atol=1e-5)


def test_errors_are_loud():
    from two_stage_gnn_amd import dense_encoders as E
    from two_stage_gnn_amd.graph import GraphBatch
    with pytest.raises(ValueError):
        GraphBatch.from_dense(torch.zeros(2, 4, 5).cuda())
    with pytest.raises(ValueError):
        GraphBatch.from_dense(torch.zeros(2, 4, 4).cuda(), sizes=[9, 1])
    m = E.GraphConv(3, 4).cuda()
    with pytest.raises(ValueError):
        m(torch.zeros(2, 4).cuda(), torch.zeros(2, 4, 4).cuda())
    g = GraphBatch.from_dense(torch.zeros(2, 4, 4).cuda(), sizes=[2, 3])
    with pytest.raises(ValueError):
        m.forward_rows(torch.zeros(3, 3).cuda(), g)           # wrong number of rows
