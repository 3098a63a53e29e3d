"""The fused GAT layer (csrc/gat_fused.hip, gat_fused.py) against the per-op path, the dense oracle and torch products:
packed projection with score columns, the two attention kernels, blocked weight gradients, attention dropout
(encoders_GAT.py:29-49, 68-84)."""
import numpy as np
import pytest
import torch

from oracle import dense_ref as R
from util_graphs import dense_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R_,K,N", [(1000, 89, 264), (9153, 256, 264), (37, 12, 40), (300, 256, 296)])
def test_rowgemm_wide_columns(R_, K, N):
    """C = A . B with 256 < N <= 384 columns (features + 2H score columns): the column-split kernel with a narrow last block"""
    from two_stage_gnn_amd import _native as nat
    torch.manual_seed(0)
    lda = (K + 3) // 4 * 4
    a = torch.zeros(R_, lda, device="cuda")
    a[:, :K] = torch.randn(R_, K, device="cuda")
    b = torch.randn(K, N, device="cuda")
    c = torch.empty(R_, N, device="cuda")
    nat.call("rowgemm_f32", a, lda, b, N, 0, None, c, N, None, R_, K, N, 0, 0)
    ref = (a[:, :K].double() @ b.double())
    err = (c.double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), err
    # and the transposed product dx = dhp . W'^T (reduction over the N = 264 packed columns)
    if K % 4 == 0:
        dx = torch.empty(R_, K, device="cuda")
        nat.call("rowgemm_f32", c, N, b, N, 1, None, dx, K, None, R_, N, K, 0, 0)
        ref2 = c.double() @ b.double().t()
        assert (dx.double() - ref2).abs().max().item() <= 2e-5 * ref2.abs().max().item()


@pytest.mark.parametrize("R_,K,N", [(1000, 89, 264), (9153, 256, 264), (50, 256, 8), (3000, 130, 132), (1, 4, 4), (700, 512, 512)])
def test_wgrad_blocks(R_, K, N):
    """dW = z^T du on 128 x 128 output blocks, one launch + one fixed-order reduction; bitwise reproducible"""
    from two_stage_gnn_amd import gat_fused as gf
    torch.manual_seed(1)
    ldz = (K + 3) // 4 * 4
    z = torch.zeros(R_, ldz, device="cuda")
    z[:, :K] = torch.randn(R_, K, device="cuda")
    du = torch.randn(R_, N, device="cuda")
    dw = gf.wgrad_blocks(z, K, du)
    assert dw is not None and tuple(dw.shape) == (K, N)
    ref = z[:, :K].double().t() @ du.double()
    err = (dw.double() - ref).abs().max().item()
    assert err <= 3e-5 * ref.abs().max().item() + 1e-6, err
    assert torch.equal(dw, gf.wgrad_blocks(z, K, du))


@pytest.mark.parametrize("R_,K,N", [(8518, 256, 264), (1000, 192, 132), (37, 256, 264)])
def test_bwd_products_merged_equals_separate(R_, K, N):
    """the weight-gradient slabs and the input-gradient product of a layer's packed projection in one launch
    (tsgnn_gat_bwd_products_f32) == the two launches they replace, bit for bit, and the torch products"""
    from two_stage_gnn_amd import gat_fused as gf, _native as nat
    gen = torch.Generator(device="cuda").manual_seed(R_ + K)
    x = torch.randn(R_, K, generator=gen, device="cuda")
    du = torch.randn(R_, N, generator=gen, device="cuda")
    wp = torch.randn(K, N, generator=gen, device="cuda") * 0.1
    dw, dx = gf.bwd_products(x, K, du, wp)
    dw_s = gf.wgrad_blocks(x, K, du)
    dx_s = torch.empty(R_, K, device="cuda")
    nat.call("rowgemm_f32", du, du.stride(0), wp, wp.stride(0), 1, None, dx_s, dx_s.stride(0), None, R_, N, K, 0, 0)
    assert torch.equal(dw, dw_s) and torch.equal(dx, dx_s)
    torch.testing.assert_close(dw.double(), x.double().t() @ du.double(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dx.double(), du.double() @ wp.double().t(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("R_,K,N,bias", [(1024, 192, 64, True), (128, 136, 8, True), (5000, 64, 128, False), (77, 300, 260, True)])
def test_wgrad_blocks_linear_layout(R_, K, N, bias):
    """the same product written as torch.nn.Linear's [out, in] gradient with db = colsum(du) from the same pass
    (message_passing._LinearOI.backward: a contiguous dW that AccumulateGrad keeps instead of copying a transposed view)"""
    from two_stage_gnn_amd import message_passing as mp
    torch.manual_seed(2)
    x = torch.randn(R_, K, device="cuda")
    dy = torch.randn(R_, N, device="cuda")
    got = mp.linear_wgrad_oi(x, K, dy, bias)
    assert got is not None
    dw, db = got
    assert tuple(dw.shape) == (N, K) and dw.is_contiguous() and (db is not None) == bias
    ref = dy.double().t() @ x.double()
    assert (dw.double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item() + 1e-6
    if bias:
        refb = dy.double().sum(0)
        assert (db.double() - refb).abs().max().item() <= 3e-5 * refb.abs().max().item() + 1e-6
    # through the autograd node: the parameter's .grad IS the tensor the kernel wrote (no copy), values as torch's Linear
    w = torch.randn(N, K, device="cuda", requires_grad=True)
    b = torch.randn(N, device="cuda", requires_grad=True) if bias else None
    xr = x.clone().requires_grad_(True)
    y = mp.linear_oi(xr, w, b)
    (y * dy).sum().backward()
    w2 = w.detach().clone().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True) if bias else None
    x2 = x.clone().requires_grad_(True)
    (torch.nn.functional.linear(x2.double(), w2.double(), b2.double() if bias else None) * dy.double()).sum().backward()
    assert w.grad.is_contiguous()
    for a_, r_ in ((w.grad, w2.grad), (xr.grad, x2.grad)) + (((b.grad, b2.grad),) if bias else ()):
        assert (a_.double() - r_.double()).abs().max().item() <= 5e-5 * r_.abs().max().item() + 1e-6


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("heads,hid,emb,layers", [([4, 4], 64, 64, 2), ([2, 3, 2], 16, 32, 3), ([8, 8], 8, 4, 2)])
@pytest.mark.parametrize("packed", [True, False])
def test_fused_layer_equals_per_op_path(heads, hid, emb, layers, packed, monkeypatch):
    """the whole encoder on the fused kernels vs the per-op kernels of attention.py: outputs and every parameter gradient
    (packed: rows + one ghost representative per graph, an isolated real node, a full graph; else B = 1 with a few padded rows)"""
    from two_stage_gnn_amd import gat_encoders as G, gat_fused as gf
    if packed:
        B, nmax = 4, 96
        x, adj, sizes = dense_batch(6, B, nmax, 13, sizes=[80, 41, 96, 9], p_edge=0.07)
        adj[1, 5, :] = 0; adj[1, :, 5] = 0
    else:
        B, nmax = 1, 64
        x, adj, sizes = dense_batch(7, B, nmax, 13, sizes=[55], p_edge=0.1)
    torch.manual_seed(3)
    m = G.DGATEncoderGraph(13, hid, emb, 3, None, num_layers=layers, num_heads=heads, final_dim="number_classes",
                           per_graph_features=packed).cuda()
    label = (torch.arange(B) % 3).cuda()
    res = []
    for fused in (True, False):
        monkeypatch.setattr(gf, "FUSED", fused)
        m.zero_grad(set_to_none=True)
        a, b = m(x.cuda(), adj.cuda(), sizes if packed else None)
        m.loss(b, label).backward()
        res.append((a.detach(), b.detach(), _grads(m)))
    (a1, b1, g1), (a0, b0, g0) = res
    torch.testing.assert_close(a1, a0, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(b1, b0, rtol=2e-5, atol=2e-6)
    assert set(g1) == set(g0)
    for k in g0:
        err = (g1[k] - g0[k]).abs().max().item()
        assert err <= 1e-4 * g0[k].abs().max().item() + 1e-8, (k, err, g0[k].abs().max().item())


def test_fused_path_is_the_one_that_runs():
    """the DD batch of the secondary benchmark config goes through gat_attn_fwd / gat_attn_bwd (launch trace)"""
    from two_stage_gnn_amd import gat_encoders as G, _native as nat
    B, nmax = 3, 128
    x, adj, sizes = dense_batch(8, B, nmax, 89, sizes=[100, 128, 60], p_edge=0.04)
    m = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes",
                           per_graph_features=True).cuda()
    nat.trace = []
    try:
        a, b = m(x.cuda(), adj.cuda(), sizes)
        m.loss(b, torch.tensor([0, 1, 0]).cuda()).backward()
        names = [t[0] for t in nat.trace]
    finally:
        nat.trace = None
    assert names.count("gat_pack_f32") == 1 and names.count("gat_unpack_f32") == 1
    assert names.count("gat_attn_fwd_f32") == 2 and names.count("gat_attn_bwd_ro_f32") == 2
    assert names.count("readout_max_fwd_f32") == 1 and names.count("readout_max_bwd_rows_f32") == 0    # (the last layer's node made the readout)
    # layer 2 (256 -> 264): weight-gradient slabs beside the input-gradient product in one launch + the slabs' reduction; layer 1
    # (92 -> 264)
    # (the input features need no gradient): the blocked weight gradient's slabs alone; ONE reduction launch for both layers, issued
    # by the parameter unpack
    assert names.count("gat_bwd_products_f32") == 1 and names.count("wgrad_blocks_slabs_f32") == 1
    assert names.count("wgrad_blocks_reduce2_f32") == 1 and names.index("wgrad_blocks_reduce2_f32") == names.index("gat_unpack_f32") - 1
    assert not any(n in names for n in ("edge_softmax_fwd_f32", "csr_sddmm_heads_f32", "node_scores2_f32"))


def _mults(p, key, B, N, H):
    """[H][B, N, N] multipliers the kernels apply to the attention elements of a padded batch (rows b*N + i); key = (seed,
    device counter snapshot) of the layer call (gat_fused.last_dropout_key)"""
    from two_stage_gnn_amd import _native as nat
    blocks = []
    for b in range(B):
        t = torch.empty(N, N, H, device="cuda")
        nat.call("gat_dropout_mult_f32", float(p), int(key[0]), key[1], b * N, N, b * N, N, H, t)
        blocks.append(t.cpu())
    full = torch.stack(blocks)                                   # [B, N, N, H]
    return [full[..., h].contiguous() for h in range(H)]


@pytest.mark.parametrize("concat", [True, False])
def test_attention_dropout_matches_oracle_with_the_same_mask(concat):
    """encoders_GAT.py:42: F.dropout on the attention matrix.  The kernels' Philox multipliers are handed to the dense oracle;
    outputs and gradients then agree like the no-dropout case (isolated columns' 1/N entries are dropped element-wise too)"""
    from two_stage_gnn_amd import gat_encoders as G, gat_fused as gf
    B, N, fin, Fo, H, p = 1, 48, 12, 16, 4, 0.3
    x, adj, sizes = dense_batch(9, B, N, fin, sizes=[40], p_edge=0.12)
    adj[0, 3, :] = 0; adj[0, :, 3] = 0
    torch.manual_seed(5)
    m = G.DGATLayer(fin, Fo, dropout=p, n_heads=H, concat=concat).cuda().train()
    xg = x.cuda().requires_grad_(True)
    torch.manual_seed(11)                                         # F.dropout on x (:71) draws from torch's generator
    y = m(xg, adj.cuda())
    key = gf.last_dropout_key
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    # oracle: the same input dropout (replayed from the same generator state) and the same attention multipliers
    torch.manual_seed(11)
    xd_keep = torch.nn.functional.dropout(torch.ones_like(xg), p, training=True).cpu()
    p_ref = {"l." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    mult = _mults(p, key, B, N, H)
    yr = R.gat_layer(p_ref, "l", xr * xd_keep, adj, concat, 0.2, att_mult=mult)
    (yr * gy.cpu()).sum().backward()
    frac = sum(float((mm == 0).float().mean()) for mm in mult) / H
    assert abs(frac - p) < 0.03                                   # the mask really drops ~p of the elements
    torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-3, atol=2e-5)
    for k, prm in m.named_parameters():
        ref = p_ref["l." + k].grad
        err = (prm.grad.cpu() - ref).abs().max().item()
        assert err <= 1e-3 * ref.abs().max().item() + 1e-6, (k, err)


def test_attention_dropout_draws_a_new_mask_per_graph_replay():
    """the dropout key lives on the device: a training step captured in a hipGraph advances it at every replay (a host seed
    would be frozen into the capture)"""
    from two_stage_gnn_amd import gat_encoders as G
    B, N = 1, 40
    x, adj, sizes = dense_batch(12, B, N, 12, sizes=[34], p_edge=0.15)
    torch.manual_seed(8)
    m = G.DGATLayer(12, 16, dropout=0.5, n_heads=2, concat=True).cuda().train()
    m.dropout = 0.0                                               # (keep F.dropout on x out of it: only the attention masks vary)
    xs, adjs = x.cuda(), adj.cuda()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            y = m(xs, adjs)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            y = m(xs, adjs)
        outs = []
        for _ in range(3):
            gr.replay(); torch.cuda.synchronize()
            outs.append(y.clone())
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])


def test_attention_dropout_eval_and_batched():
    """eval mode ignores p; a per-graph-features batch under dropout runs un-packed (padded rows are no longer identical) and
    differs from the p = 0 output; two steps draw different masks"""
    from two_stage_gnn_amd import gat_encoders as G
    B, N = 3, 40
    x, adj, sizes = dense_batch(10, B, N, 9, sizes=[30, 40, 12], p_edge=0.15)
    torch.manual_seed(6)
    m = G.DGATEncoderGraph(9, 16, 16, 2, None, num_layers=2, num_heads=[2, 2], dropouts=[0.4, 0.4], final_dim="number_classes",
                           per_graph_features=True).cuda()
    m0 = G.DGATEncoderGraph(9, 16, 16, 2, None, num_layers=2, num_heads=[2, 2], dropouts=[0.0, 0.0], final_dim="number_classes",
                            per_graph_features=True).cuda()
    m0.load_state_dict(m.state_dict())
    m.eval(); m0.eval()
    a_e, b_e = m(x.cuda(), adj.cuda(), sizes)
    a_0, b_0 = m0(x.cuda(), adj.cuda(), sizes)
    assert torch.equal(b_e, b_0) and torch.equal(a_e, a_0)
    m.train()
    torch.manual_seed(1)
    _, b_t1 = m(x.cuda(), adj.cuda(), sizes)
    _, b_t2 = m(x.cuda(), adj.cuda(), sizes)
    assert not torch.allclose(b_t1, b_0) and not torch.equal(b_t1, b_t2)
    m.loss(b_t2, torch.tensor([0, 1, 0]).cuda()).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
