"""GPU parity of the drop-in encoders against the golden vectors captured from the reference
(tests/golden, made by oracle/gen_golden.py) and against the CPU oracle at larger sizes."""
import numpy as np
import pytest
import torch

from conftest import load_golden, params_of
from oracle import dense_ref as R
from util_graphs import dense_batch, dd_like_sizes

pytestmark = pytest.mark.gpu


def load_state(module, g):
    sd = {k[2:]: torch.tensor(v) for k, v in g.items() if k.startswith("p.")}
    module.load_state_dict(sd, strict=True)
    return module.cuda()


def check_param_grads(module, g, rtol, atol):
    for k, p in module.named_parameters():
        ref = g["g." + k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=k)


@pytest.mark.parametrize("tag", ["sum_norm_bias", "self_nonorm_nobias", "weighted_norm_bias"])
@pytest.mark.parametrize("path", ["dense_small", "csr_padded"])
def test_graphconv_golden(tag, path, monkeypatch):
    from two_stage_gnn_amd import dense_encoders as E
    g = load_golden("graphconv_" + tag)
    if path == "csr_padded":
        monkeypatch.setattr(E, "DENSE_ADJ_MAX_NODES", 0)
    m = E.GraphConv(g["x"].shape[2], g["y"].shape[2], add_self=bool(g["add_self"]),
                    normalize_embedding=bool(g["normalize"]), bias=("p.bias" in g))
    load_state(m, g)
    x = torch.tensor(g["x"]).cuda().requires_grad_(True)
    y = m(x, torch.tensor(g["adj"]).cuda())
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    (y * torch.tensor(g["gy"]).cuda()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx"], rtol=1e-4, atol=1e-5)
    check_param_grads(m, g, 1e-4, 1e-5)


@pytest.mark.parametrize("tag", ["cls_bn_l3", "emb_bn_l3", "pre_nobn_l2", "cls_bn_l4_b1"])
@pytest.mark.parametrize("layout", ["packed", "padded"])
@pytest.mark.parametrize("fused", [True, False])
def test_gcn_encoder_golden(tag, layout, fused, monkeypatch):
    from two_stage_gnn_amd import dense_encoders as E
    monkeypatch.setattr(E, "FUSED_STACK", fused)
    monkeypatch.setattr(E, "FUSED_HEAD", fused)
    g = load_golden("gcn_encoder_" + tag)
    fin, hid, emb, lab = (int(v) for v in g["dims"])

    class A:
        bias = True
    m = E.GcnEncoderGraph(fin, hid, emb, lab, int(g["num_layers"]), bn=bool(g["bn"]), args=A(),
                          final_dim=str(g["final_dim"]))
    load_state(m, g)
    x, adj = torch.tensor(g["x"]).cuda(), torch.tensor(g["adj"]).cuda()
    a, b = m(x, adj, g["sizes"] if layout == "packed" else None)
    np.testing.assert_allclose(a.detach().cpu().numpy(), g["out_a"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.detach().cpu().numpy(), g["out_b"], rtol=1e-4, atol=1e-5)
    ((a * torch.tensor(g["ga"]).cuda()).sum() + (b * torch.tensor(g["gb"]).cuda()).sum()).backward()
    check_param_grads(m, g, 2e-3, 2e-4)


def assert_grads_arbitrated(named_params, p32, p64, floor=2e-3, handful=8):
    """Gradient check arbitrated by an fp64 run of the oracle.  Per parameter tensor, with mag = max|fp64|:
      * max|hip - fp64| <= max(10 * max|cpu32 - fp64|, floor * mag): the HIP result must be as close to the exact gradient as
        the fp32 CPU oracle is (x10) — the floor only matters for tensors on which fp32 itself is that inexact (piecewise
        ReLU / arg-max winners flipping on summation order);
      * at most a handful of entries (or 4x as many as the CPU fp32 run has) may be further than 1e-4 * mag from fp64, so a
        backward bug that is small against the tensor's largest entry but touches many entries cannot hide under the floor."""
    checked = 0
    for k, p in named_params:
        if p32[k].grad is None or p.grad is None:
            continue
        ref32, ref64 = p32[k].grad.double(), p64[k].grad
        hip = p.grad.detach().cpu().double()
        mag = ref64.abs().max().item()
        cpu_abs, gpu_abs = (ref32 - ref64).abs(), (hip - ref64).abs()
        cpu_err, gpu_err = cpu_abs.max().item(), gpu_abs.max().item()
        assert gpu_err <= max(10 * cpu_err, floor * mag + 1e-9), (k, gpu_err, cpu_err, mag)
        n_cpu = int((cpu_abs > 1e-4 * mag + 1e-12).sum())
        n_gpu = int((gpu_abs > 1e-4 * mag + 1e-12).sum())
        assert n_gpu <= max(handful, 4 * n_cpu), (k, n_gpu, n_cpu, hip.numel(), gpu_err, mag)
        checked += 1
    assert checked > 0



@pytest.mark.parametrize("B,nmax,nbar,fin,hid", [(8, 160, 60, 89, 128), (4, 400, 269, 89, 128), (40, 200, 60, 7, 64),
                                                 (100, 64, 20, 3, 32), (48, 64, 12, 7, 128),
                                                 (48, 64, 20, 89, 128), (100, 96, 30, 40, 128), (64, 256, 39, 3, 128)])
def test_gcn_encoder_vs_oracle_dd_shape(B, nmax, nbar, fin, hid):
    """DD-shaped batches (README.md:39: avg 269 nodes / 676 edges, 89 node labels), 3 layers h=128:
    HIP packed path vs the CPU oracle's dense formulation, outputs and all parameter gradients."""
    from two_stage_gnn_amd import dense_encoders as E
    sizes = dd_like_sizes(7, B, nbar=nbar, nmax=nmax)
    x, adj, sizes = dense_batch(21, B, nmax, fin, sizes=sizes.tolist(), p_edge=2 * 676 / 269 / 269 * (269 / nbar))

    class A:
        bias = True
    torch.manual_seed(0)
    m = E.GcnEncoderGraph(fin, hid, hid, 2, 3, bn=True, args=A(), final_dim="number_classes")
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("bias") and "conv" in k:
                p.copy_(torch.randn_like(p) * 0.2)
    m = m.cuda()
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    a_ref, b_ref = R.gcn_encoder(p_ref, x, adj, bn=True, final_dim="number_classes")
    label = torch.arange(B) % 2
    torch.nn.functional.cross_entropy(b_ref, label).backward()
    a, b = m(x.cuda(), adj.cuda(), sizes)
    torch.testing.assert_close(a.detach().cpu(), a_ref.detach(), rtol=1e-4, atol=1e-4)     # north_star: 1e-4 fp32
    torch.testing.assert_close(b.detach().cpu(), b_ref.detach(), rtol=1e-4, atol=1e-4)
    m.loss(b, label.cuda()).backward()
    # gradients are piecewise (ReLU / max-readout winners can flip on 1-ulp differences between the CPU and GPU summation
    # orders): an fp64 run of the oracle arbitrates
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in p_ref.items()}
    _, b64 = R.gcn_encoder(p64, x.double(), adj.double(), bn=True, final_dim="number_classes")
    torch.nn.functional.cross_entropy(b64, label).backward()
    assert_grads_arbitrated(m.named_parameters(), p_ref, p64)
    for k, p in m.named_parameters():
        ref = p_ref[k].grad
        if ref is not None:
            rel_l2 = ((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12)).item()
            assert rel_l2 < 1e-3, (k, rel_l2)


@pytest.mark.parametrize("shape,B,nmax,layers,hid", [("MUTAG", 32, 40, 2, 64), ("PROTEINS", 64, 620, 3, 128)])
def test_baseline_config_batches_vs_oracle(shape, B, nmax, layers, hid):
    """BASELINE.json configs 1 and 2 on the bench's own synthetic generator (CSR-native ingest, the path bench.py and
    scripts/config_bench.py run): MUTAG 2-layer h=64 batch 32, PROTEINS 3-layer h=128 batch 64 — outputs and gradients vs the
    oracle's dense formulation on the same graphs"""
    from two_stage_gnn_amd import dense_encoders as E, synthetic
    hb = synthetic.host_batch(11, B, shape, nmax)
    g, xrows, label = synthetic.to_device(hb, torch.device("cuda"))
    x, adj = synthetic.to_dense(hb)
    fin = hb["fin"]

    class A:
        bias = True
    torch.manual_seed(0)
    m = E.GcnEncoderGraph(fin, hid, hid, 2, layers, bn=True, args=A(), final_dim="number_classes").cuda()
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    a_ref, b_ref = R.gcn_encoder(p_ref, x, adj, bn=True, final_dim="number_classes")
    lab = torch.from_numpy(hb["label"])
    torch.nn.functional.cross_entropy(b_ref, lab).backward()
    a, b = m(xrows, g)
    torch.testing.assert_close(a.detach().cpu(), a_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.detach().cpu(), b_ref.detach(), rtol=1e-4, atol=1e-4)
    m.loss(b, label).backward()
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in p_ref.items()}
    _, b64 = R.gcn_encoder(p64, x.double(), adj.double(), bn=True, final_dim="number_classes")
    torch.nn.functional.cross_entropy(b64, lab).backward()
    assert_grads_arbitrated(m.named_parameters(), p_ref, p64)


# ----------------------------------------------------------------------------- GAT (encoders_GAT.py)
@pytest.mark.parametrize("tag", ["b1_concat", "b1_raw", "b2_concat"])
def test_gat_head_golden(tag):
    from two_stage_gnn_amd import gat_encoders as G
    g = load_golden("gat_head_" + tag)
    m = G.DGATHead(g["x"].shape[2], g["y"].shape[2], concat=bool(g["concat"]))
    load_state(m, g)
    x = torch.tensor(g["x"]).cuda().requires_grad_(True)
    y = m(x, torch.tensor(g["adj"]).cuda())
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    (y * torch.tensor(g["gy"]).cuda()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx"], rtol=1e-4, atol=2e-5)
    check_param_grads(m, g, 1e-3, 2e-5)


@pytest.mark.parametrize("tag", ["concat_h3", "mean_h2"])
def test_gat_layer_golden(tag):
    from two_stage_gnn_amd import gat_encoders as G
    g = load_golden("gat_layer_" + tag)
    m = G.DGATLayer(g["x"].shape[2], g["p.attention_0.w"].shape[1], n_heads=int(g["heads"]), concat=bool(g["concat"]))
    load_state(m, g)
    x = torch.tensor(g["x"]).cuda().requires_grad_(True)
    y = m(x, torch.tensor(g["adj"]).cuda())
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    (y * torch.tensor(g["gy"]).cuda()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx"], rtol=1e-4, atol=2e-5)
    check_param_grads(m, g, 1e-3, 2e-5)


@pytest.mark.parametrize("tag", ["l2", "l3"])
def test_gat_encoder_golden(tag):
    from two_stage_gnn_amd import gat_encoders as G
    g = load_golden("gat_encoder_" + tag)
    fin, hid, emb, lab = (int(v) for v in g["dims"])
    L = int(g["num_layers"])
    import contextlib, io
    m = G.DGATEncoderGraph(fin, hid, emb, lab, None, num_layers=L, num_heads=[int(h) for h in g["heads"]],
                           neg_input_slopes=[0.2] * L, dropouts=[0.0] * L, final_dim=str(g["final_dim"]))
    load_state(m, g)
    a, b = m(torch.tensor(g["x"]).cuda(), torch.tensor(g["adj"]).cuda(), g["sizes"])
    np.testing.assert_allclose(a.detach().cpu().numpy(), g["out_a"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.detach().cpu().numpy(), g["out_b"], rtol=1e-4, atol=1e-5)
    ((a * torch.tensor(g["ga"]).cuda()).sum() + (b * torch.tensor(g["gb"]).cuda()).sum()).backward()
    for k, p in m.named_parameters():
        ref = g["g." + k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-4, err_msg=k)


def test_gat_dd_graph_vs_oracle():
    """one DD-sized graph (B=1, the reference's GAT batch size, train.py:480), 2 layers x 4 heads x 64"""
    from two_stage_gnn_amd import gat_encoders as G
    x, adj, sizes = dense_batch(5, 1, 320, 89, sizes=[269], p_edge=2 * 676 / 269 / 269)
    torch.manual_seed(1)
    m = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes").cuda()
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    a_ref, b_ref = R.gat_encoder(p_ref, x, adj, final_dim="number_classes")
    a, b = m(x.cuda(), adj.cuda(), sizes)
    torch.testing.assert_close(a.detach().cpu(), a_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.detach().cpu(), b_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.nn.functional.cross_entropy(b_ref, torch.tensor([1])).backward()
    m.loss(b, torch.tensor([1]).cuda()).backward()
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in p_ref.items()}
    _, b64 = R.gat_encoder(p64, x.double(), adj.double(), final_dim="number_classes")
    torch.nn.functional.cross_entropy(b64, torch.tensor([1])).backward()
    assert_grads_arbitrated(m.named_parameters(), p_ref, p64)


@pytest.mark.parametrize("packed", [True, False])
def test_gat_batched_equals_independent_b1_forwards(packed):
    """per_graph_features=True: ONE block-diagonal forward of B DD-sized graphs = B reference forwards at B = 1 (the
    reference's GAT batch size) — outputs per graph, gradients summed over the graphs.  packed: n_b real rows + one ghost
    representative per graph (a graph that fills all Nmax slots and graphs with isolated nodes included); else all padded rows."""
    from two_stage_gnn_amd import gat_encoders as G
    B, nmax = 4, 320
    x, adj, sizes = dense_batch(6, B, nmax, 89, sizes=[269, 120, 320, 33], p_edge=2 * 676 / 269 / 269)
    adj[1, 5, :] = 0; adj[1, :, 5] = 0                                     # an isolated real node (a uniform 1/N column)
    torch.manual_seed(2)
    m = G.DGATEncoderGraph(89, 64, 64, 2, None, num_layers=2, num_heads=[4, 4], final_dim="number_classes",
                           per_graph_features=True).cuda()
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    label = torch.tensor([1, 0, 0, 1])
    refs_a, refs_b, loss_ref = [], [], 0.0
    for b in range(B):
        a_ref, b_ref = R.gat_encoder(p_ref, x[b:b + 1], adj[b:b + 1], final_dim="number_classes")
        refs_a.append(a_ref); refs_b.append(b_ref)
        loss_ref = loss_ref + torch.nn.functional.cross_entropy(b_ref, label[b:b + 1])
    loss_ref.backward()
    a, bb = m(x.cuda(), adj.cuda(), sizes if packed else None)
    torch.testing.assert_close(a.detach().cpu(), torch.cat(refs_a).detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bb.detach().cpu(), torch.cat(refs_b).detach(), rtol=1e-4, atol=1e-4)
    (m.loss(bb, label.cuda()) * B).backward()                              # mean over graphs * B = sum of the B = 1 losses
    for k, p in m.named_parameters():
        ref = p_ref[k].grad
        if ref is None:
            continue
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item() + 1e-7, (k, err, ref.abs().max().item())


# ----------------------------------------------------------------------------- DiffPool (encoders.py:236-406)
def test_diffpool_contract_golden():
    from two_stage_gnn_amd import diffpool as dp
    g = load_golden("diffpool_contract")
    s, z, adj = (torch.tensor(g[k]).cuda().requires_grad_(True) for k in ("s", "z", "adj"))
    xo, ao = dp.diffpool_contract_dense(s, z, adj)
    np.testing.assert_allclose(xo.detach().cpu().numpy(), g["x_out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ao.detach().cpu().numpy(), g["adj_out"], rtol=1e-4, atol=1e-5)
    ((xo * torch.tensor(g["gx"]).cuda()).sum() + (ao * torch.tensor(g["ga"]).cuda()).sum()).backward()
    np.testing.assert_allclose(s.grad.cpu().numpy(), g["gs"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(z.grad.cpu().numpy(), g["gz"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(adj.grad.cpu().numpy(), g["gadj"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,N,K,F,need", [(16, 64, 8, 192, (1, 1, 1)), (3, 50, 7, 36, (1, 1, 1)), (5, 8, 2, 192, (1, 0, 1)),
                                          (2, 100, 30, 64, (0, 1, 0)), (1, 1, 1, 4, (1, 1, 1))])
def test_contract_dense_fused_equals_batched_products(B, N, K, F, need, monkeypatch):
    """the one-workgroup-per-graph contraction (csrc/contract.hip) against the batched-GEMM composition it replaces and fp64:
    both outputs and the gradients of whichever inputs need one"""
    from two_stage_gnn_amd import diffpool as dp
    gen = torch.Generator().manual_seed(B * 1000 + N)
    s0 = torch.softmax(torch.randn(B, N, K, generator=gen), -1)
    z0, a0 = torch.randn(B, N, F, generator=gen), (torch.rand(B, N, N, generator=gen) < 0.2).float()
    gx, ga = torch.randn(B, K, F, generator=gen), torch.randn(B, K, K, generator=gen)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(dp, "FUSED_CONTRACT", fused)
        s, z, a = (t.clone().cuda().requires_grad_(bool(n)) for t, n in zip((s0, z0, a0), need))
        xo, ao = dp.diffpool_contract_dense(s, z, a)
        ((xo * gx.cuda()).sum() + (ao * ga.cuda()).sum()).backward()
        res.append([xo.detach(), ao.detach()] + [t.grad for t in (s, z, a) if t.grad is not None])
    sd, zd, ad = (t.double().requires_grad_(bool(n)) for t, n in zip((s0, z0, a0), need))
    x64 = sd.transpose(1, 2) @ zd
    a64 = sd.transpose(1, 2) @ ad @ sd
    ((x64 * gx.double()).sum() + (a64 * ga.double()).sum()).backward()
    ref = [x64.detach(), a64.detach()] + [t.grad for t in (sd, zd, ad) if t.grad is not None]
    assert len(res[0]) == len(res[1]) == len(ref)
    for f, c, r in zip(res[0], res[1], ref):
        scale = r.abs().max().item() + 1e-12
        assert (f.cpu().double() - r).abs().max().item() <= 2e-5 * scale
        assert (f - c).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("K,F,sizes,nmax", [(64, 192, [96, 33, 1, 64, 70], 96), (8, 36, [5, 40, 17], 40), (24, 4, [31, 32, 33], 40)])
def test_contract_rows_backward_fused_equals_ragged_products(K, F, sizes, nmax, monkeypatch):
    """backward of the row-layout contraction as one launch (csrc/contract.hip) against the four ragged products it replaces
    and against fp64 dense math: ragged graph sizes (slabs of 32 rows with short tails, a one-node graph), ghost rows zero"""
    from two_stage_gnn_amd import diffpool as dp, message_passing as mp
    from two_stage_gnn_amd.graph import GraphBatch
    B = len(sizes)
    x, adj, sizes = dense_batch(43, B, nmax, 3, sizes=sizes, p_edge=0.1)
    gen = torch.Generator().manual_seed(9)
    s0 = torch.softmax(torch.randn(B, nmax, K, generator=gen), -1)
    z0 = torch.randn(B, nmax, F, generator=gen)
    for b, n in enumerate(sizes):
        s0[b, int(n):] = 0; z0[b, int(n):] = 0
    gx, ga = torch.randn(B, K, F, generator=gen), torch.randn(B, K, K, generator=gen)
    g = GraphBatch.from_dense(adj.cuda(), sizes, layout="packed")
    res = []
    for fused in (True, False):
        monkeypatch.setattr(dp, "FUSED_CONTRACT", fused)
        S = mp.pack_rows(s0.cuda(), g).detach().requires_grad_(True)
        Z = mp.pack_rows(z0.cuda(), g).detach().requires_grad_(True)
        xo, ao = dp.diffpool_contract_rows(S, Z, g)
        ((xo * gx.cuda()).sum() + (ao * ga.cuda()).sum()).backward()
        res.append((xo.detach(), ao.detach(), S.grad.clone(), Z.grad.clone()))
    sd, zd = s0.double().requires_grad_(True), z0.double().requires_grad_(True)
    x64 = sd.transpose(1, 2) @ zd
    a64 = sd.transpose(1, 2) @ adj.double() @ sd
    ((x64 * gx.double()).sum() + (a64 * ga.double()).sum()).backward()
    ref = (x64.detach(), a64.detach(), mp.pack_rows(sd.grad.float().cuda(), g).double().cpu(), mp.pack_rows(zd.grad.float().cuda(), g).double().cpu())
    for i, (f, c, r) in enumerate(zip(res[0], res[1], ref)):
        scale = r.abs().max().item() + 1e-12
        fr, cr = f[: g.n_rows] if i >= 2 else f, c[: g.n_rows] if i >= 2 else c
        rr = r[: g.n_rows] if i >= 2 else r
        assert (fr.cpu().double() - rr).abs().max().item() <= 3e-5 * scale, i
        assert (fr - cr).abs().max().item() <= 3e-5 * scale, i
        if i >= 2:
            assert float(f[g.n_rows:].abs().max()) == 0.0 if f.size(0) > g.n_rows else True


def test_diffpool_level0_stack_pairs_equal_separate_stacks(monkeypatch):
    """the two first-level GCN stacks of DiffPool as ONE autograd node with shared launches (sage_stack._SageStackPair,
    csrc/multi.hip) against the two separate nodes: outputs, loss and every parameter gradient; and the launches really are shared"""
    from two_stage_gnn_amd import dense_encoders as E, sage_stack, _native as nat
    B, nmax, fin, hid = 6, 96, 12, 64
    x, adj, sizes = dense_batch(31, B, nmax, fin, sizes=[96, 33, 5, 64, 70, 50], p_edge=0.08)

    class A:
        bias = True
    torch.manual_seed(2)
    m = E.SoftPoolingGcnEncoder(nmax, fin, hid, hid, 2, 3, hid, assign_ratio=0.25, num_pooling=1, bn=True, linkpred=False, args=A(),
                                assign_input_dim=fin, final_dim="number_classes").cuda()
    label = (torch.arange(B) % 2).cuda()
    res = []
    for pair in (True, False):
        monkeypatch.setattr(sage_stack, "PAIR_LAUNCHES", pair)
        m.zero_grad(set_to_none=True)
        nat.trace = []
        try:
            a, b = m(x.cuda(), adj.cuda(), sizes, assign_x=x.cuda())
            loss = m.loss(b, label)
            loss.backward()
            names = [t[0] for t in nat.trace]
        finally:
            nat.trace = None
        res.append((a.detach(), b.detach(), loss.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, names))
    (a1, b1, l1, g1, n1), (a0, b0, l0, g0, n0) = res
    assert n1.count("sage_multi_f32") >= 2 and n0.count("sage_multi_f32") == 0
    assert len(n1) < len(n0)
    torch.testing.assert_close(a1, a0, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(b1, b0, rtol=2e-5, atol=2e-6)
    assert set(g1) == set(g0)
    for k in g0:
        err = (g1[k] - g0[k]).abs().max().item()
        assert err <= 2e-4 * g0[k].abs().max().item() + 1e-8, (k, err)


def test_diffpool_first_step_on_a_batch_equals_the_second():
    """a CSR-native batch builds its neighbour table lazily; built inside one stack's launch record it shifted the two records of
    the paired first-level stacks against each other, so the FIRST step on a batch ran every launch singly (another product
    kernel: slightly different numbers than every later step).  Same launches and bitwise the same loss / gradients now"""
    from two_stage_gnn_amd import dense_encoders as E, synthetic, message_passing as mp, _native as nat

    class A:
        bias = True
    torch.manual_seed(0)
    hb = synthetic.host_batch(4, 6, "DD", 256)
    g, x, lab = synthetic.to_device(hb, torch.device("cuda"))
    m = E.SoftPoolingGcnEncoder(256, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=1, bn=True, linkpred=False, args=A(),
                                assign_input_dim=89, final_dim="number_classes").cuda()
    res = []
    for it in range(3):
        m.zero_grad(set_to_none=True)
        nat.trace = []
        try:
            loss = m.loss(m(x, g, hb["sizes"], assign_x=x)[1], lab)
            loss.backward(gradient=mp.unit_seed(loss.device))
            names = [t[0] for t in nat.trace]
        finally:
            nat.trace = None
        res.append((loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, names))
    mp.check_device_errors()
    assert res[0][2].count("sage_multi_f32") >= 2 and "csr_to_ell" in res[0][2] and "csr_to_ell" not in res[1][2]
    assert res[0][2].count("sage_multi_f32") == res[1][2].count("sage_multi_f32") == res[2][2].count("sage_multi_f32")
    assert res[0][2].count("gather_rowgemm_f32") == res[1][2].count("gather_rowgemm_f32")     # (nothing ran singly the first time)
    for later in res[1:]:
        assert torch.equal(later[0], res[0][0])
        for k, v in res[0][1].items():
            assert torch.equal(later[1][k], v), k


def test_gat_column_softmax_mass_at_baseline_size():
    """attention aggregation on the full DD-shaped 32-graph batch (packed rows + one ghost representative per graph, 4 heads x
    64): every COLUMN j of the (column-)softmax of encoders_GAT.py:41-45 hands out exactly one unit of mass — to its neighbours
    through their softmax weights, or, when all-masked (isolated and padded nodes), as the uniform 1/Nmax over all rows.  With
    features that are a per-row constant c_r, out_i = sum_j alpha_ij c_j, so the multiplicity-weighted row sum of the output must
    equal the multiplicity-weighted sum of c over the graph, for every head and whatever the attention vectors are"""
    import numpy as np
    from two_stage_gnn_amd import attention as att, synthetic
    from two_stage_gnn_amd.graph import GraphBatch
    B, nmax, H, Fo = 32, 1000, 4, 64
    hb = synthetic.host_batch(2, B, "DD", nmax)
    _, adj = synthetic.to_dense(hb)
    g = GraphBatch.from_dense_ghost1(adj.cuda(), hb["sizes"])
    g.transpose_map()
    R = g.total_rows
    gen = torch.Generator(device="cuda").manual_seed(4)
    a_row = torch.randn(H, Fo, generator=gen, device="cuda"); a_col = torch.randn(H, Fo, generator=gen, device="cuda")
    c = torch.rand(R, 1, generator=gen, device="cuda") + 0.5
    h = c.expand(R, H * Fo).contiguous()
    out = att.attention_aggregate(h, a_row, a_col, g, H, 0.2, by_column=True, uniform_isolated=True)      # [R, H*Fo]
    assert out.shape == (R, H * Fo)
    gp = g.graph_ptr.long()
    rg = torch.repeat_interleave(torch.arange(B, device="cuda"), gp[1:] - gp[:-1])
    mult = g.row_mult.double()
    mass = torch.zeros(B, H * Fo, dtype=torch.float64, device="cuda").index_add_(0, rg, out.double()[: g.n_rows] * mult.unsqueeze(1))
    # sum_i mult_i out_i = sum_j mult_j c_j (column j carries c_j, and a unit of softmax mass)
    ref = torch.zeros(B, dtype=torch.float64, device="cuda").index_add_(0, rg, (c.double().view(-1)[: g.n_rows] * mult))
    torch.testing.assert_close(mass, ref.unsqueeze(1).expand(B, H * Fo), rtol=2e-4, atol=1e-3)


def test_diffpool_contraction_properties_at_baseline_size():
    """level-1 contraction X' = S^T Z, A' = S^T A S on the full BASELINE config-5 batch (DD-shaped, 16 graphs, Nmax 512, 64
    clusters, Z 192 wide) through properties the dense oracle would need 16 x 512 x 512 products for: rows of S sum to one
    (ghost rows to zero), so the entries of A'_b add up to the number of edges of graph b and the rows of X'_b to the column sums
    of Z over its nodes; A' is symmetric because A is; both outputs are linear in Z / bilinear in S"""
    from two_stage_gnn_amd import diffpool as dp, synthetic
    hb = synthetic.host_batch(4, 16, "DD", 512)
    g, _, _ = synthetic.to_device(hb, torch.device("cuda"))
    R, K, F = g.total_rows, 64, 192
    gen = torch.Generator(device="cuda").manual_seed(6)
    S = dp.row_softmax(torch.randn(R, K, generator=gen, device="cuda"), g.n_rows if g.n_ghost else None)
    Z = torch.randn(R, F, generator=gen, device="cuda")
    if g.n_ghost:
        Z[g.n_rows:] = 0
    torch.testing.assert_close(S[:g.n_rows].sum(1), torch.ones(g.n_rows, device="cuda"), rtol=1e-5, atol=1e-5)
    assert not g.n_ghost or float(S[g.n_rows:].abs().max()) == 0.0
    Xo, Ao = dp.diffpool_contract_rows(S, Z, g)
    assert Xo.shape == (g.B, K, F) and Ao.shape == (g.B, K, K)
    gp = g.graph_ptr.long()
    deg = (g.rowptr[1:] - g.rowptr[:-1]).double()[:g.n_rows]
    rg = torch.repeat_interleave(torch.arange(g.B, device="cuda"), gp[1:] - gp[:-1])
    edges = torch.zeros(g.B, dtype=torch.float64, device="cuda").index_add_(0, rg, deg)
    torch.testing.assert_close(Ao.double().sum((1, 2)), edges, rtol=1e-4, atol=1e-3)
    zsum = torch.zeros(g.B, F, dtype=torch.float64, device="cuda").index_add_(0, rg, Z[:g.n_rows].double())
    torch.testing.assert_close(Xo.double().sum(1), zsum, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(Ao, Ao.transpose(1, 2), rtol=1e-4, atol=1e-4)
    Z2 = torch.randn(R, F, generator=gen, device="cuda")
    if g.n_ghost:
        Z2[g.n_rows:] = 0
    X2, _ = dp.diffpool_contract_rows(S, Z2, g)
    X3, _ = dp.diffpool_contract_rows(S, 2.0 * Z - 0.5 * Z2, g)
    torch.testing.assert_close(X3, 2.0 * Xo - 0.5 * X2, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["p1", "p2", "p1_nomask"])
def test_diffpool_encoder_golden(tag):
    from two_stage_gnn_amd import dense_encoders as E
    g = load_golden("diffpool_" + tag)
    nmax, fin, hid, emb, lab, L, npool = (int(v) for v in g["cfg"])

    class A:
        bias = True
    m = E.SoftPoolingGcnEncoder(nmax, fin, hid, emb, lab, L, hid, assign_ratio=float(g["ratio"]), num_pooling=npool,
                                bn=True, linkpred=False, args=A(), assign_input_dim=fin, final_dim=str(g["final_dim"]))
    load_state(m, g)
    x, adj = torch.tensor(g["x"]).cuda(), torch.tensor(g["adj"]).cuda()
    bnn = g["sizes"] if int(g["masked"]) else None
    a, b = m(x, adj, bnn, assign_x=x)
    np.testing.assert_allclose(a.detach().cpu().numpy(), g["out_a"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.detach().cpu().numpy(), g["out_b"], rtol=1e-4, atol=1e-5)
    ((a * torch.tensor(g["ga"]).cuda()).sum() + (b * torch.tensor(g["gb"]).cuda()).sum()).backward()
    # gradients: arbitrated by an fp64 run of the oracle on the fixture's inputs.  The reference's own fp32
    # gradients (the fixture) play the part of the fp32 CPU run: per tensor the HIP result must be as close to the exact gradient
    # as the reference's fp32 is (x10), with the 2e-3 floor of assert_grads_arbitrated for tensors fp32 itself cannot resolve
    # (pooled-level biases in front of normalise + BN sum thousands of cancelling terms; arg-max winners flip on a last bit), and
    # only a handful of entries may be further than 1e-4 of the tensor's scale from fp64.
    p64 = {k[2:]: torch.tensor(v).double().requires_grad_(True) for k, v in g.items() if k.startswith("p.")}
    a64, b64 = R.diffpool_encoder(p64, torch.tensor(g["x"]).double(), torch.tensor(g["adj"]).double(), bnn, npool,
                                  assign_x=torch.tensor(g["x"]).double(), final_dim=str(g["final_dim"]))
    ((a64 * torch.tensor(g["ga"]).double()).sum() + (b64 * torch.tensor(g["gb"]).double()).sum()).backward()

    class _G:                                              # (what assert_grads_arbitrated reads: `.grad`)
        def __init__(self, t):
            self.grad = t
    p32 = {k: _G(torch.tensor(g["g." + k])) for k, _ in m.named_parameters() if "g." + k in g}
    for k, v in p64.items():
        if v.grad is None:
            v.grad = torch.zeros_like(v)
    assert_grads_arbitrated([(k, p) for k, p in m.named_parameters() if k in p32], p32, p64)


@pytest.mark.parametrize("tag", ["masked", "nomask", "masked2", "nomask2"])
def test_diffpool_linkpred_golden(tag):
    """f4 against the REFERENCE's own run (tests/golden/diffpool_linkpred_*.npz, oracle/gen_golden.py linkpred): the loss with the
    link-prediction term of encoders.py:409-441, `model.linkpred_clamp` set to the value the reference's uninitialised clamp tensor
    held when the fixture was made; loss, link loss and every parameter gradient (fp64-arbitrated like the encoder fixtures)"""
    from two_stage_gnn_amd import dense_encoders as E
    g = load_golden("diffpool_linkpred_" + tag)
    nmax, fin, hid, emb, lab, L, npool = (int(v) for v in g["cfg"])

    class A:
        bias = True
    m = E.SoftPoolingGcnEncoder(nmax, fin, hid, emb, lab, L, hid, assign_ratio=float(g["ratio"]), num_pooling=npool,
                                bn=True, linkpred=True, args=A(), assign_input_dim=fin, final_dim="number_classes")
    load_state(m, g)
    m.linkpred_clamp = float(g["clamp"])
    x, adj = torch.tensor(g["x"]).cuda(), torch.tensor(g["adj"]).cuda()
    bnn = g["sizes"] if int(g["masked"]) else None
    _, ypred = m(x, adj, bnn, assign_x=x)
    np.testing.assert_allclose(ypred.detach().cpu().numpy(), g["ypred"], rtol=1e-4, atol=1e-5)
    loss = m.loss(ypred, torch.tensor(g["label"]).cuda(), adj, bnn)
    np.testing.assert_allclose(float(m.link_loss), float(g["link_loss"]), rtol=1e-4)
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-4)
    loss.backward()
    p64 = {k[2:]: torch.tensor(v).double().requires_grad_(True) for k, v in g.items() if k.startswith("p.")}
    _, y64, s64 = R.diffpool_encoder(p64, torch.tensor(g["x"]).double(), torch.tensor(g["adj"]).double(), bnn, npool,
                                     assign_x=torch.tensor(g["x"]).double(), final_dim="number_classes", return_assign=True)
    (torch.nn.functional.cross_entropy(y64, torch.tensor(g["label"])) +
     R.diffpool_link_loss(s64, torch.tensor(g["adj"]).double(), bnn, clamp=float(g["clamp"]))).backward()

    class _G:
        def __init__(self, t):
            self.grad = t
    p32 = {k: _G(torch.tensor(g["g." + k])) for k, _ in m.named_parameters() if "g." + k in g}
    for k, v in p64.items():
        if v.grad is None:
            v.grad = torch.zeros_like(v)
    assert_grads_arbitrated([(k, p) for k, p in m.named_parameters() if k in p32], p32, p64)


def test_diffpool_dd_config_vs_oracle():
    """BASELINE config 5 shape (scaled down 4x in batch): Nmax=512 -> 64 -> 8, h=64, 3 layers, masked."""
    from two_stage_gnn_amd import dense_encoders as E
    B, nmax, fin, hid = 4, 512, 89, 64
    sizes = dd_like_sizes(3, B, nbar=269, nmax=nmax)
    x, adj, sizes = dense_batch(33, B, nmax, fin, sizes=sizes.tolist(), p_edge=2 * 676 / 269 / 269)

    class A:
        bias = True
    torch.manual_seed(2)
    m = E.SoftPoolingGcnEncoder(nmax, fin, hid, hid, 2, 3, hid, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False,
                                args=A(), assign_input_dim=fin, final_dim="number_classes")
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    a_ref, b_ref = R.diffpool_encoder(p_ref, x, adj, sizes, 2, assign_x=x, final_dim="number_classes")
    a, b = m(x.cuda(), adj.cuda(), sizes, assign_x=x.cuda())
    torch.testing.assert_close(a.detach().cpu(), a_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.detach().cpu(), b_ref.detach(), rtol=1e-4, atol=1e-4)
    label = torch.arange(B) % 2
    torch.nn.functional.cross_entropy(b_ref, label).backward()
    m.loss(b, label.cuda()).backward()
    # Several gradients here are ill-conditioned in fp32 (biases in front of normalize+BN at the pooled levels sum
    # thousands of cancelling terms; conv_last.bias carries F.normalize's 1/eps = 1e12 clamp factor on the ghost
    # rows, exactly as the reference does).  Judge the HIP result against an fp64 run of the oracle: it must be
    # as close to fp64 as the fp32 CPU oracle is (x10), or within 2e-3 of the tensor's scale.
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in p_ref.items()}
    _, b64 = R.diffpool_encoder(p64, x.double(), adj.double(), sizes, 2, assign_x=x.double(), final_dim="number_classes")
    torch.nn.functional.cross_entropy(b64, label).backward()
    for k, p in m.named_parameters():
        ref32, ref64 = p_ref[k].grad, p64[k].grad
        if ref32 is None or p.grad is None:
            continue
        cpu_err = (ref32.double() - ref64).abs().max().item()
        gpu_err = (p.grad.cpu().double() - ref64).abs().max().item()
        mag = ref64.abs().max().item()
        assert gpu_err <= max(10 * cpu_err, 2e-3 * mag + 1e-9), (k, gpu_err, cpu_err, mag)


@pytest.mark.parametrize("B,K,fin,hid,last", [(16, 64, 192, 64, 64), (16, 8, 192, 64, 8), (5, 20, 12, 32, 16)])
def test_dense_gcn_stack_equals_composed_layers(B, K, fin, hid, last, monkeypatch):
    """a pooled DiffPool level's GCN stack as one autograd node (layers write into the concatenation, adjacency gradient
    accumulated by the batched products) against the per-layer composition: outputs, dx, dA and every parameter gradient"""
    from two_stage_gnn_amd import dense_encoders as E

    class A:
        bias = True
    torch.manual_seed(11)
    m = E.SoftPoolingGcnEncoder(64, 12, hid, last, 2, 3, hid, assign_ratio=0.25, num_pooling=1, bn=True, linkpred=False, args=A(),
                                assign_input_dim=12, final_dim="number_classes")
    c1, cb, cl = m.build_conv_layers(fin, hid, last, 3, False, normalize=True, dropout=0.0)
    cb = torch.nn.ModuleList(cb)
    for mod in [c1, cl] + list(cb):
        mod.cuda()
        torch.nn.init.normal_(mod.bias.data, std=0.1)
    gen = torch.Generator().manual_seed(12)
    x0 = torch.randn(B, K, fin, generator=gen).cuda()
    a0 = torch.rand(B, K, K, generator=gen).cuda()
    gy = torch.randn(B, K, 2 * hid + last, generator=gen).cuda()
    from two_stage_gnn_amd import dense_stack, message_passing as mp
    res = []
    for one_launch, fused in ((True, True), (False, True), (False, False)):      # one launch per direction / one node / composed
        monkeypatch.setattr(dense_stack, "ONE_LAUNCH", one_launch)
        monkeypatch.setattr(E, "FUSED_DENSE_STACK", fused)
        x, a = x0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
        for mod in [c1, cl] + list(cb):
            mod.zero_grad(set_to_none=True)
        y, _ = m.gcn_forward_dense(x, a, c1, cb, cl)
        (y * gy).sum().backward()
        res.append([y.detach(), x.grad, a.grad] + [p.grad.clone() for mod in [c1] + list(cb) + [cl] for p in mod.parameters()])
    mp.check_device_errors()
    assert res[0][0].shape == (B, K, 2 * hid + last)
    for other in (res[0], res[1]):
        for f, c in zip(other, res[2]):
            scale = c.abs().max().item() + 1e-12
            assert (f - c).abs().max().item() <= 2e-4 * scale


@pytest.mark.parametrize("B,K,fin,hid,lasts,need_x", [(16, 64, 192, 64, (64, 8), True), (16, 8, 192, 64, (64, 4), True),
                                                       (7, 20, 12, 32, (16, 8), False)])
def test_two_dense_stacks_in_one_launch(B, K, fin, hid, lasts, need_x):
    """the embedding stack of a pooled level and the next level's assignment stack (same x and adjacency) in ONE launch per
    direction == the two stacks run one after the other: outputs, dx (summed over the stacks), dA (summed), all parameter grads"""
    from two_stage_gnn_amd import dense_encoders as E, dense_stack, message_passing as mp

    class A:
        bias = True
    torch.manual_seed(21)
    m = E.SoftPoolingGcnEncoder(64, 12, hid, 16, 2, 3, hid, assign_ratio=0.25, num_pooling=1, bn=True, linkpred=False, args=A(),
                                assign_input_dim=12, final_dim="number_classes")
    stacks = []
    for last in lasts:
        c1, cb, cl = m.build_conv_layers(fin, hid, last, 3, False, normalize=True, dropout=0.0)
        convs = [c1] + list(cb) + [cl]
        for mod in convs:
            mod.cuda()
            torch.nn.init.normal_(mod.bias.data, std=0.1)
        stacks.append(convs)
    gen = torch.Generator().manual_seed(22)
    x0 = torch.randn(B, K, fin, generator=gen).cuda()
    a0 = torch.rand(B, K, K, generator=gen).cuda()
    gys = [torch.randn(B, K, 2 * hid + last, generator=gen).cuda() for last in lasts]
    assert dense_stack.one_launch_ok(x0, a0, stacks)
    res = []
    for together in (True, False):
        x, a = x0.clone().requires_grad_(need_x), a0.clone().requires_grad_(True)
        for convs in stacks:
            for mod in convs:
                mod.zero_grad(set_to_none=True)
        ys = dense_stack.dense_gcn_stacks(x, a, stacks) if together else [dense_stack.dense_gcn_stacks(x, a, [c])[0] for c in stacks]
        sum((y * gy).sum() for y, gy in zip(ys, gys)).backward()
        res.append([y.detach() for y in ys] + ([x.grad] if need_x else []) + [a.grad]
                   + [p.grad.clone() for convs in stacks for mod in convs for p in mod.parameters()])
    mp.check_device_errors()
    for f, c in zip(res[0], res[1]):
        scale = c.abs().max().item() + 1e-12
        assert (f - c).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("nstack", [1, 2])
def test_dense_stacks_adjacency_pass_through(nstack):
    """adj_pass: the node also hands the adjacency on to its other consumer; that consumer's gradient is summed into dA inside
    the backward launch — same dA, dx and parameter gradients as autograd's own sum of the two consumers' gradients"""
    from two_stage_gnn_amd import dense_encoders as E, dense_stack, message_passing as mp
    B, K, fin, hid = 6, 32, 64, 32
    torch.manual_seed(31)
    stacks = [[E.GraphConv(fin, hid, normalize_embedding=True, bias=True), E.GraphConv(hid, hid, normalize_embedding=True, bias=True),
               E.GraphConv(hid, 16, normalize_embedding=True, bias=True)] for _ in range(nstack)]
    for convs in stacks:
        for mod in convs:
            mod.cuda()
            torch.nn.init.xavier_uniform_(mod.weight.data)
            torch.nn.init.normal_(mod.bias.data, std=0.1)
    gen = torch.Generator().manual_seed(23)
    x0 = torch.randn(B, K, fin, generator=gen).cuda()
    a0 = torch.rand(B, K, K, generator=gen).cuda()
    gys = [torch.randn(B, K, 2 * hid + 16, generator=gen).cuda() for _ in range(nstack)]
    ga = torch.randn(B, K, K, generator=gen).cuda()
    assert dense_stack.one_launch_ok(x0, a0, stacks)
    res = []
    for passing in (True, False):
        x, a = x0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
        for convs in stacks:
            for mod in convs:
                mod.zero_grad(set_to_none=True)
        out = dense_stack.dense_gcn_stacks(x, a, stacks, adj_pass=passing)
        ys, a_other = (out[:-1], out[-1]) if passing else (out, a)
        assert len(ys) == nstack
        if passing:
            assert a_other.data_ptr() == a.data_ptr()
        (sum((y * gy).sum() for y, gy in zip(ys, gys)) + (a_other * a_other * ga).sum()).backward()
        res.append([y.detach() for y in ys] + [x.grad, a.grad] + [p.grad.clone() for convs in stacks for mod in convs for p in mod.parameters()])
    mp.check_device_errors()
    for f, c in zip(res[0], res[1]):
        scale = c.abs().max().item() + 1e-12
        assert (f - c).abs().max().item() <= 2e-6 * scale
    # the pass-through output left unused: dA is the stacks' own
    x, a = x0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
    out = dense_stack.dense_gcn_stacks(x, a, stacks, adj_pass=True)
    sum((y * gy).sum() for y, gy in zip(out[:-1], gys)).backward()
    x2, a2 = x0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
    sum((y * gy).sum() for y, gy in zip(dense_stack.dense_gcn_stacks(x2, a2, stacks), gys)).backward()
    assert torch.equal(a.grad, a2.grad) and torch.equal(x.grad, x2.grad)


@pytest.mark.parametrize("masked", [True, False])
@pytest.mark.parametrize("sym", [True, False])
def test_link_pred_loss_vs_oracle(masked, sym):
    """f4: the link-prediction side loss (encoders.py:416-438) on packed assignment rows + CSR against the dense
    restatement — value and d loss / d S; directed and weighted adjacencies, ragged graph sizes, clamp below 1"""
    from two_stage_gnn_amd import diffpool as dp
    from two_stage_gnn_amd import message_passing as mp
    from two_stage_gnn_amd.graph import GraphBatch
    B, nmax, K = 5, 96, 24
    x, adj, sizes = dense_batch(41, B, nmax, 3, sizes=[96, 33, 1, 64, 70], p_edge=0.08)
    gen = torch.Generator().manual_seed(7)
    if not sym:
        adj = adj * (torch.rand(adj.shape, generator=gen) < 0.6).float() * (0.5 + torch.rand(adj.shape, generator=gen))
    s = torch.softmax(torch.randn(B, nmax, K, generator=gen) * 2, dim=-1)
    if masked:
        for b, n in enumerate(sizes):
            s[b, int(n):] = 0
    for clamp in (1.0, 0.3):
        sr = s.clone().requires_grad_(True)
        ref = R.diffpool_link_loss(sr, adj, sizes if masked else None, clamp=clamp)
        ref.backward()
        g = GraphBatch.from_dense(adj.cuda(), sizes if masked else None, layout="packed" if masked else "padded",
                                  assume_symmetric=sym)
        sp = (mp.pack_rows(s.cuda(), g) if masked else s.cuda().reshape(B * nmax, K)).detach().requires_grad_(True)
        got = dp.link_pred_loss(sp, g, clamp=clamp, masked=masked)
        torch.testing.assert_close(got.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-6)
        (got * 3.0).backward()
        gs = mp.unpack_rows(sp.grad, g).cpu() if masked else sp.grad.reshape(B, nmax, K).cpu()
        if masked:
            for b, n in enumerate(sizes):
                gs[b, int(n):] = 0                      # padded rows: not part of the packed problem
                sr.grad[b, int(n):] = 0
        torch.testing.assert_close(gs, 3.0 * sr.grad, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("hop", [2, 3])
@pytest.mark.parametrize("masked,sym", [(True, True), (True, False), (False, True)])
def test_link_pred_loss_adj_hop_vs_oracle(hop, masked, sym):
    """adj_hop > 1 (encoders.py:419-423: pred = sum_p (S S^T)^p, clamped): value and d loss / d S against the dense restatement,
    fp32 and fp64 (the gradient runs through the K x K polynomial of S^T S)"""
    from two_stage_gnn_amd import diffpool as dp
    from two_stage_gnn_amd import message_passing as mp
    from two_stage_gnn_amd.graph import GraphBatch
    B, nmax, K = 5, 40, 24
    x, adj, sizes = dense_batch(45, B, nmax, 3, sizes=[40, 13, 1, 33, 20], p_edge=0.15)
    gen = torch.Generator().manual_seed(11)
    if not sym:
        adj = adj * (torch.rand(adj.shape, generator=gen) < 0.6).float() * (0.5 + torch.rand(adj.shape, generator=gen))
    s = torch.softmax(torch.randn(B, nmax, K, generator=gen) * 1.5, dim=-1) * 0.6          # (scaled: part of the entries stays below the clamp)
    if masked:
        for b, n in enumerate(sizes):
            s[b, int(n):] = 0
    sr = s.clone().double().requires_grad_(True)
    ref = R.diffpool_link_loss(sr, adj.double(), sizes if masked else None, clamp=1.0, adj_hop=hop)
    ref.backward()
    frac_clamped = float(((sr.detach() @ sr.detach().transpose(1, 2)) >= 1).float().mean())
    assert frac_clamped < 0.9
    g = GraphBatch.from_dense(adj.cuda(), sizes if masked else None, layout="packed" if masked else "padded", assume_symmetric=sym)
    sp = (mp.pack_rows(s.cuda(), g) if masked else s.cuda().reshape(B * nmax, K)).detach().requires_grad_(True)
    got = dp.link_pred_loss(sp, g, clamp=1.0, masked=masked, adj_hop=hop)
    torch.testing.assert_close(got.detach().cpu().double(), ref.detach(), rtol=2e-4, atol=1e-6)
    (got * 3.0).backward()
    gs = mp.unpack_rows(sp.grad, g).cpu() if masked else sp.grad.reshape(B, nmax, K).cpu()
    gr = sr.grad.clone()
    if masked:
        for b, n in enumerate(sizes):
            gs[b, int(n):] = 0
            gr[b, int(n):] = 0
    err = (gs.double() - 3.0 * gr).abs().max().item()
    assert err <= 2e-3 * (3.0 * gr).abs().max().item() + 1e-7, (err, gr.abs().max().item())


def test_diffpool_linkpred_encoder_vs_oracle():
    """SoftPoolingGcnEncoder(linkpred=True, num_pooling=1): CE + link loss and all parameter gradients vs the oracle;
    num_pooling=2 reproduces the reference's failure (trap T7)"""
    from two_stage_gnn_amd import dense_encoders as E
    B, nmax, fin, hid = 4, 128, 12, 32
    sizes = dd_like_sizes(5, B, nbar=70, nmax=nmax)
    x, adj, sizes = dense_batch(35, B, nmax, fin, sizes=sizes.tolist(), p_edge=0.06)

    class A:
        bias = True
    torch.manual_seed(4)
    m = E.SoftPoolingGcnEncoder(nmax, fin, hid, hid, 2, 3, hid, assign_ratio=0.25, num_pooling=1, bn=True, linkpred=True,
                                args=A(), assign_input_dim=fin, final_dim="number_classes")
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    _, b_ref, s_ref = R.diffpool_encoder(p_ref, x, adj, sizes, 1, assign_x=x, final_dim="number_classes", return_assign=True)
    label = torch.arange(B) % 2
    link_ref = R.diffpool_link_loss(s_ref, adj, sizes)
    loss_ref = torch.nn.functional.cross_entropy(b_ref, label) + link_ref
    loss_ref.backward()
    _, b = m(x.cuda(), adj.cuda(), sizes, assign_x=x.cuda())
    loss = m.loss(b, label.cuda(), adj.cuda(), sizes)
    torch.testing.assert_close(m.link_loss.detach().cpu(), link_ref.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(loss.detach().cpu(), loss_ref.detach(), rtol=1e-4, atol=1e-5)
    loss.backward()
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in p_ref.items()}
    _, b64, s64 = R.diffpool_encoder(p64, x.double(), adj.double(), sizes, 1, assign_x=x.double(), final_dim="number_classes",
                                     return_assign=True)
    (torch.nn.functional.cross_entropy(b64, label) + R.diffpool_link_loss(s64, adj.double(), sizes)).backward()
    for k, p in m.named_parameters():
        ref32, ref64 = p_ref[k].grad, p64[k].grad
        if ref32 is None or p.grad is None:
            continue
        cpu_err = (ref32.double() - ref64).abs().max().item()
        gpu_err = (p.grad.cpu().double() - ref64).abs().max().item()
        mag = ref64.abs().max().item()
        assert gpu_err <= max(10 * cpu_err, 2e-3 * mag + 1e-9), (k, gpu_err, cpu_err, mag)
    m2 = E.SoftPoolingGcnEncoder(nmax, fin, hid, hid, 2, 3, hid, assign_ratio=0.25, num_pooling=2, bn=True, linkpred=True,
                                 args=A(), assign_input_dim=fin, final_dim="number_classes")
    _, b2 = m2(x.cuda(), adj.cuda(), sizes, assign_x=x.cuda())
    with pytest.raises(RuntimeError, match="num_pooling"):
        m2.loss(b2, label.cuda(), adj.cuda(), sizes)


# ----------------------------------------------------------------------------- triplet step (tripletnet.py)
@pytest.mark.parametrize("hidden,nmax,sizes,final", [(128, 40, [40, 17, 29], "output_dim"), (128, 64, [33, 64, 5], "pretrain"),
                                                     (64, 48, [20, 31, 9], "output_dim")])
def test_triplet_fused_stack_with_per_graph_statistics(hidden, nmax, sizes, final):
    """the triplet step on the fused conv stack (sage_stack.per_graph_stats: the slot batch-norm launches replaced by their
    row-local counterparts, tsgnn_row_ln_fwd_f32 / tsgnn_row_post_bwd_f32) against the oracle's three B = 1 forwards — embeddings,
    distances and every parameter gradient — and against the per-op path (TSGNN_PER_GRAPH_STACK=0); graphs that fill all Nmax
    slots, padded winners of the max readout (ghost rows) included"""
    from two_stage_gnn_amd import dense_encoders as E, _native as nat
    from two_stage_gnn_amd.triplet import tripletnet
    fin = 12
    x, adj, sz = dense_batch(57 + hidden, 3, nmax, fin, sizes=sizes, p_edge=0.15)

    class A:
        bias = True
    torch.manual_seed(6)
    m = E.GcnEncoderGraph(fin, hidden, hidden, 2, 3, bn=True, args=A(), final_dim=final)      # "pretrain" = 2stg+ (embedding second too)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "conv" in k and k.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.3)           # padded rows then carry values that can win the max readout
    m = m.cuda()
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    embeds = [R.gcn_encoder(p_ref, x[b:b + 1], adj[b:b + 1], bn=True, final_dim=final)[1] for b in range(3)]
    dp_ref = torch.nn.functional.pairwise_distance(embeds[0], embeds[1], 2)
    dn_ref = torch.nn.functional.pairwise_distance(embeds[0], embeds[2], 2)
    (torch.nn.MarginRankingLoss(margin=10.0)(dp_ref, dn_ref, torch.tensor([-1.0])) + 0.05 * embeds[1].norm(2)).backward()
    net = tripletnet(m)
    gs = [_G(adj[b].numpy(), x[b].numpy(), int(sz[b])) for b in range(3)]

    def run():
        m.zero_grad(set_to_none=True)
        nat.trace = []
        dp, dn, ea, ep, en = net(*gs)
        (torch.nn.MarginRankingLoss(margin=10.0)(dp, dn, torch.tensor([-1.0]).cuda()) + 0.05 * ep.norm(2)).backward()
        names = [t[0] for t in nat.trace]
        nat.trace = None
        return dp.detach(), dn.detach(), ea.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, names

    dp, dn, ea, grads, names = run()
    assert "row_post_bwd_f32" in names and "row_ln_fwd_f32" in names and "ell_spmm_f32" not in names, names
    torch.testing.assert_close(dp.cpu(), dp_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dn.cpu(), dn_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ea.cpu(), embeds[0].detach(), rtol=1e-4, atol=1e-4)
    for k, gr in grads.items():
        ref = p_ref[k].grad
        if ref is None:
            continue
        err = (gr.cpu() - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item() + 1e-6, (k, err, ref.abs().max().item())
    E.PER_GRAPH_STACK = False
    try:
        dp2, dn2, ea2, grads2, names2 = run()
    finally:
        E.PER_GRAPH_STACK = True
    assert "row_post_bwd_f32" not in names2
    torch.testing.assert_close(dp2, dp, rtol=1e-5, atol=1e-5)
    for k in grads:
        torch.testing.assert_close(grads2[k], grads[k], rtol=1e-3, atol=1e-5 + 1e-4 * float(grads[k].abs().max()))


def test_triplet_step_under_flat_trainer_equals_torch_adam():
    """the triplet step as bench.py --triplet replays it (FlatTrainer: every fused backward node writes its gradients straight into the
    flat bucket — the triplet tail's dW / db included —, clip 2.0 + Adam in the library's kernels, one hipGraph) against the same
    model stepped by autograd + clip_grad_norm_ + torch.optim.Adam: parameters after three steps"""
    import copy
    from two_stage_gnn_amd import dense_encoders as E
    from two_stage_gnn_amd.triplet import tripletnet, MarginRankingLoss
    from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep
    from two_stage_gnn_amd.graph import GraphBatch
    nmax, fin, hidden = 48, 12, 128
    x, adj, sz = dense_batch(91, 3, nmax, fin, sizes=[48, 21, 30], p_edge=0.15)

    class A:
        bias = True
    torch.manual_seed(8)
    m1 = E.GcnEncoderGraph(fin, hidden, hidden, 2, 3, bn=True, args=A(), final_dim="output_dim").cuda()
    m2 = copy.deepcopy(m1)
    g = GraphBatch.from_dense(adj.cuda(), sizes=sz, layout="packed", assume_symmetric=True)
    g.val = None                                                   # unit weights (what tripletnet's resident pieces carry)
    from two_stage_gnn_amd import message_passing as mp
    xr = mp.pack_rows(x.cuda(), g, 12)
    tgt = torch.tensor([-1.0]).cuda()
    # (a) autograd + torch's optimiser
    net2, crit2 = tripletnet(m2), torch.nn.MarginRankingLoss(margin=10.0)
    params2 = [p for p in m2.parameters()]
    opt = torch.optim.Adam(params2, lr=1e-3)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        dp, dn = net2._embed(xr, g, sz, xr)[:2]
        crit2(dp, dn, tgt).backward()
        torch.nn.utils.clip_grad_norm_([p for p in params2 if p.grad is not None], 2.0)
        opt.step()
    # (b) the flat trainer, one hipGraph per step
    net1, crit1 = tripletnet(m1), MarginRankingLoss(margin=10.0)
    tr = FlatTrainer(m1, lr=1e-3, clip=2.0)
    gs = GraphedStep(tr, lambda: crit1(*net1._embed(xr, g, sz, xr)[:2], tgt), warmup=3)      # (warm-up steps are rolled back)
    for _ in range(3):
        gs.step()
    gs.loss_value()
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p2.grad is None:
            continue
        torch.testing.assert_close(p1.detach(), p2.detach(), rtol=2e-4, atol=2e-6, msg=lambda s_, k=k: k + ": " + s_)


@pytest.mark.parametrize("D,E,bias", [(384, 128, True), (20, 7, True), (48, 16, False)])
def test_triplet_tail_kernels(D, E, bias):
    """embeddings + both pairwise distances in one launch and their backward in one launch (csrc/triplet.hip) against torch's
    Linear + F.pairwise_distance, with gradient reaching the embeddings directly as well (the norm regularisers of
    train_triplet.py:262-263) and with one of the two distances unused"""
    from two_stage_gnn_amd import triplet as T3
    gen = torch.Generator().manual_seed(D + E)
    r = torch.randn(3, D, generator=gen)
    lin = torch.nn.Linear(D, E, bias=bias)
    for used in ("both", "dp_only"):
        rr = r.clone().requires_grad_(True)
        e = lin(rr)
        dp = torch.nn.functional.pairwise_distance(e[0:1], e[1:2], 2)
        dn = torch.nn.functional.pairwise_distance(e[0:1], e[2:3], 2)
        loss = torch.nn.MarginRankingLoss(margin=5.0)(dp, dn, torch.tensor([-1.0])) + 0.1 * (e[0:1].norm(2) + e[2:3].norm(1)) if used == "both" \
            else (dp * 1.5).sum()
        g_ref = torch.autograd.grad(loss, [rr, lin.weight] + ([lin.bias] if bias else []))
        rg = r.cuda().requires_grad_(True)
        w = lin.weight.detach().cuda().requires_grad_(True)
        b = lin.bias.detach().cuda().requires_grad_(True) if bias else None
        dpg, dng, ea, ep, en = T3._TripletTail.apply(rg, w, b)
        torch.testing.assert_close(torch.cat([ea, ep, en]).detach().cpu(), e.detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dpg.detach().cpu(), dp.detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dng.detach().cpu(), dn.detach(), rtol=1e-5, atol=1e-5)
        lg = torch.nn.MarginRankingLoss(margin=5.0)(dpg, dng, torch.tensor([-1.0]).cuda()) + 0.1 * (ea.norm(2) + en.norm(1)) if used == "both" \
            else (dpg * 1.5).sum()
        g = torch.autograd.grad(lg, [rg, w] + ([b] if bias else []))
        for a, c in zip(g, g_ref):
            torch.testing.assert_close(a.cpu(), c, rtol=1e-4, atol=1e-5)


def test_triplet_with_the_gat_encoder_keeps_the_dense_inputs():
    """tripletnet on a DGATEncoderGraph (tripletnet.py is encoder-agnostic): the GAT encoder packs its batch itself, so the module hands
    it the dense tensors as the reference does; distances equal those of three separate B = 1 calls of the same module"""
    from two_stage_gnn_amd import gat_encoders as G
    from two_stage_gnn_amd.triplet import tripletnet
    nmax, fin = 20, 8
    x, adj, sizes = dense_batch(71, 3, nmax, fin, sizes=[20, 9, 14], p_edge=0.25)
    torch.manual_seed(3)
    m = G.DGATEncoderGraph(fin, 8, 8, 2, None, num_layers=2, num_heads=[2, 2], final_dim="output_dim", per_graph_features=True).cuda()
    net = tripletnet(m)
    gs = [_G(adj[b].numpy(), x[b].numpy(), int(sizes[b])) for b in range(3)]
    dp, dn, ea, ep, en = net(*gs)
    assert len(net._resident) == 0                                   # the resident CSR pieces are for the GraphSage / DiffPool family
    e = [m(x[b:b + 1].cuda(), adj[b:b + 1].cuda(), sizes[b:b + 1])[1] for b in range(3)]
    torch.testing.assert_close(ea, e[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dp, torch.nn.functional.pairwise_distance(e[0], e[1], 2), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dn, torch.nn.functional.pairwise_distance(e[0], e[2], 2), rtol=1e-4, atol=1e-5)
    (dp - dn).sum().backward()
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert any("attention" in k or "conv" in k for k in got) and any(k.startswith("map_model") for k in got)
    assert all(torch.isfinite(gr).all() for gr in got.values())      # (pred_model is not on this output's path: no gradient)


@pytest.mark.parametrize("n,margin,reduction", [(1, 1.0, "mean"), (7, 0.3, "mean"), (300, 0.0, "sum")])
def test_margin_ranking_loss_drop_in(n, margin, reduction):
    """triplet.MarginRankingLoss == torch.nn.MarginRankingLoss (train_triplet.py:235,277): value and both input gradients, mixed
    targets, hinge on and off"""
    from two_stage_gnn_amd.triplet import MarginRankingLoss
    gen = torch.Generator().manual_seed(n)
    x1, x2 = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    t = torch.where(torch.rand(n, generator=gen) < 0.5, -torch.ones(n), torch.ones(n)) if n > 1 else torch.tensor([-1.0])
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    ref = torch.nn.MarginRankingLoss(margin=margin, reduction=reduction)(a, b, t)
    (ref * 1.7).backward()
    ag, bg = x1.cuda().requires_grad_(True), x2.cuda().requires_grad_(True)
    out = MarginRankingLoss(margin=margin, reduction=reduction)(ag, bg, t.cuda())
    (out * 1.7).backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ag.grad.cpu(), a.grad, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(bg.grad.cpu(), b.grad, rtol=1e-6, atol=1e-7)


class _G:                       # stand-in for the networkx graphs cross_val.split_train_val prepares (cross_val.py:158-184)
    def __init__(self, adj, feats, n):
        self.graph = {"adj": adj, "feats": feats, "num_nodes": n, "assign_feats": feats}


@pytest.mark.parametrize("kind", ["base", "diffpool"])
def test_triplet_batched_equals_three_b1_forwards(kind):
    from two_stage_gnn_amd import dense_encoders as E
    from two_stage_gnn_amd.triplet import tripletnet
    nmax, fin = 24, 6
    x, adj, sizes = dense_batch(41, 3, nmax, fin, sizes=[24, 11, 17], p_edge=0.2)

    class A:
        bias = True
    torch.manual_seed(4)
    if kind == "base":
        m = E.GcnEncoderGraph(fin, 8, 8, 2, 3, bn=True, args=A(), final_dim="output_dim")
    else:
        m = E.SoftPoolingGcnEncoder(nmax, fin, 8, 8, 2, 3, 8, assign_ratio=0.25, num_pooling=1, bn=True, linkpred=False, args=A(),
                                    assign_input_dim=fin, final_dim="output_dim")
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "conv" in k and k.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.3)
    m = m.cuda()
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    embeds = []
    for b in range(3):                                             # the reference: three separate B = 1 forwards
        if kind == "base":
            _, e = R.gcn_encoder(p_ref, x[b:b + 1], adj[b:b + 1], bn=True, final_dim="output_dim")
        else:
            _, e = R.diffpool_encoder(p_ref, x[b:b + 1], adj[b:b + 1], sizes[b:b + 1], 1, assign_x=x[b:b + 1], final_dim="output_dim")
        embeds.append(e)
    dp_ref = torch.nn.functional.pairwise_distance(embeds[0], embeds[1], 2)
    dn_ref = torch.nn.functional.pairwise_distance(embeds[0], embeds[2], 2)
    loss_ref = torch.nn.MarginRankingLoss(margin=1.0)(dp_ref, dn_ref, torch.tensor([-1.0]))
    loss_ref.backward()
    net = tripletnet(m)
    gs = [_G(adj[b].numpy(), x[b].numpy(), int(sizes[b])) for b in range(3)]
    dp, dn, ea, ep, en = net(*gs)
    torch.testing.assert_close(dp.cpu(), dp_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dn.cpu(), dn_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ea.cpu(), embeds[0].detach(), rtol=1e-4, atol=1e-4)
    loss = torch.nn.MarginRankingLoss(margin=1.0)(dp, dn, torch.tensor([-1.0]).cuda())
    loss.backward()
    for k, p in m.named_parameters():
        ref = p_ref[k].grad
        if ref is None or p.grad is None:
            continue
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 5e-3 * ref.abs().max().item() + 1e-6, (k, err, ref.abs().max().item())
    # the graphs stay resident after their first use (triplet._resident): a second step builds nothing and gives the same bits;
    # the per-step upload of the dense adjacencies (the reference's way, TSGNN_TRIPLET_CACHE=0) agrees
    from two_stage_gnn_amd import triplet as T3
    assert T3.RESIDENT and len(net._resident) >= 3
    before = {k: id(v) for k, v in net._resident.items()}
    dp2, dn2, ea2 = net(*gs)[:3]
    assert {k: id(v) for k, v in net._resident.items()} == before
    assert torch.equal(dp2, dp) and torch.equal(dn2, dn) and torch.equal(ea2, ea)
    dp3, dn3, ea3 = net(gs[1], gs[0], gs[2])[:3]                      # the same objects in another order: assembled from the same pieces
    torch.testing.assert_close(dn3.cpu(), torch.nn.functional.pairwise_distance(embeds[1], embeds[2], 2).detach(), rtol=1e-4, atol=1e-4)
    T3.RESIDENT = False
    try:
        dpu, dnu, eau = net(*gs)[:3]
    finally:
        T3.RESIDENT = True
    torch.testing.assert_close(dpu, dp, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(eau, ea, rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------------------- fused GraphSage stack: launch-fusion variants
def _stack_run(B, nmax, sizes, fin, hid, p_edge, flags, seed=5):
    from two_stage_gnn_amd import dense_encoders as E, sage_stack as S
    x, adj, sizes = dense_batch(seed, B, nmax, fin, sizes=sizes, p_edge=p_edge)

    class A:
        bias = True
    torch.manual_seed(1)
    m = E.GcnEncoderGraph(fin, hid, hid, 2, 3, bn=True, args=A(), final_dim="number_classes")
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("bias") and "conv" in k:
                p.copy_(torch.randn_like(p) * 0.2)
    m = m.cuda()
    old = {k: getattr(S, k) for k in flags}
    try:
        for k, v in flags.items():
            setattr(S, k, v)
        a, b = m(x.cuda(), adj.cuda(), sizes)
        label = (torch.arange(B) % 2).cuda()
        m.loss(b, label).backward()
    finally:
        for k, v in old.items():
            setattr(S, k, v)
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
    return a.detach().cpu(), b.detach().cpu(), grads, (m, x, adj, sizes)


ALL_ON = dict(GATHER_FUSED=True, MERGED_FWD=True, MERGED_BWD=True, FUSED_TAIL=True, HEAD_DU=True, FUSED_BN=True, SLOT_WGRAD=True, DU_MAP=True)


@pytest.mark.parametrize("off", ["GATHER_FUSED", "MERGED_FWD", "MERGED_BWD", "FUSED_TAIL", "HEAD_DU", "FUSED_BN", "SLOT_WGRAD", "DU_MAP"])
def test_stack_fusion_variants_agree(off):
    """every launch fusion of the GraphSage stack (aggregation inside the product, product + readout partial, slabs + dX,
    readout tail + head, the last layer's dU inside the head's backward launch — on the dense (graph, chunk) grid or on the
    host's list of non-empty chunks —) gives the results of the launch sequence it replaces"""
    sizes = dd_like_sizes(3, 6, nbar=70, nmax=150).tolist()
    ref = _stack_run(6, 150, sizes, 89, 128, 0.06, ALL_ON)
    alt = _stack_run(6, 150, sizes, 89, 128, 0.06, dict(ALL_ON, **{off: False}))
    torch.testing.assert_close(alt[0], ref[0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(alt[1], ref[1], rtol=1e-5, atol=1e-5)
    assert alt[2].keys() == ref[2].keys()
    for k in ref[2]:
        scale = ref[2][k].abs().max().item() + 1e-12
        assert (alt[2][k] - ref[2][k]).abs().max().item() <= 2e-4 * scale, k


@pytest.mark.parametrize("case", ["full_graph", "equal_sizes", "single_graph", "high_degree"])
def test_stack_edge_shapes_vs_oracle(case):
    """shapes that steer the fused stack onto its other branches: a graph that fills every slot (no ghost slot can be
    dropped), equal sizes, one graph, neighbour lists longer than the fixed-width table (CSR tail: no fused gather)"""
    from two_stage_gnn_amd.graph import GraphBatch
    if case == "full_graph":
        B, nmax, sizes, p = 5, 96, [96, 40, 61, 17, 80], 0.08
    elif case == "equal_sizes":
        B, nmax, sizes, p = 4, 120, [50, 50, 50, 50], 0.1
    elif case == "single_graph":
        B, nmax, sizes, p = 1, 200, [137], 0.05
    else:
        B, nmax, sizes, p = 3, 100, [90, 75, 60], 0.35
    a, b, grads, (m, x, adj, sizes_np) = _stack_run(B, nmax, sizes, 89, 128, p, ALL_ON, seed=9)
    if case == "high_degree":
        g = GraphBatch.from_dense(adj.cuda(), sizes=sizes_np, layout="packed")
        g.val = None
        assert g.ell()[2] is not None                     # the CSR tail exists: fallback branches are the ones tested
    p_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    a_ref, b_ref = R.gcn_encoder(p_ref, x, adj, bn=True, final_dim="number_classes")
    torch.nn.functional.cross_entropy(b_ref, torch.arange(B) % 2).backward()
    torch.testing.assert_close(a, a_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b, b_ref.detach(), rtol=1e-4, atol=1e-4)
    for k, gr in grads.items():
        ref = p_ref[k].grad
        if ref is None:
            continue
        assert (gr - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-7, k
        assert ((gr - ref).norm() / (ref.norm() + 1e-12)).item() < 1e-3, k


@pytest.mark.parametrize("B,nmax,nbar", [(6, 300, 120), (48, 64, 12), (3, 40, 33)])
def test_epilogue_readout_equals_row_scan(B, nmax, nbar, monkeypatch):
    """the last layer's max readout folded into its product's epilogue (packed atomicMax per panel and graph, ghost rows from
    the filler block; panels that span one, two or many graphs) == the tail kernel scanning the layer's rows: identical
    outputs, arg-max winners (through the gradients) and parameters after a step, bit for bit"""
    from two_stage_gnn_amd import dense_encoders as E, sage_stack
    sizes = dd_like_sizes(5, B, nbar=nbar, nmax=nmax)
    x, adj, sizes = dense_batch(9, B, nmax, 12, sizes=sizes.tolist(), p_edge=min(0.5, 5.0 / nbar))

    class A:
        bias = True
    res = []
    for on in (True, False):
        monkeypatch.setattr(sage_stack, "EPILOGUE_READOUT", on)
        torch.manual_seed(4)
        m = E.GcnEncoderGraph(12, 128, 128, 3, 3, bn=True, args=A(), final_dim="number_classes")
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith("bias") and "conv" in k:
                    p.copy_(torch.randn_like(p) * 0.3)        # ghost rows (normalised bias) can win the readout
        m = m.cuda()
        a, b = m(x.cuda(), adj.cuda(), sizes)
        m.loss(b, (torch.arange(B) % 3).cuda()).backward()
        res.append((a.detach().clone(), b.detach().clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]))
        assert len(res[-1][2]) >= 8
    torch.testing.assert_close(res[0][0], res[1][0], rtol=0, atol=0)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=0, atol=0)
    for ga, gb in zip(res[0][2], res[1][2]):
        torch.testing.assert_close(ga, gb, rtol=0, atol=0)


# ------------------------------------------------------------------------------------------------ DiffPool glue (round 3)
@pytest.mark.parametrize("K,N0,N1,ld_pad", [(64, 192, 64, 0), (64, 192, 0, 0), (24, 50, 24, 4), (33, 7, 33, 1)])
def test_ragged_tn_direct_vs_fp64(K, N0, N1, ld_pad):
    """out[b] = S[rows_b]^T X[rows_b] for two operands in one launch (csrc/ragged.hip; encoders.py:374-375) against fp64 per graph:
    empty, one-row, odd and > 512-row graphs, partial tiles, padded leading dimensions"""
    from two_stage_gnn_amd import _native as nat
    sizes = [0, 1, 7, 64, 129, 513, 2, 300, 0, 1025]
    gp = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device="cuda")
    R = int(sum(sizes))
    gen = torch.Generator(device="cuda").manual_seed(5)
    S = torch.randn(R + 3, K + ld_pad, generator=gen, device="cuda")[:, :K]
    X0 = torch.randn(R + 3, N0 + ld_pad, generator=gen, device="cuda")[:, :N0]
    X1 = torch.randn(R + 3, N1 + ld_pad, generator=gen, device="cuda")[:, :N1] if N1 else None
    B = len(sizes)
    assert nat.lib().tsgnn_ragged_tn_direct_supported(K, max(sizes))
    o0 = torch.full((B, K, N0), float("nan"), device="cuda")
    o1 = torch.full((B, K, N1), float("nan"), device="cuda") if N1 else None
    nat.call("ragged_tn_direct_f32", S, S.stride(0), K, gp, B, X0, X0.stride(0), N0, o0, X1, X1.stride(0) if N1 else 0, N1, o1)
    o0b = torch.empty_like(o0)
    nat.call("ragged_tn_direct_f32", S, S.stride(0), K, gp, B, X0, X0.stride(0), N0, o0b, None, 0, 0, None)
    assert torch.equal(o0, o0b)                                 # alone or beside the second operand: the same bits
    # ... and with the max readout of X0 riding along (padded slots: none / zero rows behind the real ones)
    nmax = max(sizes)
    for ghost in (0, 1):
        o0c = torch.empty_like(o0)
        ro = torch.full((B, N0 + 4), float("nan"), device="cuda")
        arg = torch.full((B, N0), -7, dtype=torch.int32, device="cuda")
        nat.call("ragged_tn_direct_ro_f32", S, S.stride(0), K, gp, B, X0, X0.stride(0), N0, o0c, None, 0, 0, None, ro, ro.stride(0), arg,
                 nmax, R, ghost)
        assert torch.equal(o0c, o0)
        off = 0
        for b, n in enumerate(sizes):
            if n:
                ref, idx = X0[off:off + n].max(0)
                refarg = idx.int() + off
            else:
                ref, refarg = torch.zeros(N0, device="cuda"), torch.full((N0,), -1, dtype=torch.int32, device="cuda")
            if ghost and n < nmax:
                lose = ref < 0 if n else torch.ones(N0, dtype=torch.bool, device="cuda")
                ref = torch.where(lose, torch.zeros_like(ref), ref)
                refarg = torch.where(lose, torch.full_like(refarg, R + n), refarg)
            assert torch.equal(ro[b, :N0], ref), (ghost, b, n)
            assert torch.equal(arg[b], refarg), (ghost, b, n)
            off += n
    off = 0
    for b, n in enumerate(sizes):
        s64 = S[off:off + n].double()
        for o, X in ((o0, X0), (o1, X1)):
            if X is None:
                continue
            ref = s64.t() @ X[off:off + n].double()
            err = (o[b].double() - ref).abs().max().item()
            assert err <= 2e-6 * max(1.0, n ** 0.5) * max(1.0, ref.abs().max().item()), (b, n, err)
        off += n


@pytest.mark.parametrize("nmax,F", [(65, 4), (200, 192), (1000, 256), (512, 70)])
def test_readout_max_one_launch_repeated_calls(nmax, F):
    """the max readout over more than 64 slots is ONE launch whose last-arriving workgroup per graph decodes (bn_readout.hip); its
    ticket counters must be back at zero after every call: repeated calls on one workspace, against torch.max over the padded rows"""
    from two_stage_gnn_amd import message_passing as mp, _native as nat
    from two_stage_gnn_amd.graph import GraphBatch
    B = 9
    x, adj, sizes = dense_batch(17, B, nmax, 3, sizes=[min(n, nmax) for n in (nmax, 1, 64, 65, nmax // 2, 7, nmax - 1, 130, 2)], p_edge=0.02)
    g = GraphBatch.from_dense(adj.cuda(), sizes)
    gen = torch.Generator(device="cuda").manual_seed(1)
    names = []
    for it in range(3):
        feat = torch.randn(g.total_rows, F, generator=gen, device="cuda") - 0.5 * it
        nat.trace = []
        try:
            out, arg = mp.readout_max(feat, g, return_arg=True)
            names = [t[2] for t in nat.trace]
        finally:
            nat.trace = None
        ref = torch.stack([feat[g.graph_ptr[b]:g.graph_ptr[b + 1]].max(0).values for b in range(B)])
        if g.n_ghost:                                                     # padded slots take part (trap T5): the ghost rows' values
            ghost = feat[g.n_rows:g.n_rows + g.n_ghost]
            for b in range(B):
                if sizes[b] < nmax:
                    ref[b] = torch.maximum(ref[b], ghost[int(sizes[b]):].max(0).values)
        assert torch.equal(out, ref), it
        assert torch.equal(feat[arg.long(), torch.arange(F, device="cuda").expand(B, F)], out), it
    assert len(names) == 1, names
    ws = list(g._readout_ws.values())[0]
    words = ws.numel() - (B + 1) // 2
    assert int(ws[words:].abs().sum()) == 0                               # the counters are back at zero


def test_diffpool_glue_variants_agree(monkeypatch):
    """round-3 launch removals of the DiffPool step — the levels' readouts written into one buffer (no torch.cat), their gradients
    added inside the contractions' backward launches, the embedding mask's clearing inside the last paired product launch, both
    first-contraction products as one launch — against the
    launch-by-launch forms: outputs, loss, every parameter gradient; and the launches really are gone"""
    from two_stage_gnn_amd import dense_encoders as E, sage_stack, diffpool as dp, synthetic, message_passing as mp, _native as nat

    class A:
        bias = True
    torch.manual_seed(3)
    hb = synthetic.host_batch(5, 8, "DD", 512)          # 512 * 0.125 = 64 assignment columns: both stacks' last layers share a launch
    g, x, lab = synthetic.to_device(hb, torch.device("cuda"))
    m = E.SoftPoolingGcnEncoder(512, 89, 64, 64, 2, 3, 64, assign_ratio=0.125, num_pooling=2, bn=True, linkpred=False, args=A(),
                                assign_input_dim=89, final_dim="number_classes").cuda()
    res = []
    for on in (False, True, False):                     # (the first pass also builds the batch's lazily built structures)
        monkeypatch.setattr(E, "READOUT_COLUMNS", on)
        monkeypatch.setattr(E, "READOUT_IN_CONTRACT", on)
        monkeypatch.setattr(mp, "HEAD_TAIL", on)
        monkeypatch.setattr(E, "SOFTMAX_IN_CONTRACT", on)
        monkeypatch.setattr(sage_stack, "ZERO_RIDER", on)
        monkeypatch.setattr(dp, "RAGGED_DIRECT", on)
        m.zero_grad(set_to_none=True)
        nat.trace = []
        try:
            a, b = m(x, g, hb["sizes"], assign_x=x)
            loss = m.loss(b, lab)
            loss.backward()
            names = [t[0] for t in nat.trace]
        finally:
            nat.trace = None
        res.append((a.detach().clone(), b.detach().clone(), loss.detach().clone(),
                    {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, names))
    mp.check_device_errors()
    (a1, b1, l1, g1, n1), (a0, b0, l0, g0, n0) = res[1:]
    assert "sage_multi_zero_f32" in n1 and "sage_multi_zero_f32" not in n0
    assert "contract_rows_bwd_ro_f32" in n1 and "contract_dense_bwd_ro_f32" in n1 and n1.count("readout_max_bwd_rows_f32") == 0
    assert "head2_fwd_ro_f32" in n1 and "head2_bwd_ro_f32" in n1 and n0.count("readout_max_bwd_rows_f32") == 3
    assert "ragged_tn_direct_ro_f32" in n1 and n1.count("ragged_tn_f32") == 0 and n0.count("ragged_tn_f32") == 2
    assert n1.count("readout_max_fwd_f32") == 0 and n0.count("readout_max_fwd_f32") == 3
    assert n1.count("row_softmax_masked_fwd_f32") == 1 and n0.count("row_softmax_masked_fwd_f32") == 2
    assert n1.count("row_softmax_masked_bwd_f32") == 1 and n0.count("row_softmax_masked_bwd_f32") == 2
    assert len(n1) <= len(n0) - 1
    torch.testing.assert_close(a1, a0, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(b1, b0, rtol=2e-5, atol=2e-6)
    assert set(g1) == set(g0)
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        # (another summation order in the contraction; the biases in front of a batch-norm have gradients that are rounding noise)
        err = (g1[k] - g0[k]).abs().max().item()
        assert err <= 1e-3 * g0[k].abs().max().item() + 5e-4 * gmax, (k, err, gmax)
