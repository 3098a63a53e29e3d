"""CPU oracle for the torch_geometric operators on the SAGPool path (Code/sag) and the PyG-named
operators BASELINE.json's north_star lists (SAGEConv / GATConv / SAGPooling / dense_diff_pool).

TEST INFRASTRUCTURE (same import rule as dense_ref.py).

PARITY UNPINNED: the arithmetic of GCNConv / topk / filter_adj / global_*_pool lives in the third-party
package torch-geometric, pinned by the reference at "1.16.3" (README.md:20 — no such release; 1.6.3 is the
torch-1.7-era version, README.md:16).  It is absent from /root/reference and not installed here, and the
reference holds no tests or golden vectors at that boundary.  The functions below restate PyG 1.6.x's
published formulas; composition follows the reference's own call sites:
Code/sag/layers.py:14-25 (SAGPool), Code/sag/network.py:30-53 (Net).  SAGEConv / GATConv / SAGPooling /
dense_diff_pool have no call site in the reference at all.
Plain torch CPU, differentiable (autograd supplies reference gradients); loops only over graphs.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- GCNConv (a11)
def gcn_norm(edge_index, num_nodes, edge_weight=None, improved=False, dtype=torch.float32):
    """add_remaining_self_loops(fill 1, or 2 when improved) ; deg over targets ; D^-1/2 (A+I) D^-1/2.
    dtype: the arithmetic type (float64 for the arbitration runs of tests/test_gpu_fullsize.py)."""
    row, col = edge_index[0], edge_index[1]
    if edge_weight is None:
        edge_weight = torch.ones(row.numel(), dtype=dtype)
    mask = row != col
    loop_w = torch.full((num_nodes,), 2.0 if improved else 1.0, dtype=edge_weight.dtype)
    loop_w[row[~mask]] = edge_weight[~mask]                 # existing self loops keep their weight
    loop = torch.arange(num_nodes)
    row = torch.cat([row[mask], loop])
    col = torch.cat([col[mask], loop])
    w = torch.cat([edge_weight[mask], loop_w])
    deg = torch.zeros(num_nodes, dtype=w.dtype).index_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return row, col, dis[row] * w * dis[col]


def gcn_conv(x, edge_index, weight, bias=None, edge_weight=None, improved=False):
    """out = A_hat (x W) + b ; message flows source (edge_index[0]) -> target (edge_index[1])."""
    n = x.size(0)
    row, col, w = gcn_norm(edge_index, n, edge_weight, improved, dtype=x.dtype)
    xw = x @ weight
    out = torch.zeros(n, weight.size(1), dtype=x.dtype).index_add_(0, col, xw[row] * w.unsqueeze(1))
    return out + bias if bias is not None else out


# ----------------------------------------------------------------------------- topk / filter_adj (a12, a13)
def topk(score, ratio, batch, min_score=None):
    """per graph keep ceil(ratio*n) highest scores, descending (float32 ceil as PyG computes it);
    ties -> smaller index (PyG's sort is unstable; fixtures avoid ties).  min_score: PyG's threshold mode instead — every
    node whose score exceeds min(min_score, max of its graph - 1e-7), in node order."""
    B = int(batch.max()) + 1 if batch.numel() else 0
    if min_score is not None:
        keep = []
        for b in range(B):
            idx = (batch == b).nonzero().view(-1)
            thr = min(float(min_score), float(score[idx].max()) - 1e-7)
            keep.append(idx[score[idx] > thr])
        return torch.cat(keep) if keep else torch.zeros(0, dtype=torch.long)
    perm = []
    for b in range(B):
        idx = (batch == b).nonzero().view(-1)
        n = idx.numel()
        k = int(torch.ceil(torch.tensor(ratio, dtype=torch.float32) * torch.tensor(float(n), dtype=torch.float32)))
        s = score[idx].detach()
        order = sorted(range(n), key=lambda i: (-float(s[i]), i))[:k]
        perm.append(idx[torch.tensor(order, dtype=torch.long)] if k else idx[:0])
    return torch.cat(perm) if perm else torch.zeros(0, dtype=torch.long)


def filter_adj(edge_index, perm, num_nodes):
    mask = torch.full((num_nodes,), -1, dtype=torch.long)
    mask[perm] = torch.arange(perm.numel())
    row, col = mask[edge_index[0]], mask[edge_index[1]]
    keep = (row >= 0) & (col >= 0)
    return torch.stack([row[keep], col[keep]], 0)


# ----------------------------------------------------------------------------- readouts (a14)
def global_max_pool(x, batch, B):
    return torch.stack([x[batch == b].max(dim=0)[0] for b in range(B)])


def global_mean_pool(x, batch, B):
    return torch.stack([x[batch == b].mean(dim=0) for b in range(B)])


# ----------------------------------------------------------------------------- SAGPool / Net (a10)
def sag_pool(x, edge_index, batch, ratio, score_w, score_b):
    """Code/sag/layers.py:14-25."""
    if batch is None:
        batch = edge_index.new_zeros(x.size(0))
    score = gcn_conv(x, edge_index, score_w, score_b).squeeze(-1)
    perm = topk(score, ratio, batch)
    x = x[perm] * torch.tanh(score[perm]).view(-1, 1)
    return x, filter_adj(edge_index, perm, score.size(0)), batch[perm], perm


def sag_net(p, x, edge_index, ratio, batch=None, conv="gcn"):
    """Code/sag/network.py:30-53 (eval mode: dropout off).  batch=None reproduces the reference, which
    discards data.batch (:32, trap T6).  conv = "sage": the network's GCNConv layers replaced by PyG SAGEConv (BASELINE config 4 as
    worded: "SAGPool + SAGEConv"); the SAGPool layers are the reference's (layers.py:14-25)."""
    outs = []
    for i in (1, 2, 3):
        if conv == "sage":
            x = F.relu(sage_conv(x, edge_index, p["conv%d.lin_l.weight" % i], p["conv%d.lin_l.bias" % i], p["conv%d.lin_r.weight" % i]))
        else:
            x = F.relu(gcn_conv(x, edge_index, p["conv%d.weight" % i], p["conv%d.bias" % i]))
        x, edge_index, batch, _ = sag_pool(x, edge_index, batch, ratio, p["pool%d.score_layer.weight" % i],
                                           p["pool%d.score_layer.bias" % i])
        B = int(batch.max()) + 1
        outs.append(torch.cat([global_max_pool(x, batch, B), global_mean_pool(x, batch, B)], dim=1))
    x = outs[0] + outs[1] + outs[2]
    x = F.relu(F.linear(x, p["lin1.weight"], p["lin1.bias"]))
    x = F.relu(F.linear(x, p["lin2.weight"], p["lin2.bias"]))
    return F.log_softmax(F.linear(x, p["lin3.weight"], p["lin3.bias"]), dim=-1)


# ----------------------------------------------------------------------------- north_star-named PyG ops (a15)
def sage_conv(x, edge_index, w_l, b_l, w_r):
    """SAGEConv: lin_l(mean_{j in N(i)} x_j) + lin_r(x_i)."""
    n = x.size(0)
    row, col = edge_index[0], edge_index[1]
    s = torch.zeros(n, x.size(1), dtype=x.dtype).index_add_(0, col, x[row])
    deg = torch.zeros(n, dtype=x.dtype).index_add_(0, col, torch.ones(row.numel(), dtype=x.dtype)).clamp(min=1)
    out = F.linear(s / deg.unsqueeze(1), w_l, b_l)
    return out + F.linear(x, w_r)


def sage_net(p, x, edge_index, batch, num_layers):
    """two_stage_gnn_amd.pyg.SageNet: Code/sag/network.py:30-53's shape (conv + ReLU, [gmp || gap] per layer summed, three Linear,
    log_softmax; eval mode) with SAGEConv layers and no pooling — BASELINE configs 1-2 as worded."""
    B = int(batch.max()) + 1
    out = None
    for l in range(num_layers):
        x = F.relu(sage_conv(x, edge_index, p["convs.%d.lin_l.weight" % l], p["convs.%d.lin_l.bias" % l], p["convs.%d.lin_r.weight" % l]))
        r = torch.cat([global_max_pool(x, batch, B), global_mean_pool(x, batch, B)], dim=1)
        out = r if out is None else out + r
    x = F.relu(F.linear(out, p["lin1.weight"], p["lin1.bias"]))
    x = F.relu(F.linear(x, p["lin2.weight"], p["lin2.bias"]))
    return F.log_softmax(F.linear(x, p["lin3.weight"], p["lin3.bias"]), dim=-1)


def graph_conv(x, edge_index, w_l, b_l, w_r):
    """PyG GraphConv (SAGPooling's default scorer): lin_l(sum_j x_j) + lin_r(x_i)."""
    n = x.size(0)
    s = torch.zeros(n, x.size(1), dtype=x.dtype).index_add_(0, edge_index[1], x[edge_index[0]])
    return F.linear(s, w_l, b_l) + F.linear(x, w_r)


def gat_conv(x, edge_index, w, att_l, att_r, bias, heads, concat=True, slope=0.2, drop_mult=None):
    """GATConv: shared lin, per-TARGET edge softmax, self loops added, heads concatenated or averaged.  drop_mult [E', heads]
    (E' = the self-looped edge list: kept edges in order, then one loop per node): F.dropout's multiplier on the coefficients."""
    n = x.size(0)
    C = w.size(0) // heads
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    loop = torch.arange(n)
    row, col = torch.cat([row[keep], loop]), torch.cat([col[keep], loop])
    h = F.linear(x, w).view(n, heads, C)
    al = (h * att_l.view(1, heads, C)).sum(-1)
    ar = (h * att_r.view(1, heads, C)).sum(-1)
    e = F.leaky_relu(al[row] + ar[col], slope)
    emax = torch.full((n, heads), -float("inf"), dtype=x.dtype).scatter_reduce(0, col.view(-1, 1).expand(-1, heads), e, "amax")
    p = torch.exp(e - emax[col])
    den = torch.zeros(n, heads, dtype=x.dtype).index_add_(0, col, p)
    alpha = p / den[col]
    if drop_mult is not None:
        alpha = alpha * drop_mult
    out = torch.zeros(n, heads, C, dtype=x.dtype).index_add_(0, col, h[row] * alpha.unsqueeze(-1))
    out = out.reshape(n, heads * C) if concat else out.mean(dim=1)
    return out + bias if bias is not None else out


def gat_net(p, x, edge_index, batch, num_layers, heads, drop_mults=None):
    """two_stage_gnn_amd.pyg.GatNet: GATConv (concat) + ELU ..., GATConv (mean over heads), global_max_pool, Linear, log_softmax"""
    B = int(batch.max()) + 1
    for l in range(num_layers):
        last = l == num_layers - 1
        x = gat_conv(x, edge_index, p["convs.%d.lin_l.weight" % l], p["convs.%d.att_l" % l], p["convs.%d.att_r" % l],
                     p["convs.%d.bias" % l], heads, concat=not last, drop_mult=None if drop_mults is None else drop_mults[l])
        if not last:
            x = F.elu(x)
    r = global_max_pool(x, batch, B)
    return F.log_softmax(F.linear(r, p["lin.weight"], p["lin.bias"]), dim=-1)


def sag_pooling(x, edge_index, batch, ratio, w_l, b_l, w_r):
    """PyG SAGPooling (GraphConv scorer, tanh BEFORE top-k, multiplier 1)."""
    if batch is None:
        batch = edge_index.new_zeros(x.size(0))
    score = torch.tanh(graph_conv(x, edge_index, w_l, b_l, w_r).view(-1))
    perm = topk(score, ratio, batch)
    x = x[perm] * score[perm].view(-1, 1)
    return x, filter_adj(edge_index, perm, score.numel()), batch[perm], perm, score[perm]


def sage_pool_net(p, x, edge_index, batch, ratio, return_perms=False):
    """two_stage_gnn_amd.pyg.SagePoolNet: Code/sag/network.py:30-53's shape with SAGEConv layers and PyG SAGPooling (GraphConv scorer)
    — BASELINE config 4 as worded; eval mode"""
    B = int(batch.max()) + 1
    out, perms = None, []
    for l in range(3):
        x = F.relu(sage_conv(x, edge_index, p["convs.%d.lin_l.weight" % l], p["convs.%d.lin_l.bias" % l], p["convs.%d.lin_r.weight" % l]))
        x, edge_index, batch, perm, _ = sag_pooling(x, edge_index, batch, ratio, p["pools.%d.gnn.lin_l.weight" % l],
                                                    p["pools.%d.gnn.lin_l.bias" % l], p["pools.%d.gnn.lin_r.weight" % l])
        perms.append(perm)
        r = torch.cat([global_max_pool(x, batch, B), global_mean_pool(x, batch, B)], dim=1)
        out = r if out is None else out + r
    h = F.relu(F.linear(out, p["lin1.weight"], p["lin1.bias"]))
    h = F.relu(F.linear(h, p["lin2.weight"], p["lin2.bias"]))
    y = F.log_softmax(F.linear(h, p["lin3.weight"], p["lin3.bias"]), dim=-1)
    return (y, perms) if return_perms else y


def topk_pooling(x, edge_index, batch, ratio, weight):
    """PyG TopKPooling (imported, never called, Code/sag/network.py:3): score = tanh(x . p / ||p||), top-k on the score,
    x[perm] * score[perm]; weight [1, F]."""
    if batch is None:
        batch = edge_index.new_zeros(x.size(0))
    score = torch.tanh((x * weight).sum(dim=-1) / weight.norm(p=2, dim=-1))
    perm = topk(score, ratio, batch)
    x = x[perm] * score[perm].view(-1, 1)
    return x, filter_adj(edge_index, perm, score.numel()), batch[perm], perm, score[perm]


def dense_diff_pool(x, adj, s, mask=None, eps=1e-15):
    s = torch.softmax(s, dim=-1)
    if mask is not None:
        m = mask.view(x.size(0), x.size(1), 1).to(x.dtype)
        x, s = x * m, s * m
    out = s.transpose(1, 2) @ x
    out_adj = s.transpose(1, 2) @ adj @ s
    link = torch.norm(adj - s @ s.transpose(1, 2), p=2) / adj.numel()
    ent = (-s * torch.log(s + eps)).sum(dim=-1).mean()
    return out, out_adj, link, ent
