"""CPU oracle for the dense-batched message-passing path of Code/sage+gat+diffpool.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this file; the product package (two-stage-gnn_amd/) never does.

A plain torch-CPU restatement (fp32, differentiable, so autograd supplies the reference
gradients) of what the reference computes, in the reference's own dense padded layout
``x[B,Nmax,F]``, ``adj[B,Nmax,Nmax]``.  Each function cites the reference lines it follows.
Parity is PINNED: tests/test_oracle_golden.py checks every function here against the fixtures
in tests/golden/ that oracle/gen_golden.py captured from the reference itself.

Parameters are passed as a flat ``dict[str, Tensor]`` that uses the reference's state_dict key
names (``conv_first.weight`` …) so fixtures, oracle and product modules interchange.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm1d default, encoders.py:137
MASK_NEG = -9e15    # encoders_GAT.py:38


# --------------------------------------------------------------------------- GraphConv
def graph_conv(x, adj, weight, bias=None, add_self=False, normalize=False):
    """GraphConv.forward, encoders.py:30-42 (dropout omitted: p=0 on every hot-path config).
    SUM aggregation, optional +x, @W, +b, row-wise L2 normalise (F.normalize eps=1e-12)."""
    if adj.is_sparse:
        # the same product with the padded batch's block-diagonal adjacency held as ONE sparse [B*Nmax, B*Nmax] matrix
        # (bench.py's second CPU line: what a sparse CPU implementation of the reference would execute)
        B, N, Fi = x.shape
        y = torch.sparse.mm(adj, x.reshape(B * N, Fi)).reshape(B, N, Fi)
    else:
        y = torch.matmul(adj, x)                  # :33
    if add_self:
        y = y + x                                 # :35
    y = torch.matmul(y, weight)                   # :36
    if bias is not None:
        y = y + bias                              # :38
    if normalize:
        y = F.normalize(y, p=2, dim=2)            # :40
    return y


def bn_slots(x):
    """apply_bn, encoders.py:134-138: a *fresh* BatchNorm1d(Nmax) per call => gamma=1, beta=0,
    always batch statistics (even in eval), channel axis = node slot, statistics over
    (batch, feature), biased variance, eps=1e-5  (SURVEY trap T2)."""
    mean = x.mean(dim=(0, 2), keepdim=True)
    var = x.var(dim=(0, 2), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + BN_EPS)


def construct_mask(nmax, batch_num_nodes):
    """construct_mask, encoders.py:121-132: [B,Nmax,1] prefix mask of ones."""
    B = len(batch_num_nodes)
    m = torch.zeros(B, nmax)
    for i, n in enumerate(batch_num_nodes):
        m[i, : int(n)] = 1.0
    return m.unsqueeze(2)


def _conv(p, prefix, x, adj, add_self):
    return graph_conv(x, adj, p[prefix + ".weight"], p.get(prefix + ".bias"), add_self=add_self,
                      normalize=True)             # build_conv_layers passes normalize=True, :62-64


def _num_block(p, prefix):
    n = 0
    while "%s.%d.weight" % (prefix, n) in p:
        n += 1
    return n


def _linear(p, prefix, x):
    return F.linear(x, p[prefix + ".weight"], p[prefix + ".bias"])


def _heads(p, output, final_dim):
    """the three output modes, encoders.py:207-217 / :396-406 (pred_hidden_dims=[] => Linear)."""
    if final_dim == "pretrain":
        out = _linear(p, "map_model", output)
        return _linear(p, "map2_model", out), out
    if final_dim != "output_dim":
        vec = _linear(p, "pre_pred_model", output)
        return vec, _linear(p, "pred_model", vec)
    return output, _linear(p, "map_model", output)


# --------------------------------------------------------------------------- GcnEncoderGraph
def gcn_encoder_readouts(p, x, adj, bn=True, concat=True):
    """GcnEncoderGraph.forward up to the concatenated readout, encoders.py:177-205.
    max over the node axis INCLUDES padded ghost rows (mask built but unused, trap T5)."""
    add_self = not concat                          # :50
    nb = _num_block(p, "conv_block")
    x = _conv(p, "conv_first", x, adj, add_self)
    x = F.relu(x)
    if bn:
        x = bn_slots(x)
    outs = [x.max(dim=1)[0]]
    for i in range(nb):
        x = _conv(p, "conv_block.%d" % i, x, adj, add_self)
        x = F.relu(x)
        if bn:
            x = bn_slots(x)
        outs.append(x.max(dim=1)[0])
    x = _conv(p, "conv_last", x, adj, add_self)
    outs.append(x.max(dim=1)[0])
    return torch.cat(outs, dim=1) if concat else outs[-1]


def gcn_encoder(p, x, adj, bn=True, concat=True, final_dim="output_dim"):
    return _heads(p, gcn_encoder_readouts(p, x, adj, bn, concat), final_dim)


def gcn_forward(p, names, x, adj, mask=None, bn=True, concat=True):
    """gcn_forward, encoders.py:140-167: per-layer outputs concatenated on the feature axis,
    then multiplied by the mask."""
    first, block, last = names
    add_self = not concat
    x = _conv(p, first, x, adj, add_self)
    x = F.relu(x)
    if bn:
        x = bn_slots(x)
    xs = [x]
    for i in range(_num_block(p, block)):
        x = _conv(p, "%s.%d" % (block, i), x, adj, add_self)
        x = F.relu(x)
        if bn:
            x = bn_slots(x)
        xs.append(x)
    x = _conv(p, last, x, adj, add_self)
    xs.append(x)
    t = torch.cat(xs, dim=2)
    if mask is not None:
        t = t * mask
    return t


# --------------------------------------------------------------------------- DiffPool
def diffpool_contract(s, z, adj):
    """encoders.py:374-375:  X' = S^T Z ;  A' = S^T A S."""
    st = torch.transpose(s, 1, 2)
    return torch.matmul(st, z), st @ adj @ s


def diffpool_encoder(p, x, adj, batch_num_nodes, num_pooling, assign_x=None, final_dim="output_dim",
                     return_assign=False):
    """SoftPoolingGcnEncoder.forward, encoders.py:327-406 (bn is always True there: the ctor does
    not forward ``bn`` to the base class, :249-250)."""
    x_a = x if assign_x is None else assign_x
    nmax = adj.size(1)
    mask = construct_mask(nmax, batch_num_nodes) if batch_num_nodes is not None else None
    emb = gcn_forward(p, ("conv_first", "conv_block", "conv_last"), x, adj, mask)
    outs = [emb.max(dim=1)[0]]
    s = None
    for i in range(num_pooling):
        m = construct_mask(nmax, batch_num_nodes) if (batch_num_nodes is not None and i == 0) else None
        a = gcn_forward(p, ("assign_conv_first_modules.%d" % i, "assign_conv_block_modules.%d" % i,
                            "assign_conv_last_modules.%d" % i), x_a, adj, m)
        s = torch.softmax(_linear(p, "assign_pred_modules.%d" % i, a), dim=-1)      # :369
        if m is not None:
            s = s * m                                                                 # :371
        x, adj = diffpool_contract(s, emb, adj)                                       # :374-375
        x_a = x
        emb = gcn_forward(p, ("conv_first_after_pool.%d" % i, "conv_block_after_pool.%d" % i,
                              "conv_last_after_pool.%d" % i), x, adj, None)
        outs.append(emb.max(dim=1)[0])
    res = _heads(p, torch.cat(outs, dim=1), final_dim)
    return res + (s,) if return_assign else res


# --------------------------------------------------------------------------- GAT (dense, column softmax)
def gat_head(x, adj, w, a, slope=0.2, concat=True, att_mult=None):
    """DGATHead.forward, encoders_GAT.py:29-49.  Uses graph 0's features for every graph (T4);
    softmax(dim=1) of a [B,N,N] tensor normalises over the ROW index i, i.e. per source column j
    (T3); masked entries are -9e15 so an all-masked column becomes uniform 1/N.
    att_mult [B,N,N] (optional): F.dropout(attention, p, training=True) of :42 with the multipliers (0 or 1/(1-p)) supplied by the
    caller — torch's own random stream cannot be reproduced by another implementation, the arithmetic can."""
    h = x[0] @ w                                                  # :32
    fo = w.size(1)
    e = F.leaky_relu((h @ a[:fo]) + (h @ a[fo:]).t(), slope)      # e_ij = a1.h_i + a2.h_j, :35-36
    att = torch.where(adj > 0, e.expand_as(adj), torch.full_like(adj, MASK_NEG))
    att = torch.softmax(att, dim=1)                               # :41
    if att_mult is not None:
        att = att * att_mult                                      # :42
    hp = torch.matmul(att, h)                                     # :43
    return F.elu(hp) if concat else hp


def gat_layer(p, prefix, x, adj, concat=True, slope=0.2, att_mult=None):
    """DGATLayer.forward, encoders_GAT.py:70-84.  att_mult: per head, see gat_head."""
    heads = []
    i = 0
    while "%s.attention_%d.w" % (prefix, i) in p:
        heads.append(gat_head(x, adj, p["%s.attention_%d.w" % (prefix, i)],
                              p["%s.attention_%d.a" % (prefix, i)], slope, concat,
                              None if att_mult is None else att_mult[i]))
        i += 1
    if concat:
        return torch.cat(heads, dim=2)
    s = heads[0]
    for h in heads[1:]:
        s = s + h
    return F.elu(s / len(heads))


def gat_encoder(p, x, adj, final_dim="output_dim"):
    """DGATEncoderGraph.forward, encoders_GAT.py:175-198 (map2_model is Identity, :117)."""
    x = gat_layer(p, "conv_first", x, adj, True)
    i = 0
    while "conv_block.%d.attention_0.w" % i in p:
        x = gat_layer(p, "conv_block.%d" % i, x, adj, True)
        i += 1
    x = gat_layer(p, "conv_last", x, adj, False)
    x = x.max(dim=1)[0]
    if final_dim != "output_dim":
        return x, _linear(p, "pred_model", x)
    return x, _linear(p, "map_model", x)


def diffpool_link_loss(s, adj, batch_num_nodes=None, clamp=1.0, eps=1e-7, adj_hop=1):
    """SoftPoolingGcnEncoder.loss's link-prediction term, encoders.py:416-438 (adj_hop: the powers of S S^T of :419-423).  s: the assignment tensor
    [B, N, K] (masked rows already zero, :371), adj [B, N, N].  The reference clamps with ``torch.Tensor(1)`` (:424), an
    UNINITIALISED value: it is a parameter here (parity of this term is pinned by the source lines, not by a vector)."""
    pred0 = s @ s.transpose(1, 2)                                                            # :418
    tmp, pred = pred0, pred0
    for _ in range(adj_hop - 1):                                                             # :421-423
        tmp = tmp @ pred0
        pred = pred + tmp
    pred = torch.minimum(pred, torch.tensor(float(clamp), dtype=pred.dtype))                # :424
    ll = -adj * torch.log(pred + eps) - (1 - adj) * torch.log(1 - pred + eps)                # :428
    B, N = adj.size(0), adj.size(1)
    if batch_num_nodes is None:
        entries = N * N * B                                                                 # :430
    else:
        entries = float(sum(int(n) * int(n) for n in batch_num_nodes))                      # :433
        m = torch.zeros(B, N, 1)
        for b, n in enumerate(batch_num_nodes):
            m[b, :int(n)] = 1
        ll = ll * (m @ m.transpose(1, 2))                                                    # :434-436
    return ll.sum() / float(entries)                                                        # :438
