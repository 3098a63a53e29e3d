"""Drop-in for Code/sage+gat+diffpool/tripletnet.py:11-45 — the 2stg / 2stg+ training step.

The reference runs the encoder three times at B = 1 (anchor, positive, negative: three uploads of a dense
[1,Nmax,Nmax] adjacency and three sequential forwards, tripletnet.py:18-38).  Here the three graphs form ONE
block-diagonal GraphBatch and go through the kernels once; ``per_graph_bn`` keeps every graph on the per-row statistics it
would have alone in its batch, so embeddings and gradients equal the three separate B = 1 calls.
"""
import os
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as nat
from . import message_passing as mp
from .graph import GraphBatch


# ----------------------------------------------------------------------------- the graphs of the dataset, resident
# The reference uploads the three dense [1, Nmax, Nmax] adjacencies of a triplet at every step (tripletnet.py:18-33: 12 MB for DD)
# although the sampler draws them from a fixed set of graph objects (triplet_sampler.py).  Here a graph object's CSR rows, features
# and assignment features go to the device ONCE (keyed by its `adj` array; ~100 KB for a DD graph, so the whole dataset stays
# resident in a corner of the 288 GB) and a step only concatenates three of them into the block-diagonal batch: a handful of small
# device copies, no PCIe traffic, no scan of 3 Nmax^2 floats, no host synchronisation.  The arrays are taken to be immutable, as the
# reference treats them (cross_val.py:158-184 builds them once); TSGNN_TRIPLET_CACHE=0 restores the upload per step.
RESIDENT = os.environ.get("TSGNN_TRIPLET_CACHE", "1") != "0"
_MAX_RESIDENT = 1 << 17


class _Resident:
    __slots__ = ("ref", "n", "nnz", "nmax", "rowptr", "col", "val", "feats", "assign", "symmetric")


def _resident(obj, dev, cache):
    """the device-side pieces of one graph object (built at its first use)"""
    adj = obj.graph["adj"]
    key = (id(adj), dev.index)
    hit = cache.get(key)
    if hit is not None and hit.ref() is adj:
        return hit
    a = np.asarray(adj, dtype=np.float32)
    n = int(obj.graph["num_nodes"])
    if a.ndim != 2 or a.shape[0] != a.shape[1] or not 0 <= n <= a.shape[0]:
        raise ValueError("adj must be [Nmax, Nmax] with num_nodes <= Nmax")
    sub = a[:n, :n]
    r, c = np.nonzero(sub)                                              # row-major: columns ascending inside a row, as from_dense fills them
    rp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=n), out=rp[1:])
    v = sub[r, c]
    e = _Resident()
    e.ref = weakref.ref(adj) if isinstance(adj, np.ndarray) else (lambda: adj)
    e.n, e.nnz, e.nmax = n, int(r.size), int(a.shape[0])
    e.symmetric = bool(np.array_equal(sub, sub.T))
    e.rowptr = torch.from_numpy(rp).to(dev)
    e.col = torch.from_numpy(c.astype(np.int32)).to(dev)
    e.val = None if bool((v == 1.0).all()) else torch.from_numpy(np.ascontiguousarray(v)).to(dev)

    def rows(key_):
        f = np.asarray(obj.graph[key_], dtype=np.float32)[:n]
        ld = (f.shape[1] + 3) // 4 * 4                                  # 16-byte rows for the float4 gather
        out = torch.zeros(n, ld, dtype=torch.float32, device=dev)
        out[:, :f.shape[1]] = torch.from_numpy(np.ascontiguousarray(f)).to(dev)
        return out

    e.feats = rows("feats")
    fa, ff = np.asarray(obj.graph["assign_feats"]), np.asarray(obj.graph["feats"])
    e.assign = None if (fa.shape == ff.shape and np.array_equal(fa, ff)) else rows("assign_feats")
    if len(cache) >= _MAX_RESIDENT:
        cache.clear()
    cache[key] = e
    return e


def _ghost_zeros(nmax, ld, dev, cache):
    key = ("z", nmax, ld, dev.index)
    z = cache.get(key)
    if z is None:
        z = cache[key] = torch.zeros(nmax, ld, dtype=torch.float32, device=dev)
    return z


def _assemble(parts, dev, cache):
    """three resident graphs -> (GraphBatch, feature rows, assignment rows or None): the block-diagonal batch of packed rows +
    Nmax empty ghost-slot rows (GraphBatch.from_csr's layout)"""
    nmax = parts[0].nmax
    if any(p.nmax != nmax for p in parts):
        raise ValueError("the graphs of a triplet must be padded to the same Nmax")
    sizes = np.array([p.n for p in parts], dtype=np.int64)
    nnz = int(sum(p.nnz for p in parts))
    rps, cols, e0, r0 = [], [], 0, 0
    for p in parts:
        rps.append(p.rowptr[:-1] + e0 if e0 else p.rowptr[:-1])
        cols.append(p.col + r0 if r0 else p.col)
        e0 += p.nnz
        r0 += p.n
    rps.append(torch.full((nmax + 1,), nnz, dtype=torch.int32, device=dev))
    weighted = any(p.val is not None for p in parts)
    val = torch.cat([p.val if p.val is not None else torch.ones(p.nnz, device=dev) for p in parts]) if weighted else None
    col = torch.cat(cols) if nnz else torch.zeros(1, dtype=torch.int32, device=dev)
    g = GraphBatch.from_csr(torch.cat(rps), col, val, sizes, nmax, assume_symmetric=all(p.symmetric for p in parts))
    g.nnz = nnz
    x = torch.cat([p.feats for p in parts] + [_ghost_zeros(nmax, parts[0].feats.size(1), dev, cache)])
    xa = None
    if any(p.assign is not None for p in parts):
        pa = [p.assign if p.assign is not None else p.feats for p in parts]
        xa = torch.cat(pa + [_ghost_zeros(nmax, pa[0].size(1), dev, cache)])
    return g, x, xa, sizes


# ----------------------------------------------------------------------------- embeddings + distances: one launch each way
FUSED_TAIL = os.environ.get("TSGNN_TRIPLET_TAIL", "1") != "0"
_EPS = 1e-6                      # F.pairwise_distance's default


class _TripletTail(torch.autograd.Function):
    """(readouts r[3, D], map_model's weight [E, D], bias) -> (dist_p[1], dist_n[1], embed_a[1, E], embed_p, embed_n): the Linear on
    the three readout rows and both pairwise distances (tripletnet.py:35-45) as tsgnn_triplet_embed_fwd_f32 / _bwd_f32.  The five
    outputs are separate tensors, so no slice (and no zero-filled slice gradient) is launched around them."""

    @staticmethod
    def forward(ctx, r, w, b):
        w0 = w
        r, w = r.contiguous(), w.contiguous()
        D, E = int(w.size(1)), int(w.size(0))
        embed = torch.empty(3, E, dtype=torch.float32, device=r.device)
        dist = torch.empty(2, dtype=torch.float32, device=r.device)
        nat.call("triplet_embed_fwd_f32", r, r.stride(0), w, w.stride(0), b, D, E, _EPS, embed, dist)
        ctx.save_for_backward(r, w, embed, dist)
        ctx.has_bias = b is not None
        ctx.params = (w0, b)                              # (the Parameter objects: their slices of a trainer's flat gradient bucket)
        ctx.set_materialize_grads(False)                  # an unused output's gradient arrives as None, not as a zero-filled tensor
        outs = (dist[0:1], dist[1:2], embed[0:1], embed[1:2], embed[2:3])
        return outs

    @staticmethod
    def backward(ctx, g_dp, g_dn, g_a, g_p, g_n):
        r, w, embed, dist = ctx.saved_tensors
        D, E = int(w.size(1)), int(w.size(0))
        dev = r.device
        c = lambda t: t.contiguous() if t is not None else None
        d_r = torch.empty(3, D, dtype=torch.float32, device=dev)
        # straight into the trainer's flat gradient bucket when one is installed (FlatTrainer): no AccumulateGrad copy, no zeroing
        dw, sw = mp._sink_or_new(ctx.params[0], (E, D), dev)
        db, sb = mp._sink_or_new(ctx.params[1], (E,), dev) if ctx.has_bias else (None, False)
        nat.call("triplet_embed_bwd_f32", r, r.stride(0), w, w.stride(0), D, E, _EPS, embed, dist, c(g_dp), c(g_dn), c(g_a), c(g_p), c(g_n), d_r,
                 d_r.stride(0), dw, dw.stride(0), db)
        return d_r, (None if sw else dw), (None if sb else db)


class _MarginRank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, target, margin, mean):
        x1, x2, target = x1.contiguous().view(-1), x2.contiguous().view(-1), target.contiguous().view(-1).float()
        n = int(x1.numel())
        loss = torch.empty(1, dtype=torch.float32, device=x1.device)
        coef = torch.empty(n, dtype=torch.float32, device=x1.device)
        nat.call("margin_rank_fwd_f32", x1, x2, target, n, float(margin), int(mean), loss, coef)
        ctx.save_for_backward(coef)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        coef, = ctx.saved_tensors
        n = int(coef.numel())
        dx1 = torch.empty(n, dtype=torch.float32, device=coef.device) if ctx.needs_input_grad[0] else None
        dx2 = torch.empty(n, dtype=torch.float32, device=coef.device) if ctx.needs_input_grad[1] else None
        nat.call("margin_rank_bwd_f32", g.contiguous().view(-1), coef, n, dx1, dx2)
        return dx1, dx2, None, None, None


class MarginRankingLoss(nn.Module):
    """torch.nn.MarginRankingLoss (train_triplet.py:235: `criterion = torch.nn.MarginRankingLoss(margin=args.alpha)`) as one launch
    forward and one backward; same constructor and call.  Inputs that are not float32 CUDA vectors of one shape, or reduction
    'none', go to torch's implementation."""

    def __init__(self, margin=0.0, size_average=None, reduce=None, reduction="mean"):
        super().__init__()
        self.margin, self.reduction = float(margin), reduction
        self._torch = nn.MarginRankingLoss(margin=margin, size_average=size_average, reduce=reduce, reduction=reduction)
        self.reduction = self._torch.reduction

    def forward(self, input1, input2, target):
        if (self.reduction in ("mean", "sum") and input1.is_cuda and input1.dtype == torch.float32 and input2.dtype == torch.float32
                and input1.shape == input2.shape == target.shape and input1.numel() > 0):
            return _MarginRank.apply(input1, input2, target, self.margin, self.reduction == "mean")
        return self._torch(input1, input2, target)


def tail_ok(model, r):
    lin = getattr(model, "map_model", None)
    return (FUSED_TAIL and isinstance(lin, nn.Linear) and r is not None and r.is_cuda and r.dim() == 2 and r.size(0) == 3
            and r.dtype == torch.float32 and lin.in_features == r.size(1) and lin.in_features % 4 == 0 and lin.out_features <= 512
            and lin.weight.dtype == torch.float32)


def _rows_models():
    """encoders whose forward takes (packed rows, GraphBatch with ghost-slot rows): the GraphSage / DiffPool family.  Others (the GAT
    encoder packs its batch itself, with one ghost representative per graph) keep the dense inputs."""
    from .dense_encoders import GcnEncoderGraph
    return (GcnEncoderGraph,)


class tripletnet(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self._resident = {}

    def _embed(self, x, g_or_adj, sizes, assign_x):
        """model forward with the per-graph batch-norm statistics of a B = 1 call -> (dist_p, dist_n, embed_a, embed_p, embed_n)"""
        m = self.model
        prev = getattr(m, "per_graph_bn", False)
        fuse = (FUSED_TAIL and isinstance(m, _rows_models()) and getattr(m, "final_dim", None) in ("output_dim", "pretrain")
                and isinstance(getattr(m, "map_model", None), nn.Linear))       # (encoders whose _heads honours _defer_map)
        m.per_graph_bn = True
        m._defer_map = fuse
        try:
            out, embed = m(x, g_or_adj, sizes, assign_x=assign_x)
        finally:
            m.per_graph_bn = prev
            m._defer_map = False
        if fuse:
            r = out if m.final_dim == "output_dim" else embed      # the concatenated readouts (encoders.py:201-205)
            if tail_ok(m, r):
                return _TripletTail.apply(r, m.map_model.weight, m.map_model.bias)
            embed = m.map_model(r)
        embed_a, embed_p, embed_n = embed[0:1], embed[1:2], embed[2:3]
        return (F.pairwise_distance(embed_a, embed_p, 2), F.pairwise_distance(embed_a, embed_n, 2), embed_a, embed_p, embed_n)

    @staticmethod
    def _stack(graphs, key, device):
        return torch.as_tensor(np.stack([np.asarray(g.graph[key], dtype=np.float32) for g in graphs]), device=device)

    def forward(self, a, p, n):
        """a, p, n: objects with ``.graph`` = {'adj','feats','num_nodes','assign_feats'} as cross_val.split_train_val
        prepares them (cross_val.py:158-184)."""
        dev = next(self.model.parameters()).device
        trip = (a, p, n)
        if RESIDENT and dev.type == "cuda" and isinstance(self.model, _rows_models()):
            g, x, xa, sizes = _assemble([_resident(t, dev, self._resident) for t in trip], dev, self._resident)
            return self._embed(x, g, sizes, x if xa is None else xa)
        adj = self._stack(trip, "adj", dev)
        h0 = self._stack(trip, "feats", dev)
        assign = self._stack(trip, "assign_feats", dev)
        sizes = np.array([int(g.graph["num_nodes"]) for g in trip])
        return self._embed(h0, adj, sizes, assign if assign.shape == h0.shape and not torch.equal(assign, h0) else h0)
