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


class tripletnet(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self._resident = {}

    @staticmethod
    def _stack(graphs, key, device):
        return torch.as_tensor(np.stack([np.asarray(g.graph[key], dtype=np.float32) for g in graphs]), device=device)

    def forward(self, a, p, n):
        """a, p, n: objects with ``.graph`` = {'adj','feats','num_nodes','assign_feats'} as cross_val.split_train_val
        prepares them (cross_val.py:158-184)."""
        dev = next(self.model.parameters()).device
        trip = (a, p, n)
        if RESIDENT and dev.type == "cuda":
            g, x, xa, sizes = _assemble([_resident(t, dev, self._resident) for t in trip], dev, self._resident)
            prev = getattr(self.model, "per_graph_bn", False)
            self.model.per_graph_bn = True
            try:
                out, embed = self.model(x, g, sizes, assign_x=x if xa is None else xa)
            finally:
                self.model.per_graph_bn = prev
            embed_a, embed_p, embed_n = embed[0:1], embed[1:2], embed[2:3]
            return (F.pairwise_distance(embed_a, embed_p, 2), F.pairwise_distance(embed_a, embed_n, 2), embed_a, embed_p, embed_n)
        adj = self._stack(trip, "adj", dev)
        h0 = self._stack(trip, "feats", dev)
        assign = self._stack(trip, "assign_feats", dev)
        sizes = np.array([int(g.graph["num_nodes"]) for g in trip])
        prev = getattr(self.model, "per_graph_bn", False)
        self.model.per_graph_bn = True
        try:
            out, embed = self.model(h0, adj, sizes, assign_x=assign if assign.shape == h0.shape and not torch.equal(assign, h0) else h0)
        finally:
            self.model.per_graph_bn = prev
        embed_a, embed_p, embed_n = embed[0:1], embed[1:2], embed[2:3]
        dist_p = F.pairwise_distance(embed_a, embed_p, 2)
        dist_n = F.pairwise_distance(embed_a, embed_n, 2)
        return dist_p, dist_n, embed_a, embed_p, embed_n
