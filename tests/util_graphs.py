"""Seeded synthetic graph batches in the reference's dense padded layout (graph_sampler.py:102-114)."""
import numpy as np
import torch


def dense_batch(seed, B, nmax, fin, sizes=None, p_edge=0.2, weighted=False, symmetric=True):
    gen = torch.Generator().manual_seed(seed)
    if sizes is None:
        sizes = torch.randint(max(1, nmax // 4), nmax + 1, (B,), generator=gen).tolist()
    adj = torch.zeros(B, nmax, nmax)
    x = torch.zeros(B, nmax, fin)
    for b, n in enumerate(sizes):
        u = torch.rand(n, n, generator=gen)
        if symmetric:
            a = (torch.triu(u, 1) < p_edge).float() * torch.triu(torch.ones(n, n), 1)
            if weighted:
                a = a * (0.5 + torch.rand(n, n, generator=gen))
            a = a + a.t()
        else:
            a = (u < p_edge).float() * (1 - torch.eye(n))
            if weighted:
                a = a * (0.5 + torch.rand(n, n, generator=gen))
        adj[b, :n, :n] = a
        x[b, :n] = torch.randn(n, fin, generator=gen)
    return x, adj, np.asarray(sizes, dtype=np.int64)


def dd_like_sizes(seed, B, nbar=269, nmax=1000):
    g = torch.Generator().manual_seed(seed)
    n = torch.round(nbar * (1 + 0.35 * torch.randn(B, generator=g))).clamp(8, nmax).long()
    return n.numpy()
