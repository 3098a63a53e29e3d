"""Drop-in for Code/sage+gat+diffpool/tripletnet.py:11-45 — the 2stg / 2stg+ training step.

The reference runs the encoder three times at B = 1 (anchor, positive, negative: three uploads of a dense
[1,Nmax,Nmax] adjacency and three sequential forwards, tripletnet.py:18-38).  Here the three graphs form ONE
block-diagonal GraphBatch and go through the kernels once; ``per_graph_bn`` keeps every graph on the per-row statistics it
would have alone in its batch, so embeddings and gradients equal the three separate B = 1 calls.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .graph import GraphBatch


class tripletnet(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    @staticmethod
    def _stack(graphs, key, device):
        return torch.as_tensor(np.stack([np.asarray(g.graph[key], dtype=np.float32) for g in graphs]), device=device)

    def forward(self, a, p, n):
        """a, p, n: objects with ``.graph`` = {'adj','feats','num_nodes','assign_feats'} as cross_val.split_train_val
        prepares them (cross_val.py:158-184)."""
        dev = next(self.model.parameters()).device
        trip = (a, p, n)
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
