"""Drop-in for Code/sag/layers.py (SAGPool) and Code/sag/network.py (Net) on the HIP kernels.

Same constructor signatures, submodule names (``score_layer``, ``conv1..3``, ``pool1..3``, ``lin1..3``) and
return tuples as the reference, so ``latest.pth`` state_dicts (Code/sag/train.py:201-212) interchange.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import message_passing as mp
from . import pyg
from .pyg import GCNConv, filter_adj, global_max_pool as gmp, global_mean_pool as gap, topk


class SAGPool(nn.Module):
    def __init__(self, in_channels, ratio=0.8, Conv=GCNConv, non_linearity=torch.tanh):
        super().__init__()
        self.in_channels = in_channels
        self.ratio = ratio
        self.score_layer = Conv(in_channels, 1)
        self.non_linearity = non_linearity

    def forward(self, x, edge_index, edge_attr=None, batch=None):
        if batch is None:
            batch = edge_index.new_zeros(x.size(0))                      # layers.py:15-16
        score = self.score_layer(x, edge_index).view(-1)                 # :18  (squeeze)
        perm = topk(score, self.ratio, batch)                            # :20
        if self.non_linearity is torch.tanh:
            x = pyg.gather_gate(x, score, perm, use_tanh=True)           # :21 fused gather * tanh gate
        else:
            x = x[perm] * self.non_linearity(score[perm]).view(-1, 1)
        batch = batch[perm]                                              # :22
        edge_index, edge_attr = filter_adj(edge_index, edge_attr, perm, num_nodes=score.size(0))   # :23-24
        return x, edge_index, edge_attr, batch, perm


class Net(nn.Module):
    """Code/sag/network.py:9-53.  ``use_batch=False`` reproduces the reference, which throws data.batch away
    (network.py:32, trap T6: a mini-batch is pooled as ONE graph); ``use_batch=True`` gives PyG's per-graph
    semantics.  ``fused=True`` (default) runs the three conv -> pool -> readout levels as the sync-free ``sag_stack`` node
    (hipGraph-capturable); ``fused=False`` composes the PyG-named drop-ins level by level as the reference does (three
    host round trips per level for the data-dependent tensor sizes)."""

    def __init__(self, num_features, nhid, num_classes, pooling_ratio, dropout_ratio, use_batch=False, fused=True, conv="gcn"):
        """conv = "gcn": the reference's network (network.py:19-23).  conv = "sage": the same network with its conv layers replaced by PyG
        SAGEConv — BASELINE config 4 as worded ("SAGPool (ratio 0.5) + SAGEConv"); the pooling layers stay the reference's SAGPool."""
        super().__init__()
        self.num_features, self.nhid, self.num_classes = num_features, nhid, num_classes
        self.pooling_ratio, self.dropout_ratio, self.use_batch = pooling_ratio, dropout_ratio, use_batch
        self.fused = fused
        self.conv_kind = conv
        Conv = GCNConv if conv == "gcn" else pyg.SAGEConv
        self.conv1 = Conv(self.num_features, self.nhid)
        self.pool1 = SAGPool(self.nhid, ratio=self.pooling_ratio)
        self.conv2 = Conv(self.nhid, self.nhid)
        self.pool2 = SAGPool(self.nhid, ratio=self.pooling_ratio)
        self.conv3 = Conv(self.nhid, self.nhid)
        self.pool3 = SAGPool(self.nhid, ratio=self.pooling_ratio)
        dev = pyg._default_device()
        self.lin1 = nn.Linear(self.nhid * 2, self.nhid).to(dev)
        self.lin2 = nn.Linear(self.nhid, self.nhid // 2).to(dev)
        self.lin3 = nn.Linear(self.nhid // 2, self.num_classes).to(dev)

    def _fused_ok(self):
        from . import sag_stack
        pools = (self.pool1, self.pool2, self.pool3)
        convs = (self.conv1, self.conv2, self.conv3)
        ok = (self.fused and sag_stack.supported(self.nhid) and all(p.non_linearity is torch.tanh for p in pools)
              and all(isinstance(p.score_layer, GCNConv) and p.score_layer.bias is not None for p in pools))
        if self.conv_kind == "gcn":
            return ok and all(c.bias is not None for c in convs)
        return ok and all(c.lin_l.bias is not None and c.root_weight and not c.normalize for c in convs) and self.nhid % 4 == 0

    def _forward_fused(self, data):
        from . import sag_stack
        x = data.x
        g = data.edge_index if isinstance(data.edge_index, pyg.GraphBatch) else pyg.graph_of(data.edge_index, x.size(0), check_symmetry=True)
        batch = getattr(data, "batch", None) if self.use_batch else None
        sizes = pyg.segment_sizes(batch, x.size(0))
        plan = sag_stack.SagPlan.get(sizes, self.pooling_ratio, x.device, depth=3)
        params = []
        if self.conv_kind != "gcn":
            if g.symmetric and plan.levels[0].max_seg <= int(mp.nat.lib().tsgnn_sag_pool_graph_max_nodes()):
                from . import sag_stack_sage
                for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv3, self.pool3)):
                    params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight, pool.score_layer.weight, pool.score_layer.bias]
                return sag_stack_sage.sag_sage_stack(x, g, plan, params)
            return None
        for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv3, self.pool3)):
            params += [conv.weight, conv.bias, pool.score_layer.weight, pool.score_layer.bias]
        return sag_stack.sag_stack(x, g, plan, params)

    def forward(self, data):
        x = self._forward_fused(data) if self._fused_ok() else None      # network.py:33-46 in one node
        if x is not None:
            if mp.mlp3_ok(x, self.lin1, self.lin2, self.lin3):
                return mp.mlp3_log_softmax(x, self.lin1, self.lin2, self.lin3, self.dropout_ratio, self.training)   # :48-53
            x = pyg.relu(mp.linear_oi(x, self.lin1.weight, self.lin1.bias))
            x = F.dropout(x, p=self.dropout_ratio, training=self.training)
            x = pyg.relu(mp.linear_oi(x, self.lin2.weight, self.lin2.bias))
            return F.log_softmax(mp.linear_oi(x, self.lin3.weight, self.lin3.bias), dim=-1)
        x, edge_index = data.x, data.edge_index
        batch = getattr(data, "batch", None) if self.use_batch else None
        outs = []
        for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv3, self.pool3)):
            x = pyg.relu(conv(x, edge_index))
            x, edge_index, _, batch, _ = pool(x, edge_index, None, batch)
            outs.append(torch.cat([gmp(x, batch), gap(x, batch)], dim=1))
        x = outs[0] + outs[1] + outs[2]
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=self.dropout_ratio, training=self.training)
        x = F.relu(self.lin2(x))
        return F.log_softmax(self.lin3(x), dim=-1)
