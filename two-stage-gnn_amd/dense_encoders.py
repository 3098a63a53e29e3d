"""Drop-in modules for the reference's dense-batched encoders (Code/sage+gat+diffpool/encoders.py).

Same class names, constructor signatures, parameter names/shapes (``weight[Fin,Fout]``, ``bias[Fout]``,
``conv_first`` / ``conv_block`` / ``conv_last`` / ``pred_model`` …, so ``state_dict``s interchange) and call
signatures as the reference, so its scaffolding (train.py:253-260,121; tripletnet.py:36-38) can construct
and call them unchanged.  Internally nothing is dense: the adjacency becomes a CSR ``GraphBatch`` and every
tensor op on the message-passing path is a HIP kernel behind include/tsgnn.h.

Accepted inputs of ``forward(x, adj, batch_num_nodes)``:
  * the reference's tensors ``x[B,Nmax,F]``, ``adj[B,Nmax,Nmax]`` (dense -> CSR conversion on the GPU), or
  * a prebuilt ``GraphBatch`` as ``adj`` with ``x`` either padded ``[B,Nmax,F]`` or already in rows.
"""
import os
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import message_passing as mp
from .graph import GraphBatch


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


# dense adj tensor -> GraphBatch cache (the three conv layers of an encoder share one conversion)
_csr_cache = {}


def _batch_from_dense(adj, sizes, layout):
    key = (adj.data_ptr(), tuple(adj.shape), adj._version, layout,
           None if sizes is None else tuple(int(s) for s in np.asarray(sizes).reshape(-1)))
    hit = _csr_cache.get(key)
    if hit is not None and hit[0]() is adj:
        return hit[1]
    g = GraphBatch.from_dense(adj.detach(), sizes=sizes, layout=layout)
    if len(_csr_cache) > 8:
        _csr_cache.clear()
    _csr_cache[key] = (weakref.ref(adj), g)
    return g


FUSED_HEAD = True              # the two chained nn.Linear after the readout as one HIP launch (+1 backward)
FUSED_DENSE_POST = True       # pooled DiffPool levels: transform + normalise + ReLU + slot BN as one node
FUSED_DENSE_STACK = True    # pooled DiffPool levels: the whole GCN stack as one autograd node (dense_stack.py)
READOUT_PASS = True            # DiffPool: readout backward and the contraction's gradient of the same embeddings in one pass
READOUT_IN_CONTRACT = os.environ.get("TSGNN_READOUT_IN_CONTRACT", "1") != "0"   # ... and inside the contraction's node (no backward pass of its own)
SOFTMAX_IN_CONTRACT = os.environ.get("TSGNN_SOFTMAX_IN_CONTRACT", "1") != "0"   # pooled levels: the assignment softmax inside the contraction's launches
READOUT_COLUMNS = os.environ.get("TSGNN_READOUT_COLUMNS", "1") != "0"   # DiffPool: the levels' readouts written into one buffer (no cat)
FUSED_STACK = True             # GcnEncoderGraph: run the conv stack as one fused autograd node when it qualifies
PER_GRAPH_STACK = os.environ.get("TSGNN_PER_GRAPH_STACK", "1") != "0"   # ... also with per-graph statistics (the triplet step)
DENSE_ADJ_MAX_NODES = 128      # at or below this many nodes per graph a dense batched MFMA product is used


class _DenseBmm(torch.autograd.Function):
    """y[b] = adj[b] @ x[b] (+x) for small / differentiable dense adjacencies (pooled DiffPool levels,
    encoders.py:375-380): batched fp32 MFMA, gradients to BOTH operands."""

    @staticmethod
    def forward(ctx, adj, x, add_self):
        adj = adj.contiguous()
        x = x.contiguous()
        B, N, Fd = x.shape
        y = x.clone() if add_self else torch.empty_like(x)
        mp.gemm(adj, N, 1, x, Fd, 1, y, Fd, 1, N, Fd, N, batch=B, stride_a=N * N, stride_b=N * Fd, stride_c=N * Fd,
                accumulate=add_self)
        ctx.save_for_backward(adj, x)
        ctx.add_self = add_self
        return y

    @staticmethod
    def backward(ctx, dy):
        adj, x = ctx.saved_tensors
        dy = dy.contiguous()
        B, N, Fd = x.shape
        dadj = dx = None
        if ctx.needs_input_grad[0]:      # dA[b] = dY[b] x[b]^T
            dadj = torch.empty_like(adj)
            mp.gemm(dy, Fd, 1, x, 1, Fd, dadj, N, 1, N, N, Fd, batch=B, stride_a=N * Fd, stride_b=N * Fd, stride_c=N * N)
        if ctx.needs_input_grad[1]:      # dx[b] = A[b]^T dY[b] (+dY)
            dx = dy.clone() if ctx.add_self else torch.empty_like(x)
            mp.gemm(adj, 1, N, dy, Fd, 1, dx, Fd, 1, N, Fd, N, batch=B, stride_a=N * N, stride_b=N * Fd, stride_c=N * Fd,
                    accumulate=ctx.add_self)
        return dadj, dx, None


class GraphConv(nn.Module):
    """Drop-in for encoders.py:13-42: y = normalize((adj@x [+x]) @ weight + bias)."""

    def __init__(self, input_dim, output_dim, add_self=False, normalize_embedding=False, dropout=0.0, bias=True):
        super().__init__()
        self.add_self = add_self
        self.dropout = dropout
        if dropout > 0.001:
            self.dropout_layer = nn.Dropout(p=dropout)
        self.normalize_embedding = normalize_embedding
        self.input_dim = input_dim
        self.output_dim = output_dim
        dev = _default_device()
        self.weight = nn.Parameter(torch.empty(input_dim, output_dim, device=dev))
        self.bias = nn.Parameter(torch.empty(output_dim, device=dev)) if bias else None
        # the reference leaves the storage uninitialised until its encoder re-initialises it
        # (encoders.py:88-92); give a stand-alone layer the same Xavier/zero init
        nn.init.xavier_uniform_(self.weight.data, gain=nn.init.calculate_gain("relu"))
        if self.bias is not None:
            nn.init.constant_(self.bias.data, 0.0)

    # rows in, rows out (the encoders' fast path)
    def forward_rows(self, x, g):
        if self.dropout > 0.001:
            x = self.dropout_layer(x)
        z = mp.aggregate(x, g, add_self=self.add_self)
        return mp.linear_l2norm(z, self.weight, self.bias, normalize=self.normalize_embedding)

    def forward(self, x, adj):
        if isinstance(adj, GraphBatch):
            if x.dim() == 2:
                return self.forward_rows(x, adj)
            return mp.unpack_rows(self.forward_rows(mp.pack_rows(x, adj), adj), adj)
        if x.dim() != 3 or adj.dim() != 3:
            raise ValueError("GraphConv expects x[B,N,F] and adj[B,N,N]")
        B, N, Fin = x.shape
        if adj.requires_grad or N <= DENSE_ADJ_MAX_NODES:
            if self.dropout > 0.001:
                x = self.dropout_layer(x)
            y = _DenseBmm.apply(adj.float(), x.float(), self.add_self)
            v = mp.linear_l2norm(y.reshape(B * N, Fin), self.weight, self.bias, normalize=self.normalize_embedding)
            return v.reshape(B, N, self.output_dim)
        g = _batch_from_dense(adj, None, "padded")
        return self.forward_rows(x.contiguous().float().reshape(B * N, Fin), g).reshape(B, N, self.output_dim)


class GcnEncoderGraph(nn.Module):
    """Drop-in for encoders.py:45-229 (the repo's "GraphSage"/base encoder)."""

    def __init__(self, input_dim, hidden_dim, embedding_dim, label_dim, num_layers, pred_hidden_dims=[],
                 concat=True, bn=True, dropout=0.0, args=None, final_dim="output_dim"):
        super().__init__()
        self.concat = concat
        add_self = not concat
        self.bn = bn
        self.num_layers = num_layers
        self.num_aggs = 1
        self.final_dim = final_dim
        self.bias = True
        if args is not None:
            self.bias = args.bias
        self.conv_first, self.conv_block, self.conv_last = self.build_conv_layers(
            input_dim, hidden_dim, embedding_dim, num_layers, add_self, normalize=True, dropout=dropout)
        self.act = nn.ReLU()
        self.label_dim = label_dim
        self.pred_input_dim = hidden_dim * (num_layers - 1) + embedding_dim if concat else embedding_dim
        self.pre_pred_model = self.build_pred_layers(self.pred_input_dim, pred_hidden_dims, embedding_dim,
                                                     num_aggs=self.num_aggs)
        self.pred_model = self.build_pred_layers(embedding_dim, pred_hidden_dims, label_dim, num_aggs=self.num_aggs)
        self.map_model = self.build_pred_layers(self.pred_input_dim, pred_hidden_dims, embedding_dim,
                                                num_aggs=self.num_aggs)
        self.map2_model = self.build_pred_layers(pred_input_dim=embedding_dim, pred_hidden_dims=[], label_dim=2)
        # True: each graph of a batch is normalised with the statistics it would have at B = 1 (triplet settings)
        self.per_graph_bn = False
        self._init_convs()
        self.to(_default_device())

    def _init_convs(self):
        for m in self.modules():
            if isinstance(m, GraphConv):
                nn.init.xavier_uniform_(m.weight.data, gain=nn.init.calculate_gain("relu"))
                if m.bias is not None:
                    nn.init.constant_(m.bias.data, 0.0)

    def build_conv_layers(self, input_dim, hidden_dim, embedding_dim, num_layers, add_self, normalize=False,
                          dropout=0.0):
        conv_first = GraphConv(input_dim=input_dim, output_dim=hidden_dim, add_self=add_self,
                               normalize_embedding=normalize, bias=self.bias)
        conv_block = nn.ModuleList(
            [GraphConv(input_dim=hidden_dim, output_dim=hidden_dim, add_self=add_self, normalize_embedding=normalize,
                       dropout=dropout, bias=self.bias) for _ in range(num_layers - 2)])
        conv_last = GraphConv(input_dim=hidden_dim, output_dim=embedding_dim, add_self=add_self,
                              normalize_embedding=normalize, bias=self.bias)
        return conv_first, conv_block, conv_last

    def build_pred_layers(self, pred_input_dim, pred_hidden_dims, label_dim, num_aggs=1):
        pred_input_dim = pred_input_dim * num_aggs
        if len(pred_hidden_dims) == 0:
            return nn.Linear(pred_input_dim, label_dim)
        layers = []
        for pred_dim in pred_hidden_dims:
            layers.append(nn.Linear(pred_input_dim, pred_dim))
            layers.append(self.act)
            pred_input_dim = pred_dim
        layers.append(nn.Linear(pred_dim, label_dim))
        return nn.Sequential(*layers)

    def construct_mask(self, max_nodes, batch_num_nodes):
        """[B, max_nodes, 1] prefix mask (encoders.py:121-132), built without a python loop."""
        n = torch.as_tensor(np.asarray(batch_num_nodes, dtype=np.int64), device=_default_device())
        return (torch.arange(max_nodes, device=n.device)[None, :] < n[:, None]).float().unsqueeze(2)

    def apply_bn(self, x):
        """Per-node-slot batch norm of a padded [B,N,F] tensor (encoders.py:134-138)."""
        B, N, Fd = x.shape
        g = GraphBatch.uniform(B, N, x.device)
        return mp.bn_slots(x.contiguous().float().reshape(B * N, Fd), g, relu=False, bn=True).reshape(B, N, Fd)

    # ------------------------------------------------------------------ graph batch plumbing
    @staticmethod
    def make_batch(x, adj, batch_num_nodes):
        """-> (rows, GraphBatch).  With node counts: packed rows + ghost-slot rows; without: padded rows."""
        if isinstance(adj, GraphBatch):
            g = adj
        elif batch_num_nodes is not None:
            g = _batch_from_dense(adj, np.asarray(batch_num_nodes).reshape(-1), "packed")
        else:
            g = _batch_from_dense(adj, None, "padded")
        if x.dim() == 3:
            F_in = x.size(2)
            ld = (F_in + 3) // 4 * 4 if g.layout == "packed" else None      # 16-B rows for the float4 gather
            x = mp.pack_rows(x, g, ld)
        return x, g

    def _post(self, v, g):
        """ReLU then (optionally) slot batch-norm: encoders.py:179-181."""
        return mp.bn_slots(v, g, relu=True, bn=self.bn, per_graph=self.per_graph_bn)

    def gcn_forward_rows(self, x, g, conv_first, conv_block, conv_last, mask_ghost=False):
        """gcn_forward (encoders.py:140-167) on rows: per-layer outputs concatenated on the feature axis;
        ``mask_ghost`` = multiply by the embedding mask (zeroes every ghost row)."""
        if FUSED_STACK and not self.per_graph_bn and (mask_ghost or not g.n_ghost):
            from . import sage_stack
            convs = [conv_first] + list(conv_block) + [conv_last]
            if (sage_stack.eligible(g, convs, self.bn, x)
                    and bool(mp.nat.lib().tsgnn_slot_fused_supported(g.B, convs[0].output_dim))
                    and bool(mp.nat.lib().tsgnn_slot_fused_supported(g.B, convs[-1].output_dim))):
                return sage_stack.sage_stack_nodes(x, g, convs, mask_ghost)     # one autograd node, layers write into the cat
        x = self._post(conv_first.forward_rows(x, g), g)
        x_all = [x]
        for conv in conv_block:
            x = self._post(conv.forward_rows(x, g), g)
            x_all.append(x)
        x_all.append(conv_last.forward_rows(x, g))
        t = torch.cat(x_all, dim=1)
        if mask_ghost and g.n_ghost:
            t = mp.mask_ghost_rows(t, g)
        return t

    def readouts_rows(self, x, g):
        """encoders.py:177-205 up to the concatenated max readout."""
        from . import sage_stack
        convs = [self.conv_first] + list(self.conv_block) + [self.conv_last]
        if self.concat and FUSED_STACK and sage_stack.eligible(g, convs, self.bn, x) and \
                bool(mp.nat.lib().tsgnn_slot_fused_supported(g.B, convs[0].output_dim)) and \
                bool(mp.nat.lib().tsgnn_slot_fused_supported(g.B, convs[-1].output_dim)) and \
                (not self.per_graph_bn or (PER_GRAPH_STACK and g.row_graph is not None and g.n_ghost == g.nmax
                                           and convs[0].output_dim <= 256)):
            # (per-graph statistics, the triplet step: the same node with the row-local batch-norm launches, sage_stack.per_graph_stats)
            with sage_stack.per_graph_stats(self.per_graph_bn):
                return sage_stack.sage_stack_readouts(x, g, convs)
        # the layers' readouts land in their column blocks of ONE buffer (no torch.cat launch, no slice copies backwards)
        cols = mp.ReadoutColumns(g.B, self.pred_input_dim, x.device) if (self.concat and READOUT_COLUMNS) else None
        x = self._post(self.conv_first.forward_rows(x, g), g)
        out_all = [mp.readout_max(x, g, into=cols and cols.take(x.size(1)))]
        for conv in self.conv_block:
            x = self._post(conv.forward_rows(x, g), g)
            out_all.append(mp.readout_max(x, g, into=cols and cols.take(x.size(1))))
        x = self.conv_last.forward_rows(x, g)
        out_all.append(mp.readout_max(x, g, into=cols and cols.take(x.size(1))))
        if not self.concat:
            return out_all[-1]
        return cols.join(out_all) if cols is not None else torch.cat(out_all, dim=1)

    def _head_linears(self):
        """the two chained nn.Linear after the readout (None for the 2stg setting, whose first output IS the readout)"""
        if getattr(self, "_defer_map", False):
            return None
        if self.final_dim == "pretrain":
            return self.map_model, self.map2_model
        if self.final_dim != "output_dim":
            return self.pre_pred_model, self.pred_model
        return None

    def _heads(self, output):
        if self.final_dim == "pretrain":          # 2stg+
            if getattr(self, "_defer_map", False):
                return None, output                 # (tripletnet reads the embedding only; it applies map_model to `output` itself)
            if FUSED_HEAD and mp.head2_ok(output, self.map_model, self.map2_model):
                out, ypred = mp.head2(output, self.map_model, self.map2_model)
                return ypred, out
            out = self.map_model(output)
            return self.map2_model(out), out
        if self.final_dim != "output_dim":        # original
            if FUSED_HEAD and mp.head2_ok(output, self.pre_pred_model, self.pred_model):
                return mp.head2(output, self.pre_pred_model, self.pred_model)
            vec = self.pre_pred_model(output)
            return vec, self.pred_model(vec)
        if getattr(self, "_defer_map", False):    # triplet.tripletnet applies map_model itself (embeddings + distances, one launch)
            return output, None
        return output, self.map_model(output)     # 2stg

    def _stack_fusable(self, x, g):
        from . import sage_stack
        convs = [self.conv_first] + list(self.conv_block) + [self.conv_last]
        ok = (self.concat and FUSED_STACK and not self.per_graph_bn and sage_stack.eligible(g, convs, self.bn, x)
              and bool(mp.nat.lib().tsgnn_slot_fused_supported(g.B, convs[0].output_dim))
              and bool(mp.nat.lib().tsgnn_slot_fused_supported(g.B, convs[-1].output_dim)))
        return ok, convs

    def forward(self, x, adj, batch_num_nodes=None, **kwargs):
        x, g = self.make_batch(x, adj, batch_num_nodes)
        if FUSED_HEAD and type(self) is GcnEncoderGraph and self.final_dim != "output_dim":
            from . import sage_stack
            ok, convs = self._stack_fusable(x, g)
            lin1, lin2 = (self.map_model, self.map2_model) if self.final_dim == "pretrain" else (self.pre_pred_model, self.pred_model)
            if ok and sage_stack.head_ok(g, convs, lin1, lin2):
                vec, y = sage_stack.sage_stack_head(x, g, convs, lin1, lin2)       # stack + readout tail + head: one autograd node
                return (y, vec) if self.final_dim == "pretrain" else (vec, y)
        return self._heads(self.readouts_rows(x, g))

    def loss(self, pred, label, type="softmax"):
        if type == "softmax":
            return mp.cross_entropy(pred, label)
        if type == "margin":
            onehot = torch.zeros(pred.size(0), self.label_dim, dtype=torch.long, device=pred.device)
            onehot.scatter_(1, label.view(-1, 1), 1)
            return torch.nn.MultiLabelMarginLoss()(pred, onehot)
        raise ValueError(type)


class SoftPoolingGcnEncoder(GcnEncoderGraph):
    """Drop-in for encoders.py:236-441 (DiffPool).  Level 0 runs on the CSR rows of the input graphs; the
    assignment contraction X' = S^T Z, A' = S^T A S (encoders.py:374-375) is SpMM + ragged batched fp32-MFMA
    GEMMs; pooled levels (dense, differentiable adjacency) use strided batched MFMA GEMMs."""

    def __init__(self, max_num_nodes, input_dim, hidden_dim, embedding_dim, label_dim, num_layers, assign_hidden_dim,
                 assign_ratio=0.25, assign_num_layers=-1, num_pooling=1, pred_hidden_dims=[], concat=True, bn=True,
                 dropout=0.0, linkpred=True, assign_input_dim=-1, args=None, final_dim="output_dim"):
        # the reference forwards neither bn nor dropout to the base ctor (encoders.py:249-250): bn stays True
        super().__init__(input_dim, hidden_dim, embedding_dim, label_dim, num_layers, pred_hidden_dims=pred_hidden_dims,
                         concat=concat, args=args)
        add_self = not concat
        self.num_pooling = num_pooling
        self.linkpred = linkpred
        self.assign_ent = True
        self.final_dim = final_dim
        self.conv_first_after_pool = nn.ModuleList()
        self.conv_block_after_pool = nn.ModuleList()
        self.conv_last_after_pool = nn.ModuleList()
        for _ in range(num_pooling):
            c1, cb, cl = self.build_conv_layers(self.pred_input_dim, hidden_dim, embedding_dim, num_layers, add_self,
                                                normalize=True, dropout=dropout)
            self.conv_first_after_pool.append(c1)
            self.conv_block_after_pool.append(cb)
            self.conv_last_after_pool.append(cl)
        if assign_num_layers == -1:
            assign_num_layers = num_layers
        if assign_input_dim == -1:
            assign_input_dim = input_dim
        self.assign_conv_first_modules = nn.ModuleList()
        self.assign_conv_block_modules = nn.ModuleList()
        self.assign_conv_last_modules = nn.ModuleList()
        self.assign_pred_modules = nn.ModuleList()
        assign_dim = int(max_num_nodes * assign_ratio)
        for _ in range(num_pooling):
            a1, ab, al = self.build_conv_layers(assign_input_dim, assign_hidden_dim, assign_dim, assign_num_layers, add_self,
                                                normalize=True)
            assign_pred_input_dim = assign_hidden_dim * (num_layers - 1) + assign_dim if concat else assign_dim
            self.assign_pred_modules.append(self.build_pred_layers(assign_pred_input_dim, [], assign_dim, num_aggs=1))
            assign_input_dim = self.pred_input_dim
            assign_dim = int(assign_dim * assign_ratio)
            self.assign_conv_first_modules.append(a1)
            self.assign_conv_block_modules.append(ab)
            self.assign_conv_last_modules.append(al)
        self.pre_pred_model = self.build_pred_layers(self.pred_input_dim * (num_pooling + 1), pred_hidden_dims, embedding_dim,
                                                     num_aggs=self.num_aggs)
        self.pred_model = self.build_pred_layers(embedding_dim, pred_hidden_dims, label_dim, num_aggs=self.num_aggs)
        self.map_model = self.build_pred_layers(self.pred_input_dim * (num_pooling + 1), pred_hidden_dims, embedding_dim,
                                                num_aggs=self.num_aggs)
        self.map2_model = self.build_pred_layers(pred_input_dim=embedding_dim, pred_hidden_dims=[], label_dim=2)
        self._init_convs()
        self.to(_default_device())

    # gcn_forward on dense pooled tensors x[B,K,F], adj[B,K,K] (encoders.py:378-380; mask is None there)
    def gcn_forward_dense(self, x, adj, conv_first, conv_block, conv_last):
        B, K, _ = x.shape
        g = GraphBatch.uniform(B, K, x.device)
        if FUSED_DENSE_STACK:
            from . import dense_stack
            convs = [conv_first] + list(conv_block) + [conv_last]
            if dense_stack.eligible(x, adj, g, convs, self.bn, self.per_graph_bn):
                return dense_stack.dense_gcn_stack(x, adj, g, convs), g       # one autograd node, layers write into the cat

        def post(v):
            return mp.bn_slots(v.reshape(B * K, -1), g, relu=True, bn=self.bn, per_graph=self.per_graph_bn).reshape(B, K, -1)

        def hidden(conv, x):
            # transform + normalise + ReLU + slot BN as one node (one launch for the BN forward, one for the backward of
            # BN + ReLU + normalise) when the fused slot kernels take the shape; else the composed ops
            if (FUSED_DENSE_POST and self.bn and not self.per_graph_bn and conv.normalize_embedding and conv.dropout <= 0.001
                    and mp.linear_norm_bn_ok(g, conv.output_dim)):
                y = _DenseBmm.apply(adj.float(), x.float(), conv.add_self)
                return mp.linear_norm_bn(y.reshape(B * K, -1), conv.weight, conv.bias, g).reshape(B, K, -1)
            return post(conv(x, adj))
        x = hidden(conv_first, x)
        x_all = [x]
        for conv in conv_block:
            x = hidden(conv, x)
            x_all.append(x)
        x_all.append(conv_last(x, adj))
        return torch.cat(x_all, dim=2), g

    def _level0_pair_ok(self, x, x_a, g, masked):
        """both first-level stacks take the fused stack path (gcn_forward_rows) and may share launches"""
        from . import sage_stack
        if not (FUSED_STACK and sage_stack.PAIR_LAUNCHES and not self.per_graph_bn and (masked or not g.n_ghost)):
            return False
        lib = mp.nat.lib()
        for xx, convs in ((x, [self.conv_first] + list(self.conv_block) + [self.conv_last]),
                          (x_a, [self.assign_conv_first_modules[0]] + list(self.assign_conv_block_modules[0])
                           + [self.assign_conv_last_modules[0]])):
            if not (torch.is_tensor(xx) and xx.dim() == 2 and sage_stack.eligible(g, convs, self.bn, xx)
                    and bool(lib.tsgnn_slot_fused_supported(g.B, convs[0].output_dim))
                    and bool(lib.tsgnn_slot_fused_supported(g.B, convs[-1].output_dim))):
                return False
        return True

    def forward(self, x, adj, batch_num_nodes, **kwargs):
        from . import diffpool as dp
        from .pyg import linear
        x_a = kwargs["assign_x"] if "assign_x" in kwargs else x
        same_input = x_a is x
        masked = batch_num_nodes is not None
        x, g = self.make_batch(x, adj, batch_num_nodes)             # packed rows (+ghost reps) if masked, else padded
        x_a = x if same_input else (mp.pack_rows(x_a, g, (x_a.size(2) + 3) // 4 * 4 if g.layout == "packed" else None)
                                    if x_a.dim() == 3 else x_a)
        a0 = None
        if self.num_pooling > 0 and self._level0_pair_ok(x, x_a, g, masked):
            # the embedding and the first assignment stack run on the same graph: one autograd node whose launches are shared
            # pairwise (sage_stack._SageStackPair) — each kernel of one stack alone fills about half of the CUs
            from . import sage_stack
            emb, a0 = sage_stack.sage_stack_nodes_pair(
                x, x_a, g, [self.conv_first] + list(self.conv_block) + [self.conv_last],
                [self.assign_conv_first_modules[0]] + list(self.assign_conv_block_modules[0]) + [self.assign_conv_last_modules[0]],
                masked)
        else:
            emb = self.gcn_forward_rows(x, g, self.conv_first, self.conv_block, self.conv_last, mask_ghost=masked)
        # the levels' readouts land in their column blocks of ONE buffer (no torch.cat launch; mp.ReadoutColumns)
        cols = mp.ReadoutColumns(g.B, emb.size(1) * (self.num_pooling + 1), emb.device) if (self.concat and READOUT_COLUMNS) else None
        ro_next = None              # (READOUT_IN_CONTRACT) the readout of this level's embeddings, made by the contraction that reads them
        if self.num_pooling > 0 and READOUT_PASS and READOUT_IN_CONTRACT:
            # the embeddings feed the readout AND the contraction: the readout is a third output of the contraction's node and its
            # gradient is added inside the launch that produces the embeddings' gradient (diffpool._ContractRows / _ContractDense);
            # masked: the ghost rows of `emb` are constants (zeros), their gradient is discarded by the stack's backward
            ro_next = (bool(masked and g.n_ghost), cols and cols.take(emb.size(1)), bool(masked and g.n_ghost))
            out_all = []
        elif self.num_pooling > 0 and READOUT_PASS:
            # ... or a node of its own that passes the embeddings through: one backward pass sums both gradients (mp._ReadoutMax)
            ro, emb = mp.readout_max_pass(emb, g, ghost_unused=bool(masked and g.n_ghost), into=cols and cols.take(emb.size(1)))
            out_all = [ro]
        else:
            out_all = [mp.readout_max(emb, g, into=cols and cols.take(emb.size(1)))]
        dense_x = dense_adj = None
        a_next = None
        for i in range(self.num_pooling):
            lin = self.assign_pred_modules[i]
            if i == 0:
                a = a0 if a0 is not None else self.gcn_forward_rows(
                    x_a, g, self.assign_conv_first_modules[0], self.assign_conv_block_modules[0],
                    self.assign_conv_last_modules[0], mask_ghost=masked)
                s = dp.row_softmax(mp.linear_oi(a, lin.weight, lin.bias),                    # encoders.py:369
                                   g.n_rows if (masked and g.n_ghost) else None)              # :371 (ghost rows -> 0)
                self.assign_tensor = s
                self._link_graph, self._link_masked = g, masked
                if ro_next is not None:
                    dense_x, dense_adj, ro = dp.diffpool_contract_rows(s, emb, g, readout=ro_next)     # :374-375 (+ :353)
                    out_all.append(ro)
                    ro_next = None
                else:
                    dense_x, dense_adj = dp.diffpool_contract_rows(s, emb, g)                 # :374-375
            else:
                if a_next is not None:
                    a = a_next                  # came out of the launch that ran this level's embedding stack
                else:
                    a, _ = self.gcn_forward_dense(dense_x, dense_adj, self.assign_conv_first_modules[i],
                                                  self.assign_conv_block_modules[i], self.assign_conv_last_modules[i])
                Bq, Kq, Cq = a.shape
                logits = mp.linear_oi(a.reshape(Bq * Kq, Cq), lin.weight, lin.bias)
                if SOFTMAX_IN_CONTRACT:
                    # nn.Softmax(dim=-1) (:369) on the operand the contraction stages anyway, its backward in the backward launch
                    res = dp.diffpool_contract_dense(logits.reshape(Bq, Kq, -1), emb_dense, dense_adj, readout=ro_next, softmax=True)
                    s = res[-1]
                else:
                    s = dp.row_softmax(logits).reshape(Bq, Kq, -1)
                    res = dp.diffpool_contract_dense(s, emb_dense, dense_adj, readout=ro_next)
                self.assign_tensor = s
                dense_x, dense_adj = res[0], res[1]
                if ro_next is not None:
                    out_all.append(res[2])
                    ro_next = None
            a_next = None
            emb_convs = [self.conv_first_after_pool[i]] + list(self.conv_block_after_pool[i]) + [self.conv_last_after_pool[i]]
            if i + 1 < self.num_pooling and FUSED_DENSE_STACK and self.bn and not self.per_graph_bn:
                # this level's embedding stack and the NEXT level's assignment stack read the same (x, adjacency): one launch
                # forward and one backward for both (dense_stack.py)
                from . import dense_stack
                asg_convs = ([self.assign_conv_first_modules[i + 1]] + list(self.assign_conv_block_modules[i + 1])
                             + [self.assign_conv_last_modules[i + 1]])
                if dense_stack.ONE_LAUNCH and dense_stack.one_launch_ok(dense_x, dense_adj, [emb_convs, asg_convs]):
                    # (the next contraction reads the adjacency through this node's pass-through: one gradient sum, inside
                    # the stacks' backward launch)
                    emb_dense, a_next, dense_adj = dense_stack.dense_gcn_stacks(dense_x, dense_adj, [emb_convs, asg_convs],
                                                                                adj_pass=True)
                    gd = GraphBatch.uniform(dense_x.size(0), dense_x.size(1), dense_x.device)
            if a_next is None:
                emb_dense, gd = self.gcn_forward_dense(dense_x, dense_adj, self.conv_first_after_pool[i],
                                                       self.conv_block_after_pool[i], self.conv_last_after_pool[i])
            Bq, Kq, Cq = emb_dense.shape
            if i + 1 < self.num_pooling and READOUT_PASS and READOUT_IN_CONTRACT:
                ro_next = (gd, cols and cols.take(Cq))         # made by the next level's contraction
            elif i + 1 < self.num_pooling and READOUT_PASS:
                ro, e2 = mp.readout_max_pass(emb_dense.reshape(Bq * Kq, Cq), gd, into=cols and cols.take(Cq))
                emb_dense = e2.reshape(Bq, Kq, Cq)
                out_all.append(ro)
            else:
                lins = self._head_linears()
                if (i + 1 == self.num_pooling and self.concat and lins is not None and FUSED_HEAD and gd.n_ghost == 0
                        and mp.head2_tail_ok(cols, emb_dense, Kq, *lins)):
                    # the LAST level's readout fills its column block inside the head's own launches (mp._Head2Tail)
                    a, b = mp.head2_tail(cols, out_all, emb_dense.reshape(Bq * Kq, Cq), Kq, *lins)
                    return (b, a) if self.final_dim == "pretrain" else (a, b)
                out_all.append(mp.readout_max(emb_dense.reshape(Bq * Kq, Cq), gd, into=cols and cols.take(Cq)))
        if not self.concat:
            output = out_all[-1]
        else:
            output = cols.join(out_all) if cols is not None else torch.cat(out_all, dim=1)
        return self._heads(output)

    linkpred_clamp = 1.0        # the reference clamps pred_adj with an UNINITIALISED tensor (encoders.py:424); see DESIGN.md

    def loss(self, pred, label, adj=None, batch_num_nodes=None, adj_hop=1):
        """CE (+ the link-prediction side loss of encoders.py:416-440 when ``linkpred``).  ``adj`` / ``batch_num_nodes`` are
        accepted for signature compatibility; the adjacency and the mask used are those of the forward's batch."""
        loss = super().loss(pred, label)
        if self.linkpred:
            from . import diffpool as dp
            if self.num_pooling != 1:
                # trap T7: the reference multiplies the LAST level's assignment with the ORIGINAL adjacency (:418,428)
                raise RuntimeError("link-prediction loss with num_pooling >= 2: the last assignment tensor [B, %d, .] does not "
                                   "match the input adjacency (the reference fails with a shape error here too)"
                                   % self.assign_tensor.size(-2))
            self.link_loss = dp.link_pred_loss(self.assign_tensor, self._link_graph, self.linkpred_clamp,
                                               masked=self._link_masked, adj_hop=adj_hop)
            return loss + self.link_loss
        return loss
