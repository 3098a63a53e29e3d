"""BASELINE config 4 as worded — "IMDB-BINARY SAGPool (ratio 0.5) + SAGEConv h=128" — as ONE sync-free autograd node: the three
conv -> SAGPool -> readout levels of Code/sag/network.py:33-44 with the network's GCNConv layers replaced by PyG SAGEConv
(lin_l(mean_j x_j) + lin_r(x_i)); the pooling layer is the reference's own SAGPool (Code/sag/layers.py:7-25: GCNConv(C -> 1) scorer, top-k,
tanh gate, filter_adj), i.e. the per-graph kernels of sag_stack.py unchanged.

What changes against sag_stack._SagStack is the conv of a level:
    agg  = mean aggregation over the level's (filtered) CSR        tsgnn_propagate_mean_f32 (1 / deg from the row lengths: no coefficient arrays)
    y    = [agg || x] . [W_l | W_r]^T + b                           ONE row-panel product on the concatenation (K = 2 * ceil4(F_in)): the
                                                                    aggregation writes the left half of the buffer, the previous level's gated
                                                                    gather wrote the right half in place (its output stride is a parameter)
and, backward, the slabs of dW_l = agg^T dy and dW_r = x^T dy in one launch (tsgnn_sage_wgrad_pair_f32), d[agg || x] = dy . [W_l | W_r],
dx = A_mean^T dagg + dself in one launch (tsgnn_propagate_mean_f32, transpose form, self half added from its own columns); ONE reduction
at the end of the backward sums every level's slabs into nn.Linear's layout AND the score layers' per-graph partial rows, straight into
the flat gradient bucket with |grad|^2 shares when a FlatTrainer is listening.  The [W_l | W_r] images of all levels: one launch
(tsgnn_copy2d_multi_f32).  Launches per step: 1 + 3 x 3 forward, 3 + 4 + 4 + 1 backward (the GCN network of sag_stack.py: 6 and 10).  Symmetric edge lists (every TU dataset); other inputs take the
composed operators (pyg.SagePoolNet).  PARITY UNPINNED for the SAGEConv half; the SAGPool half follows layers.py:14-25."""
import numpy as np
import torch

from . import _native as nat
from . import message_passing as mp
from . import pyg_sage as ps
from . import sag_stack as ss

_f32 = mp._f32
_i32 = ss._i32


def _ceil4(k):
    return (int(k) + 3) // 4 * 4


def _prop_mean(rowptr, rowend, col, transpose, x, ldx, xself, ldxs, y, ldy, n, feat):
    nat.call("propagate_mean_f32", rowptr, rowend, col, int(transpose), x, int(ldx), xself, int(ldxs), y, int(ldy), int(n), int(feat))


def _wcats(pairs, device):
    """[W_l | W_r] of every level in nn.Linear's [out, in] layout, both halves padded to Kp = ceil4(K) columns: [H, 2 Kp] each, ONE launch"""
    outs, words = [], [2 * len(pairs)]
    for wl, wr in pairs:
        H, K = int(wl.size(0)), int(wl.size(1))
        Kp = _ceil4(K)
        o = _f32(H, 2 * Kp, device=device)
        for t, w in enumerate((wl, wr)):
            words += [w.data_ptr(), int(w.stride(0)), H, K, o.data_ptr() + 4 * t * Kp, int(o.stride(0)), Kp]
        outs.append(o)
    d = np.asarray(words, dtype=np.int64)
    nat.call("copy2d_multi_f32", d.ctypes.data)
    return outs


class _SagSageStack(torch.autograd.Function):
    """forward(x, g, plan, *[W_l, b_l, W_r, score weight, score bias] per level) -> readout [B, 2H]"""

    @staticmethod
    def forward(ctx, x, g, plan, *params):
        depth = plan.depth
        if len(params) != 5 * depth:
            raise ValueError("expected (lin_l.weight, lin_l.bias, lin_r.weight, score weight, score bias) per level")
        if plan.levels[0].N != g.total_rows or x.size(0) != g.total_rows:
            raise ValueError("plan, graph and features disagree on the number of nodes")
        if g.val is not None or not g.symmetric:
            raise NotImplementedError("the one-node SAGPool + SAGEConv stack takes unit-weight symmetric edge lists")
        dev = x.device
        H = int(params[0].size(0))
        B = plan.levels[0].B
        rowptr, col, rowend = g.rowptr, g.col, None
        dinv, self_w = ss.gcn_coef(g)                           # GCN coefficients: the SCORE layer of the pool (layers.py:18)
        nnz_bound = max(int(col.numel()), 1)
        read = _f32(B, 2 * H, device=dev)
        pool_graph_max = int(nat.lib().tsgnn_sag_pool_graph_max_nodes())
        if plan.levels[0].max_seg > pool_graph_max:
            raise NotImplementedError("graphs of more than %d nodes: use the composed operators" % pool_graph_max)
        # level 0's concatenation buffer [N, 2 Kp]: the features go to the right half (cached while x is the same resident tensor)
        K0 = int(x.size(1))
        Kp0 = _ceil4(K0)
        key = (x.data_ptr(), x._version, tuple(x.shape))
        hit = getattr(plan, "_sage_cat0", None)
        if hit is None or hit[0] != key:
            cat = torch.zeros(plan.levels[0].N, 2 * Kp0, dtype=torch.float32, device=dev)
            cat[:, Kp0:Kp0 + K0].copy_(x)
            plan._sage_cat0 = hit = (key, cat)
        cat = hit[1]
        wcats = _wcats([(params[5 * l].contiguous(), params[5 * l + 2].contiguous()) for l in range(depth)], dev)
        saved = []
        K, Kp = K0, Kp0
        for l in range(depth):
            L, Ln = plan.levels[l], plan.levels[l + 1]
            N, Kn = L.N, Ln.N
            wl, bl, wr, ws, bs = params[5 * l: 5 * l + 5]
            wcat = wcats[l]
            wsv = ss._al16(ws.contiguous().view(-1))
            bl = ss._al16(bl.contiguous())
            # agg -> the left half of the concatenation, then ONE product for both weights
            _prop_mean(rowptr, rowend, col, 0, cat[:, Kp:], cat.stride(0), None, 0, cat, cat.stride(0), N, K)
            y = _f32(N, H, device=dev)
            nat.call("rowgemm_f32", cat, cat.stride(0), wcat, wcat.stride(0), 1, bl, y, y.stride(0), None, N, 2 * Kp, H, 0, 0)
            # the level's tail (score, top-k, gated gather, readout, filter): one workgroup per graph; the kept rows land in the right half of
            # the NEXT level's concatenation
            cat_n = _f32(max(Kn, 1), 2 * H, device=dev)
            xp = cat_n[:, H:]
            perm, new_id = _i32(max(Kn, 1), device=dev), _i32(max(N, 1), device=dev)
            cnt = _i32(max(Kn, 1), device=dev)
            arg = _i32(B, H, device=dev)
            score = _f32(N, device=dev)
            last = l + 1 == depth
            if not last:
                rp_n, re_n, col_n = _i32(Kn, device=dev), _i32(Kn, device=dev), _i32(nnz_bound, device=dev)
                dinv_n, self_w_n = _f32(Kn, device=dev), _f32(Kn, device=dev)
            else:
                rp_n = re_n = col_n = dinv_n = self_w_n = None
            nat.call("sag_pool_graph_f32", y, y.stride(0), rowptr, rowend, col, dinv, self_w, wsv, bs, L.gp, Ln.gp, B, L.max_seg, H,
                     score, perm, new_id, xp, xp.stride(0), cnt, read, read.stride(0), arg, int(l > 0),
                     rp_n, re_n, col_n, dinv_n, self_w_n, None, 0)
            saved.append((cat, y, score, new_id, arg, rowptr, col, rowend, dinv, self_w, wcat, wsv, K, Kp))
            if not last:
                rowptr, col, rowend, dinv, self_w = rp_n, col_n, re_n, dinv_n, self_w_n
            cat, K, Kp = cat_n, H, H
        ctx.plan, ctx.saved_levels, ctx.H = plan, saved, H
        ctx.x_needs_grad = x.requires_grad
        ctx.params = params
        return read

    @staticmethod
    def backward(ctx, dread):
        plan, H = ctx.plan, ctx.H
        depth = plan.depth
        dread = dread.contiguous()
        dev = dread.device
        grads = [None] * (5 * depth)
        sets, sunk = [], []
        dxp = None
        dx = None
        for l in range(depth - 1, -1, -1):
            L, Ln = plan.levels[l], plan.levels[l + 1]
            N = L.N
            cat, y, score, new_id, arg, rowptr, col, rowend, dinv, self_w, wcat, wsv, K, Kp = ctx.saved_levels[l]
            wl, bl, wr, ws, bs = ctx.params[5 * l: 5 * l + 5]
            dyb = _f32(N, H, device=dev)
            part = _f32(L.B * (H + 4), device=dev)
            nat.call("sag_pool_graph_bwd_f32", y, y.stride(0), score, new_id, L.gp, Ln.gp, arg, dxp,
                     dxp.stride(0) if dxp is not None else 0, dread, dread.stride(0), rowptr, rowend, col, dinv, self_w, wsv,
                     L.B, L.max_seg, H, dyb, dyb.stride(0), part, None, None, None, 0, None, None, None, None, None)
            # the slabs of dW_l = agg^T dy (+ db) and dW_r = x^T dy: one launch; their sum waits for the end of the backward
            sl = ps.wgrad_slabs(cat, cat[:, Kp:], K, dyb)
            if sl is None:
                raise RuntimeError("SAGPool + SAGEConv stack: weight-gradient shape %d x %d is not taken by the slab kernel" % (K, H))
            dwl, s1 = mp._sink_or_new(wl, (H, K), dev)
            dbl, s2 = mp._sink_or_new(bl, (H,), dev)
            dwr, s3 = mp._sink_or_new(wr, (H, K), dev)
            dws, s4 = mp._sink_or_new(ws, tuple(ws.shape), dev)
            dbs, s5 = mp._sink_or_new(bs, (1,), dev)
            sets.append((sl[0][0], sl[0][1], K, H, dwl, dbl))
            sets.append((sl[1][0], sl[1][1], K, H, dwr, None))
            sets.append((part, L.B, 0, H + 4, None, dws, H, dbs))
            grads[5 * l: 5 * l + 5] = [None if s1 else dwl, None if s2 else dbl, None if s3 else dwr, None if s4 else dws, None if s5 else dbs]
            sunk.append(s1 and s2 and s3 and s4 and s5)
            if l > 0 or ctx.x_needs_grad:
                dcat = _f32(N, 2 * Kp, device=dev)
                nat.call("rowgemm_f32", dyb, dyb.stride(0), wcat, wcat.stride(0), 0, None, dcat, dcat.stride(0), None, N, H, 2 * Kp, 0, 0)
                # dx = A_mean^T dagg + dself   (symmetric edge list: rows of A^T = rows of A; the 1 / deg moves to the gathered rows)
                dxin = _f32(N, Kp, device=dev)
                _prop_mean(rowptr, rowend, col, 1, dcat, dcat.stride(0), dcat[:, Kp:], dcat.stride(0), dxin, dxin.stride(0), N, K)
                if l > 0:
                    dxp = dxin
                else:
                    dx = dxin[:, :K]
        sink = mp.GRAD_SINK
        all_sunk = sink is not None and all(sunk)
        if ps.reduce_oi(sets, norm_sink=sink if all_sunk else None):
            for p in ctx.params:
                sink.normed.add(p.data_ptr())
        return (dx, None, None, *grads)


def sag_sage_stack(x, g, plan, params):
    """readout[B, 2H] = sum over the levels of [gmp || gap] (network.py:36-46) with SAGEConv layers.  params: per level
    (lin_l.weight [H, in], lin_l.bias [H], lin_r.weight [H, in], score weight [H, 1], score bias [1])."""
    return _SagSageStack.apply(x, g, plan, *params)
