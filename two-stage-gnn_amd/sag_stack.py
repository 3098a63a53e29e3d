"""The three conv -> SAGPool -> readout levels of Code/sag/network.py:33-44 as ONE autograd node with no host round trip.

PyG's ``topk`` / ``filter_adj`` (layers.py:20-24) return tensors whose sizes depend on the data, so the reference path
(and the composable drop-ins in ``pyg.py``) synchronise with the host three times per level.  Two facts remove that:
  * k_b = ceil(ratio * n_b) depends on the graph SIZES only, so every level's row count, graph pointer and row -> graph
    map are host-known before the step starts (``SagPlan``, cached per mini-batch structure);
  * the filtered adjacency is only ever consumed by the next level's kernels, so it stays a CSR whose entry count lives
    in ``rowptr'[K]`` on the device (``tsgnn_csr_filter_fill``).
A step is therefore capturable in a hipGraph.  Per level (csrc/sagpool.hip):
  forward   agg = A^ x            (gcn_propagate; A^ = D^-1/2 (A+I) D^-1/2 from per-row coefficients)
            y   = agg W + b       (fp32 MFMA row-panel product; (A^ x) W = A^ (x W), GCNConv network.py:34)
            one workgroup per graph (tsgnn_sag_pool_graph_f32, graphs <= 4,096 nodes; everything below touches one graph):
              s   = A^ (relu(y) w_s) + b_s      (score layer GCNConv(C -> 1), layers.py:18)
              perm, new_id = topk(s)            (rank count / LDS bitonic sort, ties -> smaller node id)
              xp  = relu(y)[perm] * tanh(s[perm])                              (layers.py:21)
              out += [max || mean](xp)          (network.py:36,40,44 and the sum :46)
              A'  = filter(A)                   (symmetric graphs: entries from the graph's old segment base, explicit row
                                                 ends, next level's coefficients; layers.py:23-24)
              agg' = A^' xp                     (the NEXT level's aggregation: its rows, CSR and coefficients are this block's)
            (larger graphs / directed edge lists: the same steps as separate launches + scan + tsgnn_csr_filter_fill)
  backward  one workgroup per graph: [dxp = A^' dagg' of the next level per kept row ->] pooled-row gradients -> score layer
            backward -> du ; reduction of dw_s / db_s partials ; dW, db in one pass ; dagg = du W^T ; level 0: dx = A^ dagg
ReLU is applied by the consumers of ``y`` (it is stored pre-activation), so no activation tensor is written.
"""
import os

import numpy as np
import torch

from . import _native as nat
from . import message_passing as mp

_f32 = mp._f32


def _i32(*shape, device):
    return torch.empty(*shape, dtype=torch.int32, device=device)


class _Level:
    __slots__ = ("B", "N", "sizes", "gp", "row_graph", "max_seg")


class SagPlan:
    """Host-known structure of a SAGPool pipeline over one mini-batch: rows, graph pointers and row -> graph maps of
    the input level and of every pooled level.  Built once per (sizes, ratio) and cached: steady-state steps upload
    nothing."""

    _cache = {}

    def __init__(self, sizes, ratio, device, depth):
        sizes = np.asarray(sizes, dtype=np.int64).reshape(-1)
        if sizes.size == 0 or (sizes < 1).any():
            raise ValueError("every graph needs at least one node")
        self.ratio, self.depth, self.device = float(ratio), int(depth), device
        self.levels = []
        for _ in range(depth + 1):
            L = _Level()
            L.B, L.N, L.sizes, L.max_seg = int(sizes.size), int(sizes.sum()), sizes, int(sizes.max())
            gp = np.zeros(L.B + 1, dtype=np.int32)
            np.cumsum(sizes, out=gp[1:])
            L.gp = torch.from_numpy(gp).to(device)
            L.row_graph = torch.from_numpy(np.repeat(np.arange(L.B, dtype=np.int32), sizes)).to(device)
            self.levels.append(L)
            k = np.ceil(np.float32(ratio) * sizes.astype(np.float32)).astype(np.int64)      # float32, as PyG's topk computes it
            sizes = np.minimum(k, sizes)
        if self.levels[0].max_seg > int(nat.lib().tsgnn_topk_max_segment()):
            raise RuntimeError("graphs of more than %d nodes are not supported by the LDS top-k" % nat.lib().tsgnn_topk_max_segment())

    @classmethod
    def get(cls, sizes, ratio, device, depth=3):
        sizes = np.asarray(sizes, dtype=np.int64).reshape(-1)
        key = (sizes.tobytes(), float(ratio), str(device), int(depth))
        plan = cls._cache.get(key)
        if plan is None:
            if len(cls._cache) > 64:
                cls._cache.clear()
            plan = cls._cache[key] = cls(sizes, ratio, device, depth)
        return plan


def gcn_coef(g):
    """(dinv, self_w) of PyG gcn_norm for the unit-weight graph g, cached on it."""
    c = getattr(g, "_gcn_coef", None)
    if c is None:
        R = g.total_rows
        dinv, self_w = _f32(R, device=g.device), _f32(R, device=g.device)
        nat.call("gcn_coef_f32", g.rowptr, g.col, R, dinv, self_w)
        c = g._gcn_coef = (dinv, self_w)
    return c


def propagate(rowptr, col, dinv, self_w, x, n_rows, relu_in=False, bias=None, w_dot=None, dot_bias=None, want_y=True, rowend=None):
    """A^ x (+ bias) and / or its dot with w_dot; see tsgnn_gcn_propagate_re_f32 (rowend: explicit row ends or None)."""
    F = x.size(1)
    y = _f32(n_rows, F, device=x.device) if want_y else None
    t = _f32(n_rows, device=x.device) if w_dot is not None else None
    nat.call("gcn_propagate_re_f32", rowptr, rowend, col, dinv, self_w, x, x.stride(0), int(relu_in), bias, w_dot, dot_bias, y,
             y.stride(0) if want_y else 0, t, int(n_rows), int(F))
    return y, t


def _linear(z, w, bias):
    """z @ w + bias, w stored [in, out]"""
    R, K, N = z.size(0), w.size(0), w.size(1)
    v = _f32(R, N, device=z.device)
    if mp.rowgemm_ok(z, z.stride(0), w, w.stride(0), K, N, False):
        nat.call("rowgemm_f32", z, z.stride(0), w, w.stride(0), 0, bias, v, v.stride(0), None, R, K, N, 0, 0)
    else:
        nat.call("linear_l2norm_f32", z, z.stride(0), w, w.stride(0), bias, v, v.stride(0), None, R, K, N, 0)
    return v


def _linear_t(du, w):
    """du @ w^T, w stored [in, out] -> [rows, in]"""
    R, K, N = du.size(0), w.size(0), w.size(1)
    dz = _f32(R, K, device=du.device)
    if mp.rowgemm_ok(du, du.stride(0), w, w.stride(0), N, K, True):
        nat.call("rowgemm_f32", du, du.stride(0), w, w.stride(0), 1, None, dz, dz.stride(0), None, R, N, K, 0, 0)
    else:
        mp.gemm(du, du.stride(0), 1, w, 1, w.stride(0), dz, dz.stride(0), 1, R, K, N)
    return dz


PER_GRAPH_POOL = True           # level tail (score -> top-k -> gather -> readout) as one workgroup per graph when graphs are small
FUSED_NEXT_PROPAGATE = True     # ... which then also forms the next level's aggregation and, backward, takes the next level's
                                # dagg instead of dxp (A^ applied per kept row in the kernel): two launches less per pooled level
NARROW_FUSED_MAX_ROWS = 32768             # level 0 on <= 8 input columns: aggregation + transform in one launch below this many rows
FUSED_NEXT_PROPAGATE_MAX_GRAPHS = 1024   # a launch-count saving: 207 vs 213 us at 128 graphs, 1.21 vs 1.17 ms at 8,192 (few lane groups per block)


def _al16(t):
    """the float4 kernels read the small parameter vectors with 16-byte loads: a view at an odd offset is copied"""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def supported(hidden):
    return bool(nat.lib().tsgnn_sag_supported(int(hidden)))


DEFERRED_REDUCE = os.environ.get("TSGNN_SAG_DEFERRED_REDUCE", "1") != "0"   # one closing reduction for every level's slabs + score rows
MERGED_BWD = os.environ.get("TSGNN_SAG_MERGED_BWD", "1") != "0"   # a conv layer's weight-gradient slabs beside dagg = du W^T (one launch)


def _wgrad_beside_dagg(agg, du, W, nb):
    """(ws, nslab, dagg[R, K]) of a 128 -> 128 conv layer: the slab blocks of the weight gradient and the product du W^T as roles of one
    launch (tsgnn_gat_bwd_products_f32: both only read du); the slabs are summed by the backward's ONE closing reduction.  None when the
    shape is not taken"""
    R, N, K = int(du.size(0)), int(du.size(1)), int(agg.size(1))
    if not (K == 128 and N == 128 and nb <= 256 and R >= 64 and agg.stride(0) % 4 == 0 and du.stride(0) % 4 == 0 and W.stride(0) % 4 == 0
            and agg.data_ptr() % 16 == 0 and du.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0 and W.size(0) == K and W.size(1) == N):
        return None
    nslab, rps, need = mp.wgrad_plan(R, K, N, agg.stride(0), du.stride(0))
    if nslab <= 0 or nslab >= 512:
        return None
    ws = _f32(need, device=du.device)
    dagg = _f32(R, K, device=du.device)
    if not nat.try_call("gat_bwd_products_f32", agg, agg.stride(0), du, du.stride(0), R, K, N, W, W.stride(0), dagg, dagg.stride(0),
                        nslab, rps, ws):
        return None
    return ws, nslab, dagg


class _SagStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, plan, *params):
        depth = plan.depth
        if len(params) != 4 * depth:
            raise ValueError("expected (weight, bias, score weight, score bias) per level")
        if plan.levels[0].N != g.total_rows or x.size(0) != g.total_rows:
            raise ValueError("plan, graph and features disagree on the number of nodes")
        if g.val is not None:
            raise NotImplementedError("edge weights are never passed by the reference (network.py:34)")
        dev = x.device
        x = x.contiguous().float()
        H = params[0].size(1)
        B = plan.levels[0].B
        rowptr, col, rowend = g.rowptr, g.col, None          # rowend: explicit row ends once a level was filtered in-kernel
        dinv, self_w = gcn_coef(g)
        sym = bool(g.symmetric)
        rowptr_t, col_t = (rowptr, col) if sym else g.transposed(None)[:2]
        nnz_bound = max(int(col.numel()), 1)
        read = _f32(B, 2 * H, device=dev)
        pool_graph_max = int(nat.lib().tsgnn_sag_pool_graph_max_nodes())
        saved = []
        xin = x
        agg_next = None
        for l in range(depth):
            L, Ln = plan.levels[l], plan.levels[l + 1]
            N, K = L.N, Ln.N
            W, b, ws, bs = params[4 * l: 4 * l + 4]
            W, b = _al16(W.contiguous()), _al16(b.contiguous())
            wsv = _al16(ws.contiguous().view(-1))
            if agg_next is not None:
                agg = agg_next                                  # formed by the previous level's per-graph kernel
                y = _linear(agg, W, b)
            elif xin.size(1) <= 8 and N < NARROW_FUSED_MAX_ROWS:
                # narrow input (one constant column on the IMDB sets): aggregation and transform in one launch (a launch-count
                # saving: 162 vs 168 us at 128 graphs; from 32,768 rows on the thread-per-row propagate + MFMA product are faster)
                # (agg rows padded to 4 floats: the weight-gradient slab kernel reads them as aligned rows)
                agg, y = _f32(N, (xin.size(1) + 3) // 4 * 4, device=dev)[:, :xin.size(1)], _f32(N, H, device=dev)
                nat.call("gcn_propagate_affine_f32", rowptr, rowend, col, dinv, self_w, xin, xin.stride(0), agg, agg.stride(0), N,
                         xin.size(1), W, W.stride(0), b, y, y.stride(0), H)
            else:
                agg, _ = propagate(rowptr, col, dinv, self_w, xin, N, rowend=rowend)
                y = _linear(agg, W, b)
            agg_next = None
            perm, new_id = _i32(max(K, 1), device=dev), _i32(max(N, 1), device=dev)
            xp, cnt = _f32(K, H, device=dev), _i32(max(K, 1), device=dev)
            arg = _i32(B, H, device=dev)
            fused_filter = False
            if L.max_seg <= pool_graph_max and PER_GRAPH_POOL:
                # score layer, top-k, gated gather, readout and (symmetric graphs) the CSR filter: one workgroup per graph,
                # one launch; the pooled adjacency keeps each graph at its old segment base -> explicit row ends
                score = _f32(N, device=dev)
                fused_filter = sym and l + 1 < depth
                if fused_filter:
                    rp_n, re_n, col_n = _i32(K, device=dev), _i32(K, device=dev), _i32(nnz_bound, device=dev)
                    dinv_n, self_w_n = _f32(K, device=dev), _f32(K, device=dev)
                    if FUSED_NEXT_PROPAGATE and B <= FUSED_NEXT_PROPAGATE_MAX_GRAPHS:
                        agg_next = _f32(K, H, device=dev)       # the next level's A^ xp, formed by the same launch
                else:
                    rp_n = re_n = col_n = dinv_n = self_w_n = None
                nat.call("sag_pool_graph_f32", y, y.stride(0), rowptr, rowend, col, dinv, self_w, wsv, bs, L.gp, Ln.gp, B, L.max_seg, H,
                         score, perm, new_id, xp, xp.stride(0), cnt, read, read.stride(0), arg, int(l > 0),
                         rp_n, re_n, col_n, dinv_n, self_w_n, agg_next, H if agg_next is not None else 0)
            else:
                if rowend is not None:
                    raise RuntimeError("explicit row ends only arise from the per-graph kernel, whose size limit applies to every level")
                _, score = propagate(rowptr, col, dinv, self_w, y, N, relu_in=True, w_dot=wsv, dot_bias=bs, want_y=False)
                nat.call("topk_segments_f32", score, L.gp, Ln.gp, B, L.max_seg, perm, new_id)
                nat.call("sag_pool_gather_f32", y, y.stride(0), score, perm, new_id, rowptr, col, K, H, 1, xp, xp.stride(0), cnt)
                nat.call("sag_readout_f32", xp, xp.stride(0), Ln.gp, B, H, int(l > 0), read, read.stride(0), arg)
            saved.append((xin, agg, y, score, new_id, arg, rowptr, col, rowptr_t, col_t, dinv, self_w, W, wsv, rowend,
                          agg_next is not None))
            if fused_filter:
                rowptr, col, rowend, rowptr_t, col_t, dinv, self_w = rp_n, col_n, re_n, rp_n, col_n, dinv_n, self_w_n
            elif l + 1 < depth:                                 # the adjacency after the last pool is never used
                rp_n, col_n = _i32(K + 1, device=dev), _i32(nnz_bound, device=dev)
                dinv_n, self_w_n = _f32(K, device=dev), _f32(K, device=dev)
                nat.call("scan_short_i32", cnt, K, rp_n)
                nat.call("csr_filter_fill", rowptr, col, perm, new_id, K, rp_n, col_n, dinv_n, self_w_n)
                if sym:
                    rp_tn, col_tn = rp_n, col_n
                else:
                    cnt_t = _i32(max(K, 1), device=dev)
                    nat.call("sag_pool_gather_f32", None, 0, score, perm, new_id, rowptr_t, col_t, K, H, 0, None, 0, cnt_t)
                    rp_tn, col_tn = _i32(K + 1, device=dev), _i32(nnz_bound, device=dev)
                    nat.call("scan_short_i32", cnt_t, K, rp_tn)
                    nat.call("csr_filter_fill", rowptr_t, col_t, perm, new_id, K, rp_tn, col_tn, None, None)
                rowptr, col, rowptr_t, col_t, dinv, self_w = rp_n, col_n, rp_tn, col_tn, dinv_n, self_w_n
            xin = xp
        ctx.plan, ctx.saved_levels, ctx.H, ctx.sym = plan, saved, H, sym
        ctx.x_needs_grad = x.requires_grad
        ctx.params = params
        return read

    @staticmethod
    def backward(ctx, dread):
        plan, H = ctx.plan, ctx.H
        depth = plan.depth
        dread = dread.contiguous()
        dev = dread.device
        grads = [None] * (4 * depth)
        pool_graph_max = int(nat.lib().tsgnn_sag_pool_graph_max_nodes())
        sym = ctx.sym
        dxp = None
        dx = None
        nxt = None          # (dagg, rowptr, rowend, col, dinv, self_w) of level l + 1 when this level's kernel forms dxp itself
        sets, set_params, sunk = [], [], []
        for l in range(depth - 1, -1, -1):
            L, Ln = plan.levels[l], plan.levels[l + 1]
            N = L.N
            xin, agg, y, score, new_id, arg, rowptr, col, rowptr_t, col_t, dinv, self_w, W, wsv, rowend, _ = ctx.saved_levels[l]
            dyb = _f32(N, H, device=dev)
            pW, pb, pws, pbs = ctx.params[4 * l: 4 * l + 4]
            du_job = None
            if sym and L.max_seg <= pool_graph_max and PER_GRAPH_POOL:
                # pooled-row gradients -> score-layer backward -> du: one workgroup per graph; the partial rows of (dw_s, db_s) it leaves are
                # summed by the closing reduction
                part = _f32(L.B * (H + 4), device=dev)
                nat.call("sag_pool_graph_bwd_f32", y, y.stride(0), score, new_id, L.gp, Ln.gp, arg, dxp,
                         dxp.stride(0) if dxp is not None else 0, dread, dread.stride(0), rowptr, rowend, col, dinv, self_w, wsv,
                         L.B, L.max_seg, H, dyb, dyb.stride(0), part, None, None,
                         *((nxt[0], nxt[0].stride(0)) + nxt[1:] if nxt is not None else (None, 0, None, None, None, None, None)))
                du_job = (part, L.B, H)
            else:
                if nxt is not None:
                    raise RuntimeError("the in-kernel gradient propagate belongs to the per-graph backward")
                dws, dbs = _f32(H, device=dev), _f32(1, device=dev)
                dscore = _f32(N, device=dev)
                nat.call("sag_pool_bwd_f32", y, y.stride(0), score, new_id, Ln.row_graph, Ln.gp, arg, dxp,
                         dxp.stride(0) if dxp is not None else 0, dread, dread.stride(0), N, H, 1, dyb, dyb.stride(0), dscore)
                nb = int(nat.lib().tsgnn_sag_du_blocks(N, H))
                part = _f32(nb * (H + 4), device=dev)
                # dt = A^T dscore: the score layer's propagate transposed
                nat.call("sag_du_f32", rowptr_t, rowend, col_t, dinv, self_w, dscore, y, y.stride(0), wsv, dyb, dyb.stride(0), N, H, part,
                         dws, dbs)
            need_dagg = l > 0 or ctx.x_needs_grad
            dagg = None
            Kin = int(agg.size(1))
            slabs = None
            if du_job is not None and DEFERRED_REDUCE:
                if need_dagg and MERGED_BWD:
                    both = _wgrad_beside_dagg(agg, dyb, W, L.B)     # weight-gradient slabs and dagg = du W^T: ONE launch
                    if both is not None:
                        slabs, dagg = both[:2], both[2]
                if slabs is None:
                    slabs = mp.linear_wgrad_slabs(agg, Kin, dyb)
            if slabs is not None:
                # every level's slabs and score-layer partial rows wait for ONE reduction at the end of the backward, written straight into
                # the flat gradient bucket (with |grad|^2 shares) when a FlatTrainer is listening
                dW, s1 = mp._sink_or_new(pW, (Kin, H), dev)
                db, s2 = mp._sink_or_new(pb, (H,), dev)
                dws, s3 = mp._sink_or_new(pws, tuple(pws.shape), dev)
                dbs, s4 = mp._sink_or_new(pbs, (1,), dev)
                sets.append((slabs[0], slabs[1], Kin, H, dW, db, H, None, 1))
                sets.append((part, L.B, 0, H + 4, None, dws, H, dbs))
                set_params += [pW, pb, pws, pbs]
                sunk.append(s1 and s2 and s3 and s4)
                grads[4 * l: 4 * l + 4] = [None if s1 else dW, None if s2 else db, None if s3 else dws, None if s4 else dbs]
            else:
                if du_job is not None:
                    dws, dbs = _f32(H, device=dev), _f32(1, device=dev)
                    du_job = du_job + (dws, dbs)
                dW, db = mp.linear_wgrad(agg, Kin, dyb, True, du_job=du_job)
                grads[4 * l: 4 * l + 4] = [dW, db, dws.view(-1, 1), dbs]
            if need_dagg:
                if dagg is None:
                    dagg = _linear_t(dyb, W)
                nxt = None
                if l > 0 and ctx.saved_levels[l - 1][-1]:
                    # level l - 1 ran the fused per-graph forward (symmetric, filter in-kernel): its backward kernel takes
                    # dagg and this level's CSR and forms dxp = A^ dagg per kept row itself
                    nxt, dxp = (dagg, rowptr, rowend, col, dinv, self_w), None
                    continue
                dxin, _ = propagate(rowptr_t, col_t, dinv, self_w, dagg, N, rowend=rowend)
                if l > 0:
                    dxp = dxin
                else:
                    dx = dxin
        if sets:
            from .pyg_sage import reduce_oi
            sink = mp.GRAD_SINK
            if reduce_oi(sets, norm_sink=sink if (sink is not None and all(sunk)) else None):
                for p in set_params:
                    sink.normed.add(p.data_ptr())
        return (dx, None, None, *grads)


def sag_stack(x, g, plan, params):
    """readout[B, 2H] = sum over the levels of [gmp || gap] (network.py:36-46).  params: per level
    (conv weight [in, H], conv bias [H], score weight [H, 1], score bias [1])."""
    return _SagStack.apply(x, g, plan, *params)
