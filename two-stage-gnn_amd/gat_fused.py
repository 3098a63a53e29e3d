"""One GAT layer (all heads) on the fused kernels of csrc/gat_fused.hip — DGATHead / DGATLayer of
Code/sage+gat+diffpool/encoders_GAT.py:29-49, 68-84 in two launches forward (packed projection, attention + aggregation +
ELU) and four backward (column kernel, score row sums, dW' blocks + their reduction, dx), plus ONE pack and ONE unpack launch
per step for the parameters of all layers.

The heads' (w, a) are packed into W' = [W_0 | .. | W_{H-1} | W_h a1_h | W_h a2_h] so that the projection hp = x W' carries the
attention scalars a1 . h_i, a2 . h_j (:35-36) as 2H extra columns; their gradients return through dW' and are folded back onto
(w_h, a_h) by the unpack.  Attention dropout (:42) is a Philox mask regenerated in the backward.

Taken when the kernels support the head shape (tsgnn_gat_fused_supported), every graph has its own features (B == 1 or
per_graph_features) and the edge-less columns of every graph can be listed (att._isolated_list).  Everything else — the T4
broadcast of graph 0's features at B > 1, padded batches with hundreds of edge-less columns — stays on the per-op path of
attention.py."""
import os

import numpy as np
import torch

from . import _native as nat
from . import attention as att
from . import message_passing as mp

FUSED = os.environ.get("TSGNN_GAT_FUSED", "1") != "0"
_LMAX, _HMAX = 4, 8


def _f32(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def packed_width(H, Fo):
    return (H * Fo + 2 * H + 3) // 4 * 4


def heads_ok(heads):
    h0 = heads[0]
    return (FUSED and h0.w.is_cuda and 1 <= len(heads) <= _HMAX and len(heads) * h0.output_dim <= 2048
            and all(hd.w.data_ptr() % 16 == 0 and hd.w.is_contiguous() for hd in heads)
            and all(hd.input_dim == h0.input_dim and hd.output_dim == h0.output_dim for hd in heads)
            and bool(nat.lib().tsgnn_gat_fused_supported(len(heads), int(h0.output_dim))))


def batch_ok(g, rows, H, dropout_on):
    """the edge-less columns of g can be listed for the kernels (or there are none)"""
    rp_t, _, _ = g.transpose_map()
    iso = att._isolated_columns(g, rp_t, rows, H)
    ragged = getattr(g, "row_mult", None) is not None
    if dropout_on and ragged:
        return False                       # one representative per graph is exact only while all of its copies stay identical
    lst = att._isolated_list(g, iso, force=dropout_on)
    return lst is not None or att.isolated_count(g) == 0


def _desc(layers, ptrs):
    """layers: [(H, Fin, Fo, Ns, w tensors, a tensors)], ptrs: [(wp, gw, ga)] -> host int64 description"""
    words = int(nat.lib().tsgnn_gat_pack_desc_words())
    d = np.zeros(1 + len(layers) * words, dtype=np.int64)
    d[0] = len(layers)
    for i, ((H, Fin, Fo, Ns, ws, as_), (wp, gw, ga)) in enumerate(zip(layers, ptrs)):
        o = 1 + i * words
        d[o:o + 4] = (H, Fin, Fo, Ns)
        d[o + 4] = wp.data_ptr()
        d[o + 5] = gw.data_ptr() if gw is not None else 0
        d[o + 6] = ga.data_ptr() if ga is not None else 0
        for h in range(H):
            d[o + 7 + h] = ws[h].data_ptr()
            d[o + 7 + _HMAX + h] = as_[h].data_ptr()
    return d


class _PackLayers(torch.autograd.Function):
    """(w_0.., a_0..) of every layer -> the layers' W' in ONE launch; backward: all 2H parameter gradients of all layers from
    their dW' in ONE launch."""

    @staticmethod
    def forward(ctx, nheads, *params):
        layers, k = [], 0
        for H in nheads:
            ws = [p.detach().contiguous() for p in params[k:k + H]]
            as_ = [p.detach().contiguous() for p in params[k + H:k + 2 * H]]
            k += 2 * H
            Fin, Fo = int(ws[0].size(0)), int(ws[0].size(1))
            layers.append((H, Fin, Fo, packed_width(H, Fo), ws, as_))
        dev = params[0].device
        outs = [_f32(L[1], L[3], device=dev) for L in layers]
        d = _desc(layers, [(o, None, None) for o in outs])
        nat.call("gat_pack_f32", d.ctypes.data)
        ctx.layers = layers
        # the slab sets the layers' backward leaves for this node's backward to reduce: a container of THIS forward (not a module
        # global: two models, streams or threads must not see each other's entries, and a backward that never reaches this node
        # leaves nothing behind — the container dies with the graph)
        ctx.pending = []
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dwps):
        flush_reductions(ctx.pending)                            # the layers' dW' still sit in their slabs: ONE reduction launch for two
        layers = ctx.layers
        dev = layers[0][4][0].device
        ptrs, res = [], []
        for L, dwp in zip(layers, dwps):
            H, Fin, Fo, Ns = L[:4]
            if dwp is None:
                dwp = torch.zeros(Fin, Ns, dtype=torch.float32, device=dev)
            ptrs.append((dwp.contiguous(), _f32(H, Fin, Fo, device=dev), _f32(H, 2 * Fo, device=dev)))
        d = _desc(layers, ptrs)
        nat.call("gat_unpack_f32", d.ctypes.data)
        for L, (_, gw, ga) in zip(layers, ptrs):
            H, Fo = L[0], L[2]
            res += [gw[h] for h in range(H)] + [ga[h].reshape(2 * Fo, 1) for h in range(H)]
        return (None,) + tuple(res)


def pack_layers(layers_heads):
    """layers_heads: list (<= 4) of lists of DGATHead -> tuple of W' (one per layer)"""
    params = []
    for heads in layers_heads:
        params += [hd.w for hd in heads] + [hd.a for hd in heads]
    outs = _PackLayers.apply(tuple(len(h) for h in layers_heads), *params)
    pending = outs[0].grad_fn.pending if (outs and outs[0].grad_fn is not None and hasattr(outs[0].grad_fn, "pending")) else None
    for o in outs:
        # its gradient goes to _PackLayers.backward, which reduces the pending slabs first; `uses` counts the layer calls that
        # consume this W' (deferring is only sound with exactly one: autograd would otherwise SUM two not-yet-reduced gradients)
        o._tsgnn_pending = pending
        o._tsgnn_uses = [0]
    return outs


# A layer's dW' is read only when the parameters are unpacked at the end of the backward pass: the layers leave their slabs pending
# and _PackLayers.backward reduces them two products per launch (a GAT encoder has two layers: one reduction launch instead of two).
DEFER_REDUCE = os.environ.get("TSGNN_GAT_DEFER_REDUCE", "1") != "0"


def flush_reductions(pending):
    """reduce the slab sets (ws, nslab, K_in, N, dw) the layers of one forward left pending, two per launch"""
    while pending:
        a = pending.pop(0)
        if pending:
            b = pending.pop(0)
            nat.call("wgrad_blocks_reduce2_f32", a[0], a[1], a[2], a[3], a[4], a[4].stride(0), b[0], b[1], b[2], b[3], b[4], b[4].stride(0))
        else:
            nat.call("wgrad_blocks_reduce_f32", a[0], a[1], a[2], a[3], a[4], a[4].stride(0))


def _reduce(ws, nslab, K_in, N, dw, defer):
    """defer: the pending list of the pack node that will consume dw (None / False: reduce now)"""
    if defer is not None and defer is not False and DEFER_REDUCE:
        defer.append((ws, nslab, K_in, N, dw))
    else:
        nat.call("wgrad_blocks_reduce_f32", ws, nslab, K_in, N, dw, dw.stride(0))


def wgrad_blocks(z, K_in, du, defer=None):
    """dW[K_in, N] = z[:, :K_in]^T du (N = du.size(1) <= 512) in two launches; None if the shape is not taken.  defer: only the slab
    launch now, the reduction with the next flush_reductions() (the returned tensor is filled then)"""
    R, N = int(du.size(0)), int(du.size(1))
    nslab = np.zeros(1, dtype=np.int32)
    rps = np.zeros(1, dtype=np.int64)
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("wgrad_blocks_plan", R, int(K_in), N, int(z.stride(0)), int(du.stride(0)), nslab.ctypes.data, rps.ctypes.data,
                      need.ctypes.data)
    if int(nslab[0]) <= 0 or z.data_ptr() % 16 or du.data_ptr() % 16:
        return None
    ws = _f32(int(need[0]), device=du.device)
    dw = _f32(int(K_in), N, device=du.device)
    if defer is not None and defer is not False and DEFER_REDUCE:
        nat.call("wgrad_blocks_slabs_f32", z, z.stride(0), du, du.stride(0), R, int(K_in), N, int(nslab[0]), int(rps[0]), ws)
        _reduce(ws, int(nslab[0]), int(K_in), N, dw, defer)
        return dw
    nat.call("wgrad_blocks_f32", z, z.stride(0), du, du.stride(0), R, int(K_in), N, int(nslab[0]), int(rps[0]), ws, dw, dw.stride(0))
    return dw


MERGED_BWD_PRODUCTS = os.environ.get("TSGNN_GAT_MERGED_BWD", "1") != "0"   # a layer's weight-gradient slabs beside its input-gradient product


def bwd_products(x, K_in, du, wp, defer=None):
    """(dW'[K_in, N], dx[R, x.size(1)]) of hp = x W' from du = dhp: the slab launch of wgrad_blocks and the product du W'^T as ONE
    launch (tsgnn_gat_bwd_products_f32) + the slabs' reduction; None when the shape is not taken (K_in <= 128, ...)"""
    R, N = int(du.size(0)), int(du.size(1))
    if not (128 < K_in <= 512 and N <= 512 and N % 4 == 0 and x.size(1) == K_in and x.stride(0) % 4 == 0 and du.stride(0) % 4 == 0
            and wp.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and du.data_ptr() % 16 == 0 and wp.data_ptr() % 16 == 0 and R > 0):
        return None
    nslab = np.zeros(1, dtype=np.int32)
    rps = np.zeros(1, dtype=np.int64)
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("wgrad_blocks_plan", R, int(K_in), N, int(x.stride(0)), int(du.stride(0)), nslab.ctypes.data, rps.ctypes.data,
                      need.ctypes.data)
    if int(nslab[0]) <= 0:
        return None
    ws = _f32(int(need[0]), device=du.device)
    dw = _f32(int(K_in), N, device=du.device)
    dx = _f32(R, int(K_in), device=du.device)
    if not nat.try_call("gat_bwd_products_f32", x, x.stride(0), du, du.stride(0), R, int(K_in), N, wp, wp.stride(0), dx, dx.stride(0),
                        int(nslab[0]), int(rps[0]), ws):
        return None
    _reduce(ws, int(nslab[0]), int(K_in), N, dw, defer)
    return dw, dx


class _GatLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wp, g, H, Fo, slope, mean_heads, apply_elu, drop_p, seed, ctr, readout=False):
        """x [R, >= Fin] (16-byte rows), wp [Fin, Ns] -> y [R, H*Fo] ([R, Fo] with mean_heads); readout: returns the max readout of y
        over each graph's rows [B, .] instead (the LAST layer, encoders_GAT.py:189) — its gradient then reaches the backward kernel
        as (dout, winners) and dy is never a tensor"""
        R, Fin, Ns, C = int(x.size(0)), int(wp.size(0)), int(wp.size(1)), H * Fo
        dev = x.device
        hp = _f32(R, Ns, device=dev)
        nat.call("rowgemm_f32", x, x.stride(0), wp, wp.stride(0), 0, None, hp, hp.stride(0), None, R, Fin, Ns, 0, 0)
        rp_t, col_t, src_e_t = g.transpose_map()
        iso = att._isolated_columns(g, rp_t, R, H)
        lst = att._isolated_list(g, iso, force=drop_p > 0.0)
        y = _f32(R, Fo if mean_heads else C, device=dev)
        stat = _f32(R, H, 2, device=dev)
        i_idx, i_w, i_ptr = lst if lst is not None else (None, None, None)
        nat.call("gat_attn_fwd_f32", hp, hp.stride(0), g.rowptr, g.col, rp_t, col_t, R, H, Fo, float(slope), att._row_seg(g),
                 int(g.nmax), i_idx, i_w, i_ptr, 1.0 / max(int(g.nmax), 1), int(mean_heads), int(apply_elu), float(drop_p),
                 int(seed), ctr, stat, y, y.stride(0))
        ctx.cfg = (g, H, Fo, slope, mean_heads, apply_elu, drop_p, seed, Fin)
        # wp came from pack_layers: its gradient's consumer (the pack node's backward) reduces pending slabs — sound only while this
        # call is the ONLY consumer of wp (a second one, a hook or retain_grad would make autograd sum un-reduced gradients)
        ctx.pending = getattr(wp, "_tsgnn_pending", None)
        ctx.uses = getattr(wp, "_tsgnn_uses", None)
        if ctx.uses is not None:
            ctx.uses[0] += 1
        ctx.ctr = ctr
        ctx.lst = lst
        ctx.readout = bool(readout)
        if readout:
            out, arg = mp.readout_fwd_raw(y, g)
            ctx.save_for_backward(x, wp, hp, y, iso, stat, arg)
            return out
        ctx.save_for_backward(x, wp, hp, y, iso, stat)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp, hp, y, iso, stat = ctx.saved_tensors[:6]
        g, H, Fo, slope, mean_heads, apply_elu, drop_p, seed, Fin = ctx.cfg
        lst = ctx.lst
        R, Ns, C = int(x.size(0)), int(wp.size(1)), H * Fo
        dev = x.device
        ro = None
        if ctx.readout:                                          # dy = the readout's gradient: handed on as (dout, winners)
            dro = mp.readout_dout_in_place(dy)
            if dro.stride(0) % 4 or dro.data_ptr() % 16:
                dro = dro.contiguous()
            ro, dy = (dro, ctx.saved_tensors[6]), None
        else:
            dy = dy.contiguous()
        rp_t, col_t, src_e_t = g.transpose_map()
        nnz = max(int(g.nnz), 1)
        dhp = _f32(R, Ns, device=dev)
        t1, t2, S = _f32(nnz, H, device=dev), _f32(nnz, H, device=dev), _f32(R, H, device=dev)
        i_idx, i_w, i_ptr = lst if lst is not None else (None, None, None)
        us = 1.0 / max(int(g.nmax), 1)
        dupart = _f32(int(g.B) * int(nat.lib().tsgnn_gat_bwd_parts(int(g.B))) * C, device=dev) if lst is not None else None
        nat.call("gat_attn_bwd_ro_f32", hp, hp.stride(0), y, y.stride(0), dy, dy.stride(0) if dy is not None else 0, rp_t, col_t, R, H, Fo,
                 float(slope), int(mean_heads), int(apply_elu), g.graph_ptr, int(g.B), i_idx, i_w, i_ptr, iso if lst is not None else None,
                 H, us, float(drop_p), int(seed), ctx.ctr, stat, dhp, Ns, t1, t2, S, dupart,
                 ro[0] if ro else None, ro[0].stride(0) if ro else 0, ro[1] if ro else None, g.row_graph if ro else None)
        fin = lst is not None and drop_p == 0.0                 # (with dropout the backward completes the listed columns itself)
        nat.call("gat_score_rowsum_f32", g.rowptr, g.col, att._inverse_entry_map(g, src_e_t), t1, t2, S, R, H, dhp, dhp.stride(0), C,
                 dupart if fin else None, int(g.B), i_idx if fin else None, i_w if fin else None, i_ptr if fin else None, us)
        want_w = ctx.needs_input_grad[1]
        # defer the slab reduction to the pack node only when that node will run (W' needs a gradient) and this is W''s one consumer
        defer = ctx.pending if (want_w and ctx.pending is not None and ctx.uses is not None and ctx.uses[0] == 1 and not wp.retains_grad) else None
        if ctx.needs_input_grad[0] and want_w and MERGED_BWD_PRODUCTS:
            both = bwd_products(x, Fin, dhp, wp, defer=defer)    # dW' slabs and dx = dhp W'^T side by side in one launch
            if both is not None:
                return both[1], both[0], None, None, None, None, None, None, None, None, None, None
        dwp = None
        if want_w:
            dwp = wgrad_blocks(x, Fin, dhp, defer=defer)
            if dwp is None:
                dwp = mp.gemm_tn_splitk(x, Fin, dhp)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _f32(R, int(x.size(1)), device=dev)
            if x.size(1) > Fin:
                dx[:, Fin:].zero_()
            nat.call("rowgemm_f32", dhp, dhp.stride(0), wp, wp.stride(0), 1, None, dx, dx.stride(0), None, R, Ns, Fin, 0, 0)
        return dx, dwp, None, None, None, None, None, None, None, None, None, None


def readout_ok(g, rows):
    """the last layer's max readout may ride in the layer's node (every row is a slot of its graph: no separate ghost rows)"""
    return READOUT_IN_LAYER and g.row_graph is not None and g.n_ghost == 0 and int(g.total_rows) == int(rows)


def gat_layer(x, wp, g, H, Fo, slope, mean_heads, apply_elu, drop_p=0.0, readout=False):
    """drop_p > 0: attention dropout with a fresh mask per call — the key is (process seed, device counter): the counter is
    advanced and snapshotted ON THE DEVICE (two tiny launches), so a step captured in a hipGraph draws a new mask at every
    replay and the backward of this call regenerates exactly the mask of its forward."""
    seed, ctr = 0, None
    if drop_p > 0.0:
        seed, ctr = dropout_key(x.device)
    return _GatLayer.apply(x, wp, g, int(H), int(Fo), float(slope), bool(mean_heads), bool(apply_elu), float(drop_p), int(seed), ctr,
                           bool(readout))


READOUT_IN_LAYER = os.environ.get("TSGNN_GAT_READOUT_IN_LAYER", "1") != "0"   # the last layer's max readout inside its autograd node

_drop_state = {}        # device -> (process seed, int64 device counter)
last_dropout_key = None  # (seed, counter snapshot) of the most recent dropout layer call (tests hand it to the dense oracle)


def dropout_key(dev):
    """(seed, snapshot): the process seed (torch's CPU generator at first use: torch.manual_seed governs it) and a private
    snapshot of the device counter after advancing it"""
    global last_dropout_key
    st = _drop_state.get(dev)
    if st is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        st = _drop_state[dev] = (seed, torch.zeros(1, dtype=torch.int64, device=dev))
    st[1].add_(1)
    snap = st[1].clone()
    last_dropout_key = (st[0], snap)
    return st[0], snap
