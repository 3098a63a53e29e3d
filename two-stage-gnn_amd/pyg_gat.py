"""PyG ``GATConv`` as FUSED launches (csrc/gatconv.hip): per-target edge softmax with the attention scalars riding in the projection.

A layer is: projection hp = x W' (features + both attention scalars of every head: W' packed once per step for all layers), row statistics,
ONE wave-per-target-row launch (softmax coefficients from the scalars, sources gathered 8 at a time, mean over heads / bias / ELU in
registers) forward; backward: target-side launch (dpre, d alpha, S, d s_dst, per-entry terms), two launches over A^T (dh, d s_src), the
bias gradient as a column sum of dpre, the two products dW' = x^T dhp and dx = dhp W'^T side by side, and ONE unpack launch per step that
folds dW' back onto (lin_l.weight, att_l, att_r) of every layer.  No per-edge alpha tensor survives the forward; nothing is a torch op.

PARITY UNPINNED: the oracle is oracle/pyg_ref.gat_conv.  Attention dropout > 0 takes the per-op path of pyg.GATConv
(a per-entry multiplier the test hands to the oracle); the reference never enables dropout in its GAT (train.py:259-261).
"""
import os

import numpy as np
import torch

from . import _native as nat
from . import attention as att
from . import gat_fused as gf
from . import message_passing as mp

FUSED = os.environ.get("TSGNN_GATCONV_FUSED", "1") != "0"
_f32 = mp._f32


def packed_width(H, Co):
    return (H * Co + 2 * H + 3) // 4 * 4


def conv_ok(conv, x):
    return (FUSED and x.is_cuda and conv.lin_l.weight.is_contiguous() and conv.lin_l.weight.data_ptr() % 16 == 0
            and bool(nat.lib().tsgnn_gatconv_supported(int(conv.heads), int(conv.out_channels)))
            and not (conv.training and conv.dropout > 0.0) and conv.in_channels <= 512)


def _desc(layers, ptrs):
    """layers: [(H, Fin, Co, Ns, w, att_r, att_l)], ptrs: [(wp, gw, gar, gal)] -> host int64 description"""
    words = int(nat.lib().tsgnn_gatconv_pack_desc_words())
    d = np.zeros(1 + len(layers) * words, dtype=np.int64)
    d[0] = len(layers)
    for i, ((H, Fin, Co, Ns, w, ar, al), (wp, gw, gar, gal)) in enumerate(zip(layers, ptrs)):
        o = 1 + i * words
        d[o:o + 4] = (H, Fin, Co, Ns)
        d[o + 4], d[o + 5] = w.data_ptr(), int(w.stride(0))
        d[o + 6], d[o + 7], d[o + 8] = ar.data_ptr(), al.data_ptr(), wp.data_ptr()
        d[o + 9] = gw.data_ptr() if gw is not None else 0
        d[o + 10] = gar.data_ptr() if gar is not None else 0
        d[o + 11] = gal.data_ptr() if gal is not None else 0
    return d


class _Pack(torch.autograd.Function):
    """(lin_l.weight, att_l, att_r) of every layer -> the layers' W' in ONE launch; backward: all parameter gradients from the dW' in ONE."""

    @staticmethod
    def forward(ctx, shapes, *params):
        layers = []
        for i, (H, Co) in enumerate(shapes):
            w, al, ar = params[3 * i].detach(), params[3 * i + 1].detach().contiguous().view(-1), params[3 * i + 2].detach().contiguous().view(-1)
            layers.append((H, int(w.size(1)), Co, packed_width(H, Co), w, ar, al))
        dev = params[0].device
        outs = [_f32(L[1], L[3], device=dev) for L in layers]
        d = _desc(layers, [(o, None, None, None) for o in outs])
        nat.call("gatconv_pack_f32", d.ctypes.data)
        ctx.layers = layers
        ctx.pending = []                      # slab sets the layers' backward leaves for this node to reduce (gat_fused.flush_reductions)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dwps):
        gf.flush_reductions(ctx.pending)
        layers = ctx.layers
        dev = layers[0][4].device
        ptrs = []
        for L, dwp in zip(layers, dwps):
            H, Fin, Co, Ns = L[:4]
            if dwp is None:
                dwp = torch.zeros(Fin, Ns, dtype=torch.float32, device=dev)
            ptrs.append((dwp.contiguous(), _f32(H * Co, Fin, device=dev), _f32(1, H, Co, device=dev), _f32(1, H, Co, device=dev)))
        d = _desc(layers, ptrs)
        nat.call("gatconv_unpack_f32", d.ctypes.data)
        res = []
        for (_, gw, gar, gal) in ptrs:
            res += [gw, gal, gar]
        return (None,) + tuple(res)


def pack_layers(convs):
    """convs: list (<= 4) of pyg.GATConv -> tuple of W' (one per layer)"""
    params = []
    for c in convs:
        params += [c.lin_l.weight, c.att_l, c.att_r]
    outs = _Pack.apply(tuple((int(c.heads), int(c.out_channels)) for c in convs), *params)
    pending = outs[0].grad_fn.pending if (outs and outs[0].grad_fn is not None and hasattr(outs[0].grad_fn, "pending")) else None
    for o in outs:
        o._tsgnn_pending = pending
        o._tsgnn_uses = [0]
    return outs


class _GatConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wp, bias, g, H, Co, slope, mean_heads, apply_elu):
        """x [R, >= Fin] (16-byte rows), wp [Fin, Ns] -> y [R, H * Co] ([R, Co] with mean_heads)"""
        R, Fin, Ns, C = int(x.size(0)), int(wp.size(0)), int(wp.size(1)), H * Co
        dev = x.device
        hp = _f32(R, Ns, device=dev)
        nat.call("rowgemm_f32", x, x.stride(0), wp, wp.stride(0), 0, None, hp, hp.stride(0), None, R, Fin, Ns, 0, 0)
        y = _f32(R, Co if mean_heads else C, device=dev)
        stat = _f32(R, H, 2, device=dev)
        nat.call("gatconv_fwd_f32", hp, hp.stride(0), g.rowptr, g.col, R, H, Co, float(slope), int(mean_heads), int(apply_elu), bias, stat,
                 y, y.stride(0))
        ctx.cfg = (g, H, Co, slope, mean_heads, apply_elu, Fin)
        ctx.pending = getattr(wp, "_tsgnn_pending", None)
        ctx.uses = getattr(wp, "_tsgnn_uses", None)
        if ctx.uses is not None:
            ctx.uses[0] += 1
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, wp, hp, y, stat)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp, hp, y, stat = ctx.saved_tensors
        g, H, Co, slope, mean_heads, apply_elu, Fin = ctx.cfg
        R, Ns, C = int(x.size(0)), int(wp.size(1)), H * Co
        dev = x.device
        dy = dy.contiguous()
        if dy.data_ptr() % 16:
            dy = dy.clone()
        rp_t, col_t, src_e_t = g.transpose_map()
        nnz = max(int(g.nnz), 1)
        dhp = _f32(R, Ns, device=dev)
        dpre = _f32(R, C, device=dev)
        alpha, t1, t2 = _f32(nnz, H, device=dev), _f32(nnz, H, device=dev), _f32(nnz, H, device=dev)
        S = _f32(R, H, device=dev)
        nat.call("gatconv_bwd_rows_f32", hp, hp.stride(0), y, y.stride(0), dy, dy.stride(0), g.rowptr, g.col, R, H, Co, float(slope),
                 int(mean_heads), int(apply_elu), stat, dpre, dpre.stride(0), dhp, Ns, alpha, t1, t2, S)
        # the source side, over A^T: dh_j = sum_i alpha_ij dpre_i (alpha read through the entry map) and d s_src[j] = sum_i t1 - t2 S_i
        nat.call("csr_spmm_heads_epi_f32", rp_t, col_t, alpha, H, Co, dpre, dpre.stride(0), 0, dhp, dhp.stride(0), R,
                 None, None, 0, None, None, 0, None, 0, None, 1, None, 1.0, src_e_t)
        nat.call("gat_score_rowsum_f32", rp_t, col_t, src_e_t, t1, t2, S, R, H, dhp, dhp.stride(0), C + H, None, int(g.B), None, None, None, 1.0)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # bias is added after the aggregation (and after the mean over heads): its gradient is the column sum of the gradient of
            # the pre-activation — dpre [R, C] for concatenated heads; with the mean, of d(out) = dy * ELU'(y) [R, Co] (= sum_h dpre_h)
            if not mean_heads:
                db = mp.colsum(dpre)
            elif not apply_elu:
                db = mp.colsum(dy)
            else:
                db = mp.colsum(dpre).view(H, Co).sum(0)
        want_w = ctx.needs_input_grad[1]
        defer = ctx.pending if (want_w and ctx.pending is not None and ctx.uses is not None and ctx.uses[0] == 1 and not wp.retains_grad) else None
        if ctx.needs_input_grad[0] and want_w and gf.MERGED_BWD_PRODUCTS:
            both = gf.bwd_products(x, Fin, dhp, wp, defer=defer)
            if both is not None:
                return both[1], both[0], db, None, None, None, None, None, None
        dwp = None
        if want_w:
            dwp = gf.wgrad_blocks(x, Fin, dhp, defer=defer)
            if dwp is None:
                dwp = mp.gemm_tn_splitk(x, Fin, dhp)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _f32(R, int(x.size(1)), device=dev)
            if x.size(1) > Fin:
                dx[:, Fin:].zero_()
            nat.call("rowgemm_f32", dhp, dhp.stride(0), wp, wp.stride(0), 1, None, dx, dx.stride(0), None, R, Ns, Fin, 0, 0)
        return dx, dwp, db, None, None, None, None, None, None


def gat_conv(x, wp, bias, g, H, Co, slope=0.2, mean_heads=False, apply_elu=False):
    return _GatConv.apply(x, wp, bias, g, int(H), int(Co), float(slope), bool(mean_heads), bool(apply_elu))
