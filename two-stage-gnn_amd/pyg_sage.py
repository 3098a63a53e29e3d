"""PyG ``SAGEConv`` (and ``GraphConv``) as FUSED launches, and a graph classifier built from them as one autograd node.

BASELINE.json's north_star names the torch_geometric operator surface (SAGEConv / GATConv / SAGPooling / dense_diff_pool) and words its
configs 1-2 with SAGEConv ("MUTAG SAGEConv 2-layer h=64", "PROTEINS SAGEConv 3-layer h=128").  The reference never calls SAGEConv
(PARITY UNPINNED — ``oracle/pyg_ref.py`` restates PyG's documented formula); the one PyG network it does have,
``Code/sag/network.py:9-53``, fixes the shape of a PyG graph classifier in this repository: conv -> ReLU per layer, ``[gmp || gap]`` of every
layer summed, three ``Linear`` layers, ``log_softmax``, ``nll_loss``.  ``SageNet`` is that network with SAGEConv layers and no pooling.

Launches (csrc/sageconv.hip):
  layer forward    ONE launch: neighbour gather + 1/deg + [mean || x] . [W_l ; W_r] + bias (+ ReLU) (+ the layer's max / sum readouts in the
                   epilogue: packed atomicMax + 64-bit fixed-point integer sums, order-independent = bitwise repeatable)
  layer backward   du = (dxs + readout gradients) * [h > 0] (row-wise) -> weight-gradient slabs of both weights -> dx through the SAME fused
                   kernel on the gradient rows (symmetric edge lists: A^T = A; 1/deg moves to the gathered rows)
  all layers       one slab reduction straight into nn.Linear's [out, in] layout (lin_l.weight, lin_l.bias, lin_r.weight)
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import _native as nat
from . import message_passing as mp

_f32 = mp._f32


def conv_ok(K, N):
    return bool(nat.lib().tsgnn_sage_conv_supported(int(K), int(N)))


def inv_degree(g):
    """1 / max(in-degree, 1) per row [R] (PyG's mean aggregation divides by the number of incoming edges), cached on the graph"""
    inv = getattr(g, "_inv_deg_row", None)
    if inv is None:
        inv = g._inv_deg_row = (1.0 / (g.rowptr[1:] - g.rowptr[:-1]).clamp(min=1).to(torch.float32)).contiguous()
    return inv


def ell_of(g, transposed=False):
    """(table, width, tail_ptr, tail_col) of g's neighbour table, or of its transpose (directed edge lists: dx = A^T ...)"""
    if not transposed or g.symmetric:
        ell, W, tail = g.ell()
        return ell, W, (tail[0] if tail is not None else None), (tail[1] if tail is not None else None)
    hit = getattr(g, "_ell_t", None)
    if hit is None:
        from .graph import _i32, exclusive_scan
        rp, col, _ = g.transposed(None)
        R = g.total_rows
        deg = rp[1:] - rp[:-1]
        maxdeg = int(deg.max().item()) if R > 0 else 0
        W = 4 if maxdeg <= 4 else (8 if maxdeg <= 8 else 16)
        ell = _i32(max(R * W, 1), g.device)
        tail_cnt = torch.zeros(R, dtype=torch.int32, device=g.device) if maxdeg > W else None
        nat.call("csr_to_ell", rp, col, R, W, ell, tail_cnt)
        tp = tc = None
        if tail_cnt is not None:
            tp = exclusive_scan(tail_cnt)
            tc = _i32(max(int(tp[-1].item()), 1), g.device)
            nat.call("csr_tail_fill", rp, col, tp, R, W, tc)
        hit = g._ell_t = (ell, W, tp, tc)
    return hit


def _rows16(x):
    """x as a contiguous fp32 matrix whose rows are 16-byte aligned (a width that is no multiple of 4 is padded with zero columns)"""
    x = mp._check(x)
    if x.size(1) % 4 or x.data_ptr() % 16:
        pad = (-x.size(1)) % 4
        x = torch.nn.functional.pad(x, (0, pad)) if pad else x.clone()
    return x


PACK_FLOATS = 16384          # one fragment-major weight copy: [4 waves][16 steps][64 lanes] float4


def pack_weights(items, device):
    """items: list of (weight [N = out, K = in] (nn.Linear), kn) — kn = 0: the forward form (k runs over `in`), kn = 1: the form of the
    input gradient (k runs over `out`).  ONE launch (per 16 matrices); returns a list of fragment-major copies (views of one buffer)."""
    buf = _f32(len(items) * PACK_FLOATS, device=device)
    outs = [buf[i * PACK_FLOATS:(i + 1) * PACK_FLOATS] for i in range(len(items))]
    for c0 in range(0, len(items), 16):
        chunk = items[c0:c0 + 16]
        words = [len(chunk)]
        for j, (w, kn) in enumerate(chunk):
            n_out, k_in = int(w.size(0)), int(w.size(1))
            K, N = (n_out, k_in) if kn else (k_in, n_out)
            words += [w.data_ptr(), int(w.stride(0)), K, N, int(kn), outs[c0 + j].data_ptr()]
        d = np.asarray(words, dtype=np.int64)
        nat.call("sage_conv_pack_f32", d.ctypes.data)
    return outs


def conv_fwd(g, x, wl_pk, wr_pk, bl, K, N, mean, relu_out=False, normalize=False, want_z=True, ro=None):
    """one launch: (out [R, N], z [R, ceil4(K)] or None, rinv or None).  ro = (packed, sums) of this layer: readout epilogue."""
    R = g.total_rows
    ell, W, tp, tc = ell_of(g)
    out = _f32(R, N, device=x.device)
    z = _f32(R, (K + 3) // 4 * 4, device=x.device) if want_z else None
    rinv = _f32(R, device=x.device) if normalize else None
    nat.call("sage_conv_f32", ell, W, tp, tc, x, x.stride(0), x, x.stride(0), inv_degree(g) if mean else None,
             wl_pk, wr_pk, bl, out, out.stride(0), z, z.stride(0) if z is not None else 0, rinv, R, K, N,
             int(relu_out), int(normalize), ro[0] if ro else None, ro[1] if ro else None, g.row_graph if ro else None,
             g.graph_ptr if ro else None, None, 0, None, 0, None, None, None, 0)
    return out, z, rinv


def conv_dx(g, du, dus, wl_pk, wr_pk, K, N, K_out, post=None):
    """dx [R, K_out] = A_mean^T (du W_l) + du W_r through the same fused kernel: the gathered rows are dus = du / deg (rows scaled by
    their OWN degree; dus = du for sum aggregation), the self rows du; wl_pk / wr_pk: the kn = 1 packs; K = the layer's input width,
    N = its output width.  post = (h, dread, arg, want_scaled): the epilogue finishes the dU of the layer below — returns
    (du_below, dus_below or None) instead of dx (tsgnn_sage_conv_f32's post epilogue)."""
    R = g.total_rows
    ell, W, tp, tc = ell_of(g, transposed=True)
    dx = _f32(R, K_out, device=du.device, zero=K_out > K)
    if post is None:
        nat.call("sage_conv_f32", ell, W, tp, tc, dus, dus.stride(0), du, du.stride(0), None,
                 wl_pk, wr_pk, None, dx, dx.stride(0), None, 0, None, R, N, K, 0, 0, None, None, None, None, None, 0, None, 0, None, None, None, 0)
        return dx
    h, dread, arg, want_scaled = post
    dx2 = _f32(R, K_out, device=du.device) if want_scaled else None
    nat.call("sage_conv_f32", ell, W, tp, tc, dus, dus.stride(0), du, du.stride(0), None,
             wl_pk, wr_pk, None, dx, dx.stride(0), None, 0, None, R, N, K, 0, 0, None, None, g.row_graph, g.graph_ptr,
             h, h.stride(0), dread, dread.stride(0), arg, inv_degree(g) if want_scaled else None, dx2, K_out if want_scaled else 0)
    return dx, dx2


SLAB_MERGE = int(os.environ.get("TSGNN_SAGE_SLAB_MERGE", "2"))
WGRAD_PAIR = os.environ.get("TSGNN_SAGE_WGRAD_PAIR", "1") != "0"     # both weights' slabs in one launch
POST_EPILOGUE = os.environ.get("TSGNN_SAGE_POST_EPILOGUE", "1") != "0"   # the dU of the layer below in the input-gradient launch's epilogue


def wgrad_slabs(z, x, K, du):
    """slab partials ((ws_l, nslab), (ws_r, nslab)) of (dW_l^T, db) = (z^T du, colsum du) and dW_r^T = x^T du — one launch
    (tsgnn_sage_wgrad_pair_f32), or two launches of the row-slab kernel; None when the shape is not taken"""
    R, N = int(du.size(0)), int(du.size(1))
    if WGRAD_PAIR and z.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and z.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 \
            and du.data_ptr() % 16 == 0:
        nslab, rps, need = mp.wgrad_plan(R, K, N, z.stride(0), du.stride(0))
        if nslab >= 2 * SLAB_MERGE and SLAB_MERGE > 1:
            # the plan sizes ONE role to the chip; two roles share the launch, so each gets slabs of SLAB_MERGE times the rows: fewer
            # partials to write and to sum (DD b32: 50 MB of slabs across the three layers at 128 slabs per weight)
            rps = rps * SLAB_MERGE
            nslab = -(-R // rps)
            need = nslab * (K + 1) * N
        if nslab > 0:
            ws = _f32(2 * need, device=du.device)
            if nat.try_call("sage_wgrad_pair_f32", z, z.stride(0), x, x.stride(0), du, du.stride(0), R, int(K), N, nslab, rps, ws[:need],
                            ws[need:]):
                return (ws[:need], nslab), (ws[need:], nslab)
    a = mp.linear_wgrad_slabs(z, K, du)
    b = mp.linear_wgrad_slabs(x, K, du)
    if a is None or b is None:
        return None
    return a, b


def reduce_oi(sets, norm_sink=None):
    """sets: list of (ws, nslab, K, N, dw_oi [N, K], db or None) — or (part, nb, 0, F + 4, None, dw_s, F, db_s): the SAGPool score layer's
    partial rows, column sums to dw_s[0:F] and db_s[0]; a ninth entry kn = 1 writes dw as [K, N] (GCNConv's layout) —: ONE launch for up to 12 sets (tsgnn_sage_wgrad_reduce_oi_f32).
    norm_sink: a GradSink whose optimiser wants the |grad|^2 shares of these gradients (and its step counter advanced) from this
    launch; returns True when the shares were left."""
    normed = norm_sink is not None
    for i in range(0, len(sets), 12):
        chunk = sets[i:i + 12]
        words = [len(chunk)]
        for st in chunk:
            ws, nslab, K, N, dw, db = st[:6]
            n_db, tail = (st[6], st[7]) if len(st) > 6 else (N, None)
            kn = int(st[8]) if len(st) > 8 else 0
            words += [ws.data_ptr(), int(nslab), int(K), int(N), dw.data_ptr() if dw is not None else 0,
                      int(dw.stride(0)) if dw is not None else 0, db.data_ptr() if db is not None else 0, int(n_db),
                      tail.data_ptr() if tail is not None else 0, kn]
        d = np.asarray(words, dtype=np.int64)
        parts = step = None
        if norm_sink is not None:
            parts = norm_sink.norm_slots(int(nat.lib().tsgnn_sage_wgrad_reduce_oi_blocks(d.ctypes.data)))
            step = norm_sink.step_state if (parts is not None and not norm_sink.stepped) else None
        normed = normed and parts is not None
        nat.call("sage_wgrad_reduce_oi_f32", d.ctypes.data, parts, step)
        if step is not None:
            norm_sink.stepped = True
    return normed


class _SageConv(torch.autograd.Function):
    """out = lin_l(aggr_j x_j) + lin_r(x_i) [normalised]: one launch forward; backward = slabs x 2, one reduction, one launch for dx"""

    @staticmethod
    def forward(ctx, x, g, wl, bl, wr, mean, normalize):
        wl, wr = wl.contiguous(), wr.contiguous()
        need = any(ctx.needs_input_grad)
        pk = pack_weights([(wl, 0), (wr, 0)], x.device)
        out, z, rinv = conv_fwd(g, x, pk[0], pk[1], bl, int(wl.size(1)), int(wl.size(0)), mean, normalize=normalize, want_z=need)
        ctx.g, ctx.mean, ctx.normalize, ctx.has_bias = g, mean, normalize, bl is not None
        ctx.save_for_backward(x, z, wl, wr, out if normalize else None, rinv)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, z, wl, wr, out, rinv = ctx.saved_tensors
        g = ctx.g
        dout = mp._check(dout)
        R, N, K = dout.size(0), wl.size(0), wl.size(1)
        if ctx.normalize:
            du = torch.empty_like(dout)
            nat.call("l2norm_bwd_f32", out, out.stride(0), dout, dout.stride(0), rinv, du, du.stride(0), R, N)
        else:
            du = dout if dout.data_ptr() % 16 == 0 else dout.clone()
        dwl = dbl = dwr = dx = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[4] or (ctx.has_bias and ctx.needs_input_grad[3]):
            sl = wgrad_slabs(z, x, K, du)
            if sl is not None:
                dwl, dwr = _f32(N, K, device=du.device), _f32(N, K, device=du.device)
                dbl = _f32(N, device=du.device) if ctx.has_bias else None
                reduce_oi([(sl[0][0], sl[0][1], K, N, dwl, dbl), (sl[1][0], sl[1][1], K, N, dwr, None)])
            else:
                dwl_t, dbl = mp.linear_wgrad(z, K, du, ctx.has_bias)
                dwr_t, _ = mp.linear_wgrad(x, K, du, False)
                dwl, dwr = dwl_t.t(), dwr_t.t()
        if ctx.needs_input_grad[0]:
            pk = pack_weights([(wl, 1), (wr, 1)], du.device)
            dus = du * inv_degree(g).unsqueeze(1) if ctx.mean else du
            dx = conv_dx(g, du, dus, pk[0], pk[1], K, N, x.size(1))
        return dx, None, dwl, dbl, dwr, None, None


def sage_conv(x, g, wl, bl, wr, mean=True, normalize=False):
    return _SageConv.apply(_rows16(x), g, wl, bl, wr, bool(mean), bool(normalize))     # (the padding, if any, is autograd's)


# ------------------------------------------------------------------------------------------------ the conv stack as one node
def _ro_workspace(g, L, B, H):
    """packed maxima + integer sums [L, B, H] of a stack on this batch: zero between steps (the decode launch re-zeroes what it read)"""
    key = (L, B, H)
    ws = getattr(g, "_sage_ro_ws", None)
    if ws is None or ws["key"] != key:
        ws = g._sage_ro_ws = {"key": key, "packed": mp.register_clear_on_error(torch.zeros(L * B * H, dtype=torch.int64, device=g.device)),
                              "sums": mp.register_clear_on_error(torch.zeros(L * B * H, dtype=torch.int64, device=g.device)), "dirty": False}
    return ws


class _SageStack(torch.autograd.Function):
    """read[B, 2H] = sum_l [gmp(h_l) || gap(h_l)],  h_l = relu(SAGEConv_l(h_{l-1}))  (network.py:33-46 without the pooling):
    forward(x, g, L, has_bias, *[wl, bl, wr] per layer).  L launches + one decode forward."""

    @staticmethod
    def forward(ctx, x, g, L, has_bias, *params):
        dev = x.device
        B, R = g.B, g.total_rows
        wls = [params[3 * l].contiguous() for l in range(L)]
        bls = [params[3 * l + 1] if has_bias else None for l in range(L)]
        wrs = [params[3 * l + 2].contiguous() for l in range(L)]
        H = int(wls[0].size(0))
        ws = _ro_workspace(g, L, B, H)
        if ws["dirty"]:
            ws["packed"].zero_(); ws["sums"].zero_()
        ws["dirty"] = True
        hs, zs = [x], []
        # fragment-major copies of every weight matrix this step reads: the forward forms of all layers and the input-gradient forms
        # of the layers that have one — ONE launch
        items = [(w, 0) for l in range(L) for w in (wls[l], wrs[l])]
        first_dx = 0 if ctx.needs_input_grad[0] else 1
        items += [(w, 1) for l in range(first_dx, L) for w in (wls[l], wrs[l])]
        pk = pack_weights(items, dev)
        ctx.pk_bwd = {l: (pk[2 * L + 2 * (l - first_dx)], pk[2 * L + 2 * (l - first_dx) + 1]) for l in range(first_dx, L)}
        for l in range(L):
            ro = (ws["packed"][l * B * H:(l + 1) * B * H], ws["sums"][l * B * H:(l + 1) * B * H])
            h, z, _ = conv_fwd(g, hs[-1], pk[2 * l], pk[2 * l + 1], bls[l], int(wls[l].size(1)), H, True, relu_out=True, want_z=True, ro=ro)
            hs.append(h); zs.append(z)
        read = _f32(B, 2 * H, device=dev)
        arg = torch.empty(L * B * H, dtype=torch.int32, device=dev)
        nat.call("sage_readout_decode_f32", ws["packed"], ws["sums"], g.graph_ptr, B, L, H, read, read.stride(0), arg)
        ws["dirty"] = False
        ctx.g, ctx.L, ctx.has_bias, ctx.H = g, L, has_bias, H
        ctx.hs, ctx.zs, ctx.arg, ctx.wls, ctx.wrs = hs, zs, arg, wls, wrs
        ctx.params = params
        return read

    @staticmethod
    def backward(ctx, dread):
        g, L, H = ctx.g, ctx.L, ctx.H
        B, R = g.B, g.total_rows
        dread = dread.contiguous()
        dev = dread.device
        hs, zs = ctx.hs, ctx.zs
        grads = [None] * (3 * L)
        sets = []
        sunk = []
        dxs = None
        dx0 = None
        du = dus = None
        for l in range(L - 1, -1, -1):
            wl, wr = ctx.wls[l], ctx.wrs[l]
            K = int(wl.size(1))
            need_dx = l > 0 or ctx.needs_input_grad[0]
            if du is None:
                # (the last layer, or every layer with TSGNN_SAGE_POST_EPILOGUE=0): the row-wise pass as a launch of its own
                du = _f32(R, H, device=dev)
                dus = _f32(R, H, device=dev) if need_dx else None
                nat.call("sage_relu_readout_bwd_f32", hs[l + 1], hs[l + 1].stride(0), dxs, dxs.stride(0) if dxs is not None else 0, dread,
                         dread.stride(0), ctx.arg[l * B * H:(l + 1) * B * H], g.row_graph, g.graph_ptr, R, H, 1, du, du.stride(0),
                         inv_degree(g) if need_dx else None, dus, H if need_dx else 0)
            sl = wgrad_slabs(zs[l], hs[l], K, du)
            if sl is None:
                raise RuntimeError("SAGEConv stack: weight-gradient shape %d x %d is not taken by the slab kernel" % (K, H))
            dwl, s1 = mp._sink_or_new(ctx.params[3 * l], (H, K), dev)
            dwr, s3 = mp._sink_or_new(ctx.params[3 * l + 2], (H, K), dev)
            dbl, s2 = mp._sink_or_new(ctx.params[3 * l + 1], (H,), dev) if ctx.has_bias else (None, False)
            sets.append((sl[0][0], sl[0][1], K, H, dwl, dbl))
            sets.append((sl[1][0], sl[1][1], K, H, dwr, None))
            grads[3 * l], grads[3 * l + 1], grads[3 * l + 2] = (None if s1 else dwl), (None if s2 else dbl), (None if s3 else dwr)
            sunk.append(s1 and s3 and (s2 or not ctx.has_bias))
            if l > 0:
                if POST_EPILOGUE:
                    # the input gradient of this layer IS (up to the readout terms and the ReLU mask) the dU of the layer below: its
                    # launch's epilogue finishes it — no row-wise launch between the layers
                    below_dx = l - 1 > 0 or ctx.needs_input_grad[0]
                    du, dus = conv_dx(g, du, dus, ctx.pk_bwd[l][0], ctx.pk_bwd[l][1], K, H, K,
                                      post=(hs[l], dread, ctx.arg[(l - 1) * B * H:l * B * H], below_dx))
                    dxs = None
                else:
                    dxs = conv_dx(g, du, dus, ctx.pk_bwd[l][0], ctx.pk_bwd[l][1], K, H, K)
                    du = dus = None
            elif ctx.needs_input_grad[0]:
                dx0 = conv_dx(g, du, dus, ctx.pk_bwd[0][0], ctx.pk_bwd[0][1], K, H, hs[0].size(1))
        sink = mp.GRAD_SINK
        all_sunk = sink is not None and all(sunk)
        if reduce_oi(sets, norm_sink=sink if all_sunk else None):
            for l in range(L):
                sink.normed.add(ctx.params[3 * l].data_ptr()); sink.normed.add(ctx.params[3 * l + 2].data_ptr())
                if ctx.has_bias:
                    sink.normed.add(ctx.params[3 * l + 1].data_ptr())
        return (dx0, None, None, None) + tuple(grads)


def sage_stack(x, g, convs):
    """sum over the layers of [global_max_pool || global_mean_pool] of relu(conv(x)) as one autograd node"""
    has_bias = convs[0].lin_l.bias is not None
    params = []
    for c in convs:
        params += [c.lin_l.weight, c.lin_l.bias if has_bias else c.lin_l.weight.new_zeros(1), c.lin_r.weight]
    return _SageStack.apply(_rows16(x), g, len(convs), has_bias, *params)


def stack_ok(g, convs, x):
    H = convs[0].out_channels
    return (x.is_cuda and g.val is None and g.row_graph is not None and g.n_ghost == 0 and 1 <= len(convs) <= 4 and H % 4 == 0
            and all(c.out_channels == H and conv_ok(c.in_channels, H) and not c.normalize and c.aggr == "mean" and c.root_weight
                    and (c.lin_l.bias is not None) == (convs[0].lin_l.bias is not None) for c in convs)
            and all(c.in_channels == H for c in convs[1:]) and g.symmetric)
