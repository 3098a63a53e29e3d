"""GCN stack of a pooled DiffPool level as ONE autograd node (encoders.py:378-380 -> gcn_forward :140-167 with the dense,
differentiable adjacency A' = S^T A S of encoders.py:375).

The composed form (per layer _DenseBmm -> linear_norm_bn, torch.cat of the layer outputs) leaves autograd to slice the
concatenation's gradient (three strided copies per stack), to add the two gradients every hidden output receives (next layer +
concatenation) and to add the adjacency gradients of the layers (one add per layer): ~9 element-wise launches per stack on
tensors of 16 x 64 x 64 floats.  Here the layers write straight into the concatenation, the backward reads its gradient in place
through leading dimensions (tsgnn_slot_post_bwd_f32 takes the next layer's dx and the direct gradient as two inputs), and the
adjacency gradient is accumulated by the batched product itself.
"""
import torch

from . import _native as nat
from . import message_passing as mp


def _f32(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def eligible(x, adj, g, convs, bn, per_graph_bn):
    if not bn or per_graph_bn or x.dim() != 3 or adj.dim() != 3 or x.dtype != torch.float32 or adj.dtype != torch.float32:
        return False
    if x.size(2) % 4:
        return False
    for c in convs:
        if c.add_self or not c.normalize_embedding or c.dropout > 0.001 or c.output_dim % 4:
            return False
    return all(mp.linear_norm_bn_ok(g, c.output_dim) for c in convs[:-1])


def _affine_norm(z, w, bias, dst, ldd, rinv, R, K, N):
    """dst = normalize(z W + b) rows (encoders.py:36-40)"""
    if mp.rowgemm_ok(z, z.stride(0), w, w.stride(0), K, N, False) and dst.data_ptr() % 16 == 0 and ldd % 4 == 0:
        nat.call("rowgemm_f32", z, z.stride(0), w, w.stride(0), 0, bias, dst, ldd, rinv, R, K, N, 1, 0)
    else:
        nat.call("linear_l2norm_f32", z, z.stride(0), w, w.stride(0), bias, dst, ldd, rinv, R, K, N, 1)


class _DenseGcnStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, g, *params):
        B, K, Fin = x.shape
        R = B * K
        dev = x.device
        ws = [w.contiguous() for w in params[0::2]]
        bs = list(params[1::2])
        L = len(ws)
        widths = [w.size(1) for w in ws]
        offs = [sum(widths[:l]) for l in range(L)]
        total = sum(widths)
        adj = adj.contiguous()
        x2 = x.contiguous().reshape(R, Fin)
        out = _f32(R, total, device=dev)
        aggs, vs, rinvs, stats = [], [], [], []
        xin, ldx, fin = x2, Fin, Fin
        for l in range(L):
            N = widths[l]
            agg = _f32(R, fin, device=dev)                                                   # A_b . x_b  (encoders.py:33)
            mp.gemm(adj, K, 1, xin, ldx, 1, agg, fin, 1, K, fin, K, batch=B, stride_a=K * K, stride_b=K * ldx, stride_c=K * fin)
            dst = out[:, offs[l]:offs[l] + N]
            rinv = _f32(R, device=dev)
            if l == L - 1:                                                                   # conv_last: no ReLU / BN (:165)
                _affine_norm(agg, ws[l], bs[l], dst, total, rinv, R, fin, N)
                vs.append(None); stats.append(None)
            else:
                v = _f32(R, N, device=dev)
                _affine_norm(agg, ws[l], bs[l], v, N, rinv, R, fin, N)
                mean, rstd = _f32(g.nmax, device=dev), _f32(g.nmax, device=dev)
                nat.call("slot_bn_fwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, v, N, N, 1, mean, rstd,
                         dst, total, None, 0)
                vs.append(v); stats.append((mean, rstd))
            aggs.append(agg); rinvs.append(rinv)
            xin, ldx, fin = dst, total, N
        ctx.g, ctx.dims = g, (B, K, Fin, widths, offs, total)
        ctx.has_bias = [b is not None for b in bs]
        ctx.save_for_backward(x2, adj, out, *ws, *aggs, *rinvs, *[t for t in vs if t is not None],
                              *[t for st in stats if st is not None for t in st])
        return out.view(B, K, total)

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        B, K, Fin, widths, offs, total = ctx.dims
        L = len(widths)
        R = B * K
        sv = ctx.saved_tensors
        x2, adj, out = sv[0], sv[1], sv[2]
        ws = sv[3:3 + L]
        aggs = sv[3 + L:3 + 2 * L]
        rinvs = sv[3 + 2 * L:3 + 3 * L]
        vs = sv[3 + 3 * L:3 + 3 * L + (L - 1)]
        st = sv[3 + 3 * L + (L - 1):]
        dev = out.device
        dout = mp._check(dout.reshape(R, total))
        need_adj, need_x = ctx.needs_input_grad[1], ctx.needs_input_grad[0]
        dadj = _f32(B, K, K, device=dev) if need_adj else None
        grads = [None] * (2 * L)
        dx_next = None
        first_adj = True
        slab_sets = []
        for l in range(L - 1, -1, -1):
            N = widths[l]
            fin = Fin if l == 0 else widths[l - 1]
            xin, ldx = (x2, Fin) if l == 0 else (out[:, offs[l - 1]:offs[l - 1] + fin], total)
            dsl = dout[:, offs[l]:offs[l] + N]
            du = _f32(R, N, device=dev)
            if l == L - 1:
                nat.call("l2norm_bwd_f32", out[:, offs[l]:offs[l] + N], total, dsl, total, rinvs[l], du, N, R, N)
            else:
                mean, rstd = st[2 * l], st[2 * l + 1]
                nat.call("slot_post_bwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, vs[l], N, dx_next, N,
                         dsl, total, None, 0, None, N, 1, 1, mean, rstd, rinvs[l], du, N)
            want_w, want_b = ctx.needs_input_grad[3 + 2 * l], ctx.has_bias[l] and ctx.needs_input_grad[4 + 2 * l]
            if want_w:
                got = mp.linear_wgrad_slabs(aggs[l], fin, du) if fin <= 128 and N <= 128 else None
                if got is not None:                      # slab partials now, ONE fixed-order reduction for the whole stack below
                    dw = _f32(fin, N, device=dev)
                    db = _f32(N, device=dev) if want_b else None
                    slab_sets.append((got[0], got[1], fin, N, dw, db))
                    grads[2 * l], grads[2 * l + 1] = dw, db
                else:
                    grads[2 * l], grads[2 * l + 1] = mp.linear_wgrad(aggs[l], fin, du, want_b)
            elif want_b:
                grads[2 * l + 1] = mp.colsum(du)
            if not (need_adj or need_x or l > 0):
                continue
            dagg = _f32(R, fin, device=dev)                                                  # d(A x) = du W^T
            w = ws[l]
            if mp.rowgemm_ok(du, N, w, w.stride(0), N, fin, True):
                nat.call("rowgemm_f32", du, N, w, w.stride(0), 1, None, dagg, fin, None, R, N, fin, 0, 0)
            else:
                mp.gemm(du, N, 1, w, 1, w.stride(0), dagg, fin, 1, R, fin, N)
            if need_adj:                                                                     # dA_b (+)= dagg_b x_b^T
                mp.gemm(dagg, fin, 1, xin, 1, ldx, dadj, K, 1, K, K, fin, batch=B, stride_a=K * fin, stride_b=K * ldx,
                        stride_c=K * K, accumulate=not first_adj)
                first_adj = False
            if l > 0 or need_x:                                                              # dx_b = A_b^T dagg_b
                dxin = _f32(R, fin, device=dev)
                mp.gemm(adj, 1, K, dagg, fin, 1, dxin, fin, 1, K, fin, K, batch=B, stride_a=K * K, stride_b=K * fin,
                        stride_c=K * fin)
                dx_next = dxin
        if slab_sets:
            mp.wgrad_reduce_multi(slab_sets)
        dx = dx_next.view(B, K, Fin) if need_x else None
        return (dx, dadj, None, *grads)


def dense_gcn_stack(x, adj, g, convs):
    """concatenated layer outputs [B, K, sum(widths)] of conv_first / conv_block / conv_last on the pooled level (x, adj)"""
    if ONE_LAUNCH and one_launch_ok(x, adj, [convs]):
        return dense_gcn_stacks(x, adj, [convs])[0]
    params = []
    for c in convs:
        params += [c.weight, c.bias]
    return _DenseGcnStack.apply(x, adj, g, *params)


# ----------------------------------------------------------------------------------------------- one launch per direction
import os

import numpy as np

ONE_LAUNCH = os.environ.get("TSGNN_DENSE_ONE_LAUNCH", "1") != "0"    # pooled-level stacks as one launch forward, one backward
_MAXL = 4
_ws = {}


def _workspace(dev):
    """(barrier words, error word) of the device-wide barriers.  The words are zeroed once and re-armed by the kernels
    themselves; they are keyed by (device, stream) — launches on different streams may overlap and must not count each other's
    arrivals — while the error word is the device's one (message_passing.device_error_word)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    w = _ws.get(key)
    if w is None:
        words = torch.zeros(32 + 256, dtype=torch.int32, device=dev)
        mp.register_barrier_words(dev, words)
        w = _ws[key] = (words, mp.device_error_word(dev))
    return w


def one_launch_ok(x, adj, stacks):
    if x.dim() != 3 or adj.dim() != 3 or x.dtype != torch.float32 or adj.dtype != torch.float32 or not x.is_cuda:
        return False
    if len(stacks) not in (1, 2) or any(len(c) != len(stacks[0]) for c in stacks):
        return False
    B, K, fin0 = x.shape
    L = len(stacks[0])
    hid = stacks[0][0].output_dim
    for convs in stacks:
        for i, c in enumerate(convs):
            if c.add_self or not c.normalize_embedding or c.dropout > 0.001:
                return False
            if i < L - 1 and c.output_dim != hid:
                return False
    lasts = [convs[-1].output_dim for convs in stacks] * 2
    return bool(nat.lib().tsgnn_dense_stack_supported(int(B), int(K), len(stacks), L, int(fin0), int(hid), int(lasts[0]), int(lasts[1])))


def _describe(x2, ldx, fin0, adj, B, K, stacks, stats, ws, bwd=None):
    """the int64 description of tsgnn_dense_stack_*_f32 (layout: csrc/dense_stack.hip::ds_unpack)"""
    P = lambda t: 0 if t is None else int(t.data_ptr())
    d = [P(x2), int(ldx), int(fin0), P(adj), int(B), int(K), len(stacks), P(stats), P(ws[0]), P(ws[1])]
    if bwd is None:
        d += [0, 0, 0, 0, 0, 0, 0, 0, 0]
    else:
        d += [P(bwd["dagg"]), P(bwd["dxn"]), P(bwd["slabs"]), int(bwd["slab_floats"]), int(bwd["finmax"]), P(bwd["dx"]),
              int(bwd["lddx"]), P(bwd["dadj"]), P(bwd["dadj_part"])]
    for si in range(2):
        if si < len(stacks):
            st = stacks[si]
            d += [P(st["out"]), int(st["ldo"]), P(st.get("dout")), int(st.get("lddo", 0)), len(st["layers"])]
            for li in range(_MAXL):
                if li < len(st["layers"]):
                    y = st["layers"][li]
                    d += [P(y["w"]), int(y["ldw"]), P(y["bias"]), int(y["fin"]), int(y["n"]), int(y["off"]), P(y["agg"]), P(y["v"]),
                          P(y["rinv"]), P(y["mean"]), P(y["rstd"]), P(y.get("dw")), P(y.get("db")), int(y.get("slab_off", 0))]
                else:
                    d += [0] * 14
        else:
            d += [0] * (5 + 14 * _MAXL)
    d.append(P(bwd.get("dadj_add")) if bwd is not None else 0)
    return np.asarray(d, dtype=np.int64)


class _DenseGcnStacks(torch.autograd.Function):
    """forward(x, adj, nstack, L, *params): params = per stack, per layer (weight, bias).  Returns one concatenated output per
    stack.  ONE launch forward, ONE backward (tsgnn_dense_stack_fwd_f32 / _bwd_f32)."""

    @staticmethod
    def forward(ctx, x, adj, nstack, L, adj_pass, *params):
        """adj_pass: also return ``adj`` itself as a last differentiable output, for the OTHER consumer of the same adjacency
        (the next level's contraction A'' = S^T A' S): its gradient then arrives here and is summed into this node's dA inside
        the backward launch, instead of a separate element-wise add by autograd"""
        B, K, fin0 = x.shape
        R = B * K
        dev = x.device
        x2 = x.contiguous().reshape(R, fin0)
        adj = adj.contiguous()
        ws = _workspace(dev)
        stats = _f32(_MAXL * nstack * R * 2, device=dev)
        stacks, outs = [], []
        for si in range(nstack):
            layers, off, fin = [], 0, fin0
            widths = [params[(si * L + l) * 2].size(1) for l in range(L)]
            total = sum(widths)
            out = _f32(R, total, device=dev)
            for l in range(L):
                w = params[(si * L + l) * 2].contiguous()
                bias = params[(si * L + l) * 2 + 1]
                n = widths[l]
                last = l == L - 1
                layers.append({"w": w, "ldw": w.stride(0), "bias": bias, "fin": fin, "n": n, "off": off,
                               "agg": _f32(R, fin, device=dev), "v": None if last else _f32(R, n, device=dev),
                               "rinv": _f32(R, device=dev), "mean": None if last else _f32(K, device=dev),
                               "rstd": None if last else _f32(K, device=dev)})
                off += n
                fin = n
            stacks.append({"out": out, "ldo": total, "layers": layers})
            outs.append(out)
        desc = _describe(x2, fin0, fin0, adj, B, K, stacks, stats, ws)       # (kept alive across the call: the kernel arguments are
        nat.call("dense_stack_fwd_f32", desc.ctypes.data)                      # read from it on the host at launch)
        ctx.stacks, ctx.x2, ctx.adj, ctx.dims, ctx.stats, ctx.ws = stacks, x2, adj, (B, K, fin0, nstack, L), stats, ws
        ctx.params = params
        ctx.adj_pass = bool(adj_pass)
        if adj_pass:
            return tuple(o.view(B, K, -1) for o in outs) + (adj.view_as(adj),)
        return tuple(o.view(B, K, -1) for o in outs)

    @staticmethod
    def backward(ctx, *douts):
        B, K, fin0, nstack, L = ctx.dims
        dadj_add = None
        if ctx.adj_pass:
            dadj_add, douts = douts[-1], douts[:-1]
            if dadj_add is not None:
                dadj_add = dadj_add.contiguous()
        R = B * K
        stacks, x2, adj = ctx.stacks, ctx.x2, ctx.adj
        dev = x2.device
        need_x, need_adj = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        finmax = max(max(y["fin"] for y in st["layers"]) for st in stacks)
        slab = 0
        grads = []
        # (work on copies of the layer descriptions: a gradient tensor that stayed referenced from ctx would not be "stolen" by
        # AccumulateGrad but cloned — one copy launch per parameter)
        stacks = [dict(st, layers=[dict(y) for y in st["layers"]]) for st in stacks]
        for si, st in enumerate(stacks):
            d = douts[si]
            d = torch.zeros(R, st["ldo"], device=dev) if d is None else mp._check(d.reshape(R, st["ldo"]))
            st["dout"], st["lddo"] = d, d.stride(0)
        for st in stacks:
            off = 0
            for y in st["layers"]:
                y["slab_off"] = off
                off += (y["fin"] + 1) * y["n"]
                y["dw"] = _f32(y["fin"], y["n"], device=dev)
                y["db"] = _f32(y["n"], device=dev) if y["bias"] is not None else None
                grads += [y["dw"], y["db"]]
            slab = max(slab, off)
        tiles = (K + 15) // 16
        bwd = {"dagg": _f32(nstack * R * finmax, device=dev), "dxn": _f32(nstack * R * finmax, device=dev),
               "slabs": _f32(tiles * B * nstack * slab, device=dev), "slab_floats": slab, "finmax": finmax,
               "dx": _f32(R, fin0, device=dev) if need_x else None, "lddx": fin0,
               "dadj": _f32(B, K, K, device=dev) if need_adj else None,
               "dadj_part": _f32(2 * R * K, device=dev) if (need_adj and nstack == 2) else None,
               "dadj_add": dadj_add if need_adj else None}
        desc = _describe(x2, fin0, fin0, adj, B, K, stacks, ctx.stats, ctx.ws, bwd)
        nat.call("dense_stack_bwd_f32", desc.ctypes.data)
        dx = bwd["dx"].view(B, K, fin0) if need_x else None
        dadj = bwd["dadj"]
        del stacks, bwd, desc
        return (dx, dadj, None, None, None, *grads)


def dense_gcn_stacks(x, adj, stacks, adj_pass=False):
    """the GCN stacks `stacks` (1 or 2 lists of GraphConv modules, same depth) on the pooled level (x[B,K,F], adj[B,K,K]):
    one concatenated output [B, K, sum(widths)] per stack, one launch forward and one backward for all of them"""
    params = []
    for convs in stacks:
        for c in convs:
            params += [c.weight, c.bias]
    return _DenseGcnStacks.apply(x, adj, len(stacks), len(stacks[0]), bool(adj_pass), *params)
