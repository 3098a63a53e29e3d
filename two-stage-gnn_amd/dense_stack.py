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
    params = []
    for c in convs:
        params += [c.weight, c.bias]
    return _DenseGcnStack.apply(x, adj, g, *params)
