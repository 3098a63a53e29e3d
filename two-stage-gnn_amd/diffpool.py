"""DiffPool assignment softmax and the S^T.Z / S^T.A.S contractions (encoders.py:365-376) on
fp32 MFMA (v_mfma_f32_32x32x2_f32 via tsgnn_gemm_f32), forward and explicit backward.

Level-1 contraction works on the ROWS of a GraphBatch: A is sparse there (99.5 % zeros on DD), so
A.S is one SpMM and both S^T.(.) products are ragged batched GEMMs over each graph's row range — the
reference's dense [B,N,N] @ [B,N,K] bmm is never formed.  Pooled levels (adjacency already dense and small)
use strided batched GEMMs.
"""
import os

import torch

from . import _native as nat
from . import message_passing as mp


RAGGED_DIRECT = os.environ.get("TSGNN_RAGGED_DIRECT", "1") != "0"       # first contraction: S^T Z and S^T (A S) as one launch (ragged.hip)
FUSED_CONTRACT = os.environ.get("TSGNN_FUSED_CONTRACT", "1") != "0"     # pooled-level contraction: one launch each way (contract.hip)


def _f32(*shape, device, zero=False):
    return (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=device)


class _RowSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, zero_from):
        x = x.contiguous()
        y = torch.empty_like(x)
        nat.call("row_softmax_masked_fwd_f32", x, x.stride(0), x.size(0), x.size(1), y, y.stride(0), zero_from)
        ctx.save_for_backward(y)
        ctx.zero_from = zero_from
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        nat.call("row_softmax_masked_bwd_f32", y, y.stride(0), dy, dy.stride(0), y.size(0), y.size(1), dx, dx.stride(0), ctx.zero_from)
        return dx, None


def row_softmax(x2d, zero_from=None):
    """nn.Softmax(dim=-1) of a [rows, K] matrix (encoders.py:369); rows >= ``zero_from`` (ghost rows) are zeroed, which is the
    multiplication by the embedding mask that follows it (:370-371) without a second pass."""
    return _RowSoftmax.apply(x2d, int(x2d.size(0) if zero_from is None else zero_from))


# ----------------------------------------------------------------------------- strided batched matmul
def _bmm_raw(a, b, ta, tb, out=None):
    """c[z] = op(a[z]) @ op(b[z]) for contiguous 3-D a, b; ``out`` given: accumulated into it."""
    B = a.size(0)
    M, K = (a.size(2), a.size(1)) if ta else (a.size(1), a.size(2))
    N = b.size(1) if tb else b.size(2)
    c = out if out is not None else _f32(B, M, N, device=a.device)
    sam, sak = (1, a.size(2)) if ta else (a.size(2), 1)
    sbk, sbn = (1, b.size(2)) if tb else (b.size(2), 1)
    mp.gemm(a, sam, sak, b, sbk, sbn, c, N, 1, M, N, K, batch=B, stride_a=a.size(1) * a.size(2),
            stride_b=b.size(1) * b.size(2), stride_c=M * N, accumulate=out is not None)
    return c


class _Bmm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, ta, tb):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        ctx.t = (ta, tb)
        return _bmm_raw(a, b, ta, tb)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        ta, tb = ctx.t
        dc = dc.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            # C = op(A) op(B):  d op(A) = dC op(B)^T ; dA = (d op(A))^T if ta
            da = _bmm_raw(b, dc, tb, True) if ta else _bmm_raw(dc, b, False, not tb)
        if ctx.needs_input_grad[1]:
            db = _bmm_raw(dc, a, True, ta) if tb else _bmm_raw(a, dc, not ta, False)
        return da, db, None, None


def bmm(a, b, trans_a=False, trans_b=False):
    return _Bmm.apply(a, b, bool(trans_a), bool(trans_b))


class _ContractDense(torch.autograd.Function):
    """(S^T Z, S^T A S) of a pooled level as one node: the three gradient contributions to S are accumulated by the
    batched products themselves (composed from bmm nodes autograd adds them with two element-wise launches).
    forward(s, z, adj, ro, softmax) -> xo, ao[, readout of z][, S]:
      ro = (uniform GraphBatch of z's rows, column block or None): the max readout of z (encoders.py:383) is a further output;
      softmax: `s` holds the assignment LOGITS, S = softmax(s, dim=-1) (:369) is formed inside the launch and returned too
      (not differentiable: the node's gradient goes to the logits)."""

    @staticmethod
    def forward(ctx, s, z, adj, ro=None, softmax=False):
        s, z, adj = s.contiguous(), z.contiguous(), adj.contiguous()
        B, N, K, F = s.size(0), s.size(1), s.size(2), z.size(2)
        ctx.fused = bool(FUSED_CONTRACT and s.is_cuda and nat.lib().tsgnn_contract_dense_supported(int(N), int(K), int(F)))
        ctx.ro, ctx.softmax = None, bool(softmax)
        out = arg = None
        if ro is not None:                                     # the max readout of z is part of this node (see diffpool_contract_dense)
            gd, into = ro
            ctx.ro = gd
            ctx.set_materialize_grads(False)
            if ctx.fused:                                      # ... and of its launch: the staged operand is scanned in LDS
                out = into.t if into is not None else _f32(B, F, device=s.device)
                arg = torch.empty(B, F, dtype=torch.int32, device=s.device)
            else:
                out, arg = mp.readout_fwd_raw(z.reshape(B * N, F), gd, into)
        sm = None
        if softmax:
            sm = torch.empty_like(s)
            if not ctx.fused:
                nat.call("row_softmax_masked_fwd_f32", s, K, B * N, K, sm, K, B * N)
        if ctx.fused:                                          # one workgroup per graph, operands in LDS: one launch each way
            xo, ao, t = _f32(B, K, F, device=s.device), _f32(B, K, K, device=s.device), _f32(B, K, N, device=s.device)
            nat.call("contract_dense_fwd_ro_f32", s, z, adj, B, N, K, F, xo, ao, t, out, out.stride(0) if out is not None else 0, arg, sm)
        if softmax:
            s = sm                                             # what the products used, and what the backward needs
        if not ctx.fused:
            xo = _bmm_raw(s, z, True, False)
            t = _bmm_raw(s, adj, True, False)                  # S^T A
            ao = _bmm_raw(t, s, False, False)
        outs = [xo, ao]
        if ro is not None:
            ctx.save_for_backward(s, z, adj, t, arg)
            outs.append(out)
        else:
            ctx.save_for_backward(s, z, adj, t)
        if softmax:
            ctx.mark_non_differentiable(sm)
            outs.append(sm)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dxo, dao, *rest):
        s, z, adj, t = ctx.saved_tensors[:4]
        arg = ctx.saved_tensors[4] if ctx.ro is not None else None
        dro = rest[0] if ctx.ro is not None else None
        B, N, K, F = s.size(0), s.size(1), s.size(2), z.size(2)
        if ctx.ro is not None and (dxo is None or dao is None):   # (materialisation is off for the readout's sake)
            dxo = dxo if dxo is not None else torch.zeros(B, K, F, device=s.device)
            dao = dao if dao is not None else torch.zeros(B, K, K, device=s.device)
        dxo, dao = dxo.contiguous(), dao.contiguous()
        ns, nz, na = ctx.needs_input_grad[:3]
        ds = dz = dadj = None
        if ctx.fused:
            ds = torch.empty_like(s) if ns else None
            dz = torch.empty_like(z) if nz else None
            dadj = torch.empty_like(adj) if na else None
            fold = dro is not None and nz
            if fold:
                dro = mp.readout_dout_in_place(dro)
            if not nat.try_call("contract_dense_bwd_ro_f32", s, z, adj, t, dxo, dao, B, N, K, F, ds, dz, dadj, dro if fold else None,
                                dro.stride(0) if fold else 0, arg if fold else None, 1 if ctx.softmax else 0):
                # (the readout's gradient could not ride in the dz pass: alignment of a gradient slice)
                nat.call("contract_dense_bwd_ro_f32", s, z, adj, t, dxo, dao, B, N, K, F, ds, dz, dadj, None, 0, None, 1 if ctx.softmax else 0)
                if fold:
                    dz = mp.readout_bwd_raw(dro, arg, ctx.ro, B * N, False, dpass=dz.reshape(B * N, F)).reshape(B, N, F)
            return ds, dz, dadj, None, None
        if nz:
            dz = _bmm_raw(s, dxo, False, False)                # X' = S^T Z : dZ = S dX'
        if ns or na:
            dt = _bmm_raw(dao, s, False, True)                 # A' = T S  : dT = dA' S^T
        if ns:
            ds = _bmm_raw(z, dxo, False, True)                 #             dS  = Z dX'^T
            _bmm_raw(t, dao, True, False, out=ds)              #             dS += T^T dA'
            _bmm_raw(adj, dt, False, True, out=ds)             # T = S^T A : dS += A dT^T
            if ctx.softmax:
                dl = torch.empty_like(ds)
                nat.call("row_softmax_masked_bwd_f32", s, K, ds, K, B * N, K, dl, K, B * N)
                ds = dl
        if na:
            dadj = _bmm_raw(s, dt, False, False)               #             dA  = S dT
        if dro is not None and nz:
            dz = mp.readout_bwd_raw(dro, arg, ctx.ro, B * N, False, dpass=dz.reshape(B * N, F)).reshape(B, N, F)
        return ds, dz, dadj, None, None


def diffpool_contract_dense(s, z, adj, readout=None, softmax=False):
    """encoders.py:374-375 on dense [B,N,*] tensors: (S^T Z, S^T A S).  readout = (uniform GraphBatch of z's rows, column block or
    None): the max readout of z over each graph's nodes (:383) becomes a third output of the SAME node, so that its gradient is
    added inside the launch that produces dz (no pass of its own over the rows).  softmax: `s` holds the assignment logits; the
    softmax of :369 is formed inside the launch (and its backward inside the backward launch) and S is returned as the last output."""
    return _ContractDense.apply(s, z, adj, readout, bool(softmax))


# ----------------------------------------------------------------------------- ragged (row-layout) contraction
def _ragged_tn(S, X, g):
    """out[b] = S[rows_b]^T @ X[rows_b]  -> [B, K, F]   (K-ragged batched GEMM)."""
    K, F = S.size(1), X.size(1)
    out = _f32(g.B, K, F, device=S.device)
    tiles = ((K + 31) // 32) * ((F + 31) // 32)
    if (tiles <= 16 and K <= 128 and F <= 256 and F % 4 == 0 and S.stride(0) % 4 == 0 and X.stride(0) % 4 == 0
            and S.data_ptr() % 16 == 0 and X.data_ptr() % 16 == 0 and g.n_rows > 0):
        srp, ssp, nslab = g.row_slabs()
        ws = _f32(nslab * (K + 1) * F, device=S.device)
        nat.call("ragged_tn_f32", S, S.stride(0), X, X.stride(0), K, F, srp, nslab, ssp, g.B, ws, out)
        return out
    mp.gemm(S, 1, S.stride(0), X, X.stride(0), 1, out, F, 1, K, F, 0, batch=g.B, stride_c=K * F, seg_ptr=g.graph_ptr,
            ragged=1, max_seg=int(g.sizes.max()))
    return out


def _ragged_nn(X, Y, g, trans_y, out_cols, rows, out=None):
    """out[rows_b] (+)= X[rows_b] @ op(Y[b])   (M-ragged batched GEMM); rows outside every graph stay zero.
    ``out`` given: the product is accumulated into it (no allocation, no fill, no separate add)."""
    Kd = X.size(1)
    acc = out is not None
    if out is None:
        out = _f32(rows, out_cols, device=X.device, zero=True)
    sbk, sbn = (1, Y.size(2)) if trans_y else (Y.size(2), 1)
    mp.gemm(X, X.stride(0), 1, Y, sbk, sbn, out, out.stride(0), 1, 0, out_cols, Kd, batch=g.B,
            stride_b=Y.size(1) * Y.size(2), seg_ptr=g.graph_ptr, ragged=2, max_seg=int(g.sizes.max()), accumulate=acc)
    return out


def _slabs32(g):
    """(slab_row_ptr, slab_graph, nslab) of 32-row slabs, cached on the graph (row_slabs itself caches one slab size only)"""
    c = getattr(g, "_slabs32", None)
    if c is None:
        import numpy as np
        starts, graphs, off = [], [], 0
        for b, n in enumerate(g.sizes):
            n = int(n)
            for r in range(off, off + n, 32):
                starts.append(r); graphs.append(b)
            off += n
        c = g._slabs32 = (torch.from_numpy(np.asarray(starts + [off], dtype=np.int32)).to(g.device),
                          torch.from_numpy(np.asarray(graphs if graphs else [0], dtype=np.int32)).to(g.device), len(starts))
    return c


class _ContractRows(torch.autograd.Function):
    """X'[b] = S_b^T Z_b ;  A'[b] = S_b^T (A S)_b   over the real rows of every graph."""

    @staticmethod
    def forward(ctx, S, Z, g, ro=None):
        S, Z = S.contiguous(), Z.contiguous()
        ctx.ro = None
        out = arg = None
        K, F = S.size(1), Z.size(1)
        direct = bool(RAGGED_DIRECT and g.n_rows > 0 and nat.lib().tsgnn_ragged_tn_direct_supported(int(K), int(g.sizes.max())))
        ro_in_launch = False
        if ro is not None:                                     # the max readout of Z is part of this node (see diffpool_contract_rows)
            ghost_unused, into, ghost_zero = ro
            ctx.ro = (bool(ghost_unused),)
            ctx.set_materialize_grads(False)
            # ... and of the products' launch when the padded slots' rows are known (none, or zeros)
            ro_in_launch = direct and (g.n_ghost == 0 or bool(ghost_zero))
            if not ro_in_launch:
                out, arg = mp.readout_fwd_raw(Z, g, into)
        AS = mp.spmm_raw(g.rowptr, g.col, g.val, S, g.total_rows)
        xo = ao = None
        if direct:
            # both products in one launch, a workgroup per (32 x 32 output tile, graph): no slabs, no reduction launches (ragged.hip)
            xo, ao = _f32(g.B, K, F, device=S.device), _f32(g.B, K, K, device=S.device)
            if ro_in_launch:
                out = into.t if into is not None else _f32(g.B, F, device=S.device)
                arg = torch.empty(g.B, F, dtype=torch.int32, device=S.device)
            if not nat.try_call("ragged_tn_direct_ro_f32", S, S.stride(0), K, g.graph_ptr, g.B, Z, Z.stride(0), F, xo, AS, AS.stride(0), K, ao,
                                out if ro_in_launch else None, out.stride(0) if ro_in_launch else 0, arg if ro_in_launch else None,
                                g.nmax, g.n_rows, 1 if g.n_ghost else 0):
                xo = ao = None
                if ro_in_launch:
                    out, arg = mp.readout_fwd_raw(Z, g, into)
        if xo is None:
            xo = _ragged_tn(S, Z, g)
            ao = _ragged_tn(S, AS, g)
        ctx.g = g
        if ro is not None:
            ctx.save_for_backward(S, Z, AS, arg)
            return xo, ao, out
        ctx.save_for_backward(S, Z, AS)
        return xo, ao

    @staticmethod
    def backward(ctx, dxo, dao, dro=None):
        S, Z, AS = ctx.saved_tensors[:3]
        arg = ctx.saved_tensors[3] if ctx.ro is not None else None
        g = ctx.g
        R, K, F = S.size(0), S.size(1), Z.size(1)
        if ctx.ro is not None and (dxo is None or dao is None):   # (materialisation is off for the readout's sake)
            dxo = dxo if dxo is not None else torch.zeros(g.B, K, F, device=S.device)
            dao = dao if dao is not None else torch.zeros(g.B, K, K, device=S.device)
        dxo, dao = dxo.contiguous(), dao.contiguous()
        # the readout's gradient: inside the fused launch when the caller discards the ghost rows' share (or there are none)
        ro_fold = dro is not None and (g.n_ghost == 0 or ctx.ro[0])
        if (FUSED_CONTRACT and S.is_cuda and nat.lib().tsgnn_contract_rows_bwd_supported(int(K), int(F)) and Z.stride(0) % 4 == 0
                and S.stride(0) % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in (S, Z, AS, dxo, dao))):
            # the three row-ragged products (+ their zero fills) as one launch: one workgroup per 32-row slab of one graph
            srp, slab_graph, nslab = _slabs32(g)
            dZ, dS, dAS = _f32(R, F, device=S.device), _f32(R, K, device=S.device), _f32(R, K, device=S.device)
            folded = False
            if ro_fold:
                dro_ = mp.readout_dout_in_place(dro)
                folded = bool(dro_.stride(0) % 4 == 0 and dro_.data_ptr() % 16 == 0 and nat.try_call(
                    "contract_rows_bwd_ro_f32", S, S.stride(0), Z, Z.stride(0), AS, AS.stride(0), dxo, dao, srp, slab_graph, nslab, K, F,
                    dZ, dZ.stride(0), dS, dS.stride(0), dAS, dAS.stride(0), g.n_rows, R, dro_, dro_.stride(0), arg))
            if not folded:
                nat.call("contract_rows_bwd_f32", S, S.stride(0), Z, Z.stride(0), AS, AS.stride(0), dxo, dao, srp, slab_graph, nslab, K, F,
                         dZ, dZ.stride(0), dS, dS.stride(0), dAS, dAS.stride(0), g.n_rows, R)
                if dro is not None:
                    dZ = mp.readout_bwd_raw(dro, arg, g, R, ctx.ro[0], dpass=dZ)
            rp, col, val = g.transposed()
            mp.spmm_raw(rp, col, val, dAS, g.total_rows, out=dS, accumulate=True)     # AS = A S  : dS += A^T d(AS)
            return dS, dZ, None, None
        dZ = _ragged_nn(S, dxo, g, False, F, R)                     # dZ_b = S_b dX'_b
        dS = _ragged_nn(Z, dxo, g, True, K, R)                      # S^T Z     : dS_b  = Z_b dX'_b^T
        _ragged_nn(AS, dao, g, True, K, R, out=dS)                  # S^T (AS)  : dS_b += (AS)_b dA'_b^T   (accumulated in place)
        dAS = _ragged_nn(S, dao, g, False, K, R)                    #             d(AS)_b = S_b dA'_b
        rp, col, val = g.transposed()
        mp.spmm_raw(rp, col, val, dAS, g.total_rows, out=dS, accumulate=True)     # AS = A S  : dS += A^T d(AS)
        if dro is not None:
            dZ = mp.readout_bwd_raw(dro, arg, g, R, ctx.ro[0], dpass=dZ)
        return dS, dZ, None, None


def diffpool_contract_rows(S, Z, g, readout=None):
    """readout = (ghost_unused, column block or None, ghost_zero): the max readout of Z (encoders.py:353) as a third output of the
    same node — made by the products' launch when the padded slots' rows are known to be zeros (ghost_zero: masked embeddings) or
    absent, and its gradient is added inside the launch that produces dZ (ghost_unused: the caller discards the ghost rows' share)"""
    return _ContractRows.apply(S, Z, g, readout)


# ----------------------------------------------------------------------------- link-prediction side loss (f4)
class _LinkPredLoss(torch.autograd.Function):
    """encoders.py:416-440: value and d loss / d S from one launch set (csrc/linkpred.hip).  adj_hop > 1 (:419-423):
    pred = sum_{p=1..hop} (S S^T)^p = (S M) S^T with M_b = sum_{p<hop} (S_b^T S_b)^p, a K x K matrix per graph — the pair / edge
    kernels run on the two row operands P = S M and S (one pass per operand), the chain rule through M is K x K algebra."""

    @staticmethod
    def forward(ctx, s, g, clamp, masked, adj_hop):
        import numpy as np
        s = s.contiguous()
        K = s.size(1)
        rows = g.n_rows
        tile = int(nat.lib().tsgnn_linkpred_tile_rows())
        cache = getattr(g, "_lp_slabs", None)
        if cache is None:
            srp, seg, nslab = g.row_slabs(tile)
            counts = np.diff(seg.cpu().numpy())
            slab_graph = torch.from_numpy(np.repeat(np.arange(g.B, dtype=np.int32), counts)).to(g.device)
            cache = g._lp_slabs = (srp, slab_graph, nslab)     # (row_slabs itself re-cuts when asked for another slab size)
        srp, slab_graph, nslab = cache
        sizes = g.sizes.astype(np.float64)
        entries = float((sizes * sizes).sum()) if masked else float(g.nmax) * g.nmax * g.B
        ny = int(nat.lib().tsgnn_linkpred_chunks())
        ws = _f32(ny * rows * K, device=s.device)
        part = _f32(ny * nslab + 2 * ((rows + 3) // 4), device=s.device)
        loss = _f32(1, device=s.device)
        if g.symmetric:
            rp_t = col_t = val_t = None
        else:
            rp_t, col_t, val_t = g.transposed()
        pad = s.size(0) > rows
        if adj_hop == 1:
            ds = _f32(s.size(0), K, device=s.device, zero=pad)
            nat.call("linkpred_loss_f32", s, s.stride(0), K, rows, srp, slab_graph, nslab, g.graph_ptr, g.rowptr, g.col, g.val,
                     rp_t, col_t, val_t, float(clamp), 1.0 / entries, ds, ds.stride(0), ws, part, loss)
        else:
            B, R = g.B, s.size(0)
            G = _ragged_tn(s, s, g)                                            # S_b^T S_b                      [B, K, K]
            pw = [torch.eye(K, device=s.device).expand(B, K, K).contiguous()]
            for _ in range(adj_hop - 1):
                pw.append(torch.bmm(pw[-1], G))                                # G^p  (K x K: library plumbing, not the hot path)
            M = torch.stack(pw).sum(0)
            P = _ragged_nn(s, M, g, False, K, R)                               # P_b = S_b M_b, pred = P S^T    (:419-423)
            dP, dQ = _f32(R, K, device=s.device, zero=pad), _f32(R, K, device=s.device, zero=pad)
            nat.call("linkpred_loss_xy_f32", P, P.stride(0), s, s.stride(0), K, rows, srp, slab_graph, nslab, g.graph_ptr, g.rowptr,
                     g.col, g.val, float(clamp), 1.0 / entries, 1, dP, dP.stride(0), ws, part, loss)
            rt, ct, vt = (g.rowptr, g.col, g.val) if rp_t is None else (rp_t, col_t, val_t)
            nat.call("linkpred_loss_xy_f32", s, s.stride(0), P, P.stride(0), K, rows, srp, slab_graph, nslab, g.graph_ptr, rt, ct, vt,
                     float(clamp), 1.0 / entries, 0, dQ, dQ.stride(0), ws, part, None)
            ds = dQ
            _ragged_nn(dP, M.transpose(1, 2).contiguous(), g, False, K, R, out=ds)      # P = S M   : dS += dP M^T
            dM = _ragged_tn(s, dP, g)                                                    #             dM  = S^T dP
            dG = torch.zeros_like(G)
            for p_ in range(1, adj_hop):                                                 # M = sum_p G^p : dG = sum_p sum_q (G^q)^T dM (G^(p-1-q))^T
                for q in range(p_):
                    dG += torch.bmm(torch.bmm(pw[q].transpose(1, 2), dM), pw[p_ - 1 - q].transpose(1, 2))
            _ragged_nn(s, (dG + dG.transpose(1, 2)).contiguous(), g, False, K, R, out=ds)   # G = S^T S : dS += S (dG + dG^T)
        ctx.save_for_backward(ds)
        return loss.view(())

    @staticmethod
    def backward(ctx, dl):
        (ds,) = ctx.saved_tensors
        return ds * dl, None, None, None, None


def link_pred_loss(s, g, clamp=1.0, masked=True, adj_hop=1):
    """DiffPool's link-prediction side loss on the packed assignment rows s[total_rows, K] of GraphBatch g
    (ghost rows carry zeros and are outside every graph: the reference's adj_mask)."""
    if int(adj_hop) < 1:
        raise ValueError("adj_hop >= 1")
    return _LinkPredLoss.apply(s, g, float(clamp), bool(masked), int(adj_hop))
