"""Differentiable message-passing operators over a GraphBatch (explicit HIP forward AND backward).

Each torch.autograd.Function here is a thin host wrapper around C-ABI kernels of libtsgnn_hip.so
(include/tsgnn.h).  Feature matrices are fp32 ``[g.n_rows + g.n_ghost, F]`` row-major on the GPU.
The reference relies on autograd through dense bmm's (encoders.py:30-42); here every backward is a
hand-written kernel as SURVEY §8(a) requires.
"""
import os

import numpy as np
import torch

from . import _native as nat


def _f32(*shape, device, zero=False):
    return (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=device)


def _check(x, rows=None):
    if not x.is_cuda:
        raise RuntimeError("two_stage_gnn_amd operators run on the GPU only (no CPU fallback)")
    if x.dtype != torch.float32 or x.dim() != 2:
        raise ValueError("expected a float32 [rows, features] matrix")
    if rows is not None and x.size(0) != rows:
        raise ValueError("feature matrix has %d rows, the graph batch has %d" % (x.size(0), rows))
    return x if x.is_contiguous() else x.contiguous()


# ----------------------------------------------------------------------------- aggregation
def spmm_raw(rowptr, col, val, x, n_rows, self_w=None, self_scalar=0.0, relu_in=False, out=None, accumulate=False):
    y = out if out is not None else _f32(x.size(0), x.size(1), device=x.device)
    nat.call("csr_spmm_f32", rowptr, col, val, self_w, x, x.stride(0), y, y.stride(0), int(n_rows), int(x.size(1)),
             float(self_scalar), int(relu_in), int(accumulate))
    return y


ELL_MAX_ROWS = 16384      # the fixed-width table only pays while the batch is cache-resident and launch-bound


def ell_ok(x):
    F = x.size(1)
    return F % 4 == 0 and F <= 256 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0


def spmm_ell(g, x, self_scalar=0.0, out=None, rows=None):
    """unit-weight aggregation through the fixed-width index table (+ CSR tail for rows with > W neighbours).
    ``rows``: only the first ``rows`` output rows are produced (the caller never reads the others)."""
    ell, W, tail = g.ell()
    y = out if out is not None else _f32(x.size(0), x.size(1), device=x.device)
    n = g.total_rows if rows is None else int(rows)
    nat.call("ell_spmm_f32", ell, W, x, x.stride(0), y, y.stride(0), n, int(x.size(1)), float(self_scalar))
    if tail is not None:
        spmm_raw(tail[0], tail[1], None, x, n, out=y, accumulate=True)
    return y


class _Aggregate(torch.autograd.Function):
    """y = A x (+ x)   — GraphConv.forward lines encoders.py:33-35."""

    @staticmethod
    def forward(ctx, x, g, add_self, val, self_w):
        x = _check(x, g.total_rows)
        ctx.g, ctx.add_self, ctx.val, ctx.self_w = g, add_self, val, self_w
        # weights that require grad (PyG GCNConv(edge_weight=) with learnt weights): their gradient is a sampled product of dy and x
        ctx.x = x if ctx.needs_input_grad[3] or ctx.needs_input_grad[4] else None
        ctx.fast = val is None and self_w is None and g.val is None and ell_ok(x) and g.total_rows <= ELL_MAX_ROWS
        if ctx.fast:
            return spmm_ell(g, x, 1.0 if add_self else 0.0)
        return spmm_raw(g.rowptr, g.col, val, x, g.total_rows, self_w=self_w, self_scalar=1.0 if add_self else 0.0)

    @staticmethod
    def backward(ctx, dy):
        g = ctx.g
        dy = _check(dy)
        if ctx.fast and g.symmetric and ell_ok(dy):
            return spmm_ell(g, dy, 1.0 if ctx.add_self else 0.0), None, None, None, None
        val = ctx.val.detach() if ctx.val is not None else None
        self_w = ctx.self_w.detach() if ctx.self_w is not None else None
        rowptr_t, col_t, val_t = g.transposed(val)
        dx = spmm_raw(rowptr_t, col_t, val_t, dy, g.total_rows, self_w=self_w,
                      self_scalar=1.0 if ctx.add_self else 0.0)
        dval = dself = None
        if ctx.x is not None:
            if dy.size(1) > 1024:
                raise NotImplementedError("weight gradients of the aggregation for more than 1024 features")
            dval = _f32(max(g.nnz, 1), device=dy.device)
            dself = _f32(g.total_rows, device=dy.device)
            nat.call("sddmm_rows_f32", g.rowptr, g.col, dy, dy.stride(0), ctx.x, ctx.x.stride(0), g.total_rows, int(dy.size(1)),
                     dval, dself)
            dval = dval[: ctx.val.numel()] if ctx.needs_input_grad[3] else None
            dself = dself if ctx.needs_input_grad[4] else None
        return dx, None, None, dval, dself


def aggregate(x, g, add_self=False, val="graph", self_w=None):
    """Sum-aggregate neighbour rows. ``val``: 'graph' -> g.val (None = unit weights) or an explicit
    per-entry weight tensor aligned with g.col (e.g. GCN-normalised weights)."""
    if isinstance(val, str):
        val = g.val
    return _Aggregate.apply(x, g, bool(add_self), val, self_w)


# ----------------------------------------------------------------------------- transform + L2 normalise
def gemm(A, sam, sak, B, sbk, sbn, C, scm, scn, M, N, K, batch=1, stride_a=0, stride_b=0, stride_c=0,
         seg_ptr=None, ragged=0, max_seg=0, alpha=1.0, accumulate=False):
    nat.call("gemm_f32", A, int(sam), int(sak), B, int(sbk), int(sbn), C, int(scm), int(scn), int(M), int(N), int(K),
             int(batch), int(stride_a), int(stride_b), int(stride_c), seg_ptr, int(ragged), int(max_seg), float(alpha),
             int(accumulate))


def gemm_tn_splitk(Z, K_in, dU, out=None):
    """out[K_in, N] = Z[:, :K_in]^T @ dU   (reduction over graph rows, split-K, reproducible)."""
    R, N = dU.size(0), dU.size(1)
    plan = np.zeros(1, dtype=np.int32)
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("gemm_splitk_plan", int(K_in), int(N), int(R), plan.ctypes.data, need.ctypes.data)
    ws = _f32(max(int(need[0]), 1), device=dU.device)
    out = out if out is not None else _f32(K_in, N, device=dU.device)
    nat.call("gemm_splitk_f32", Z, 1, Z.stride(0), dU, dU.stride(0), 1, out, int(K_in), int(N), int(R), int(plan[0]),
             ws, 0)
    return out


def wgrad_plan(rows, K_in, N, ldz, lddu):
    """(nslab, rows_per_slab, workspace floats) of the weight-gradient slab kernel; nslab = 0: shape unsupported."""
    nslab = np.zeros(1, dtype=np.int32)
    rps = np.zeros(1, dtype=np.int64)
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("linear_wgrad_plan", int(rows), int(K_in), int(N), int(ldz), int(lddu), nslab.ctypes.data, rps.ctypes.data,
                      need.ctypes.data)
    return int(nslab[0]), int(rps[0]), int(need[0])


def linear_wgrad_slabs(z, K_in, du, bias_only_rows=0):
    """slab partials of (dW, db) without the reduction; returns (ws, nslab) or None when the shape is unsupported.
    The last ``bias_only_rows`` rows of du feed db only (ghost rows: their z is zero)."""
    R, N = du.size(0) - int(bias_only_rows), du.size(1)
    nslab = np.zeros(1, dtype=np.int32)
    rps = np.zeros(1, dtype=np.int64)
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("linear_wgrad_plan", int(R), int(K_in), int(N), int(z.stride(0)), int(du.stride(0)), nslab.ctypes.data,
                      rps.ctypes.data, need.ctypes.data)
    if int(nslab[0]) <= 0 or z.data_ptr() % 16 or du.data_ptr() % 16:
        return None
    ws = _f32(int(need[0]), device=du.device)
    nat.call("linear_wgrad_f32", z, z.stride(0), du, du.stride(0), R, int(K_in), int(N), int(nslab[0]), int(rps[0]),
             int(bias_only_rows), ws, None, None)
    return ws, int(nslab[0])


def wgrad_reduce_multi(sets, norm_sink=None):
    """sets: list of (ws, nslab, K, N, dw, db-or-None), at most 4 per launch.  norm_sink: a GradSink whose optimiser wants the
    |grad|^2 shares of these gradients (and its step counter advanced) from this launch."""
    for i in range(0, len(sets), 4):
        args = []
        nblk = 0
        for t in range(4):
            if i + t < len(sets):
                ws, nslab, K, N, dw, db = sets[i + t]
                args += [ws, int(nslab), int(K), int(N), dw, db]
                nblk += ((K + 1) * N + 63) // 64
            else:
                args += [None, 0, 0, 0, None, None]
        parts = norm_sink.norm_slots(nblk) if norm_sink is not None else None
        # the step counter is advanced ONCE per optimiser step, by the first such launch (a model with several fused nodes —
        # DiffPool's paired stacks and pooled levels — reduces weight gradients more than once per backward; every call used to
        # advance it, and Adam's bias correction ran a step ahead: found by tests/test_gpu_fullsize.py)
        step = norm_sink.step_state if (norm_sink is not None and parts is not None and i == 0 and not norm_sink.stepped) else None
        nat.call("wgrad_reduce_multi_f32", *args, parts, step)
        if norm_sink is not None and parts is not None:
            norm_sink.stepped = norm_sink.stepped or step is not None


def linear_wgrad_oi(z, K_in, du, want_db):
    """(dW[N, K_in] — torch.nn.Linear's layout —, db[N] or None) = (du^T z[:, :K_in], colsum(du)): 128 x 128 output blocks in one
    launch + one fixed-order reduction (K_in, N <= 512); None when the shape is not taken"""
    R, N = int(du.size(0)), int(du.size(1))
    nslab = np.zeros(1, dtype=np.int32)
    rps = np.zeros(1, dtype=np.int64)
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("wgrad_blocks_plan", R, int(K_in), N, int(z.stride(0)), int(du.stride(0)), nslab.ctypes.data, rps.ctypes.data,
                      need.ctypes.data)
    if int(nslab[0]) <= 0 or z.data_ptr() % 16 or du.data_ptr() % 16 or z.stride(0) < (int(K_in) + 3) // 4 * 4:
        return None
    ws = _f32(int(need[0]), device=du.device)
    dw = _f32(N, int(K_in), device=du.device)
    db = _f32(N, device=du.device) if want_db else None
    if not nat.try_call("wgrad_blocks_oi_f32", z, z.stride(0), du, du.stride(0), R, int(K_in), N, int(nslab[0]), int(rps[0]), ws, dw,
                        dw.stride(0), db):
        return None
    return dw, db


def linear_wgrad(z, K_in, du, want_db, du_job=None):
    """(dW[K_in,N], db[N] or None) in one pass over the rows; falls back to split-K GEMM + column sums.
    du_job = (part, nb, F, dws, dbs): the partial rows tsgnn_sag_pool_graph_bwd_f32 left behind (called with dws = dbs = NULL);
    their sum rides in this product's reduction launch when the one-pass slab kernel takes the shape, else it is launched here."""
    R, N = du.size(0), du.size(1)
    if du_job is not None:
        part, nb, F_du, dws, dbs = du_job
        nslab = np.zeros(1, dtype=np.int32)
        rps = np.zeros(1, dtype=np.int64)
        need = np.zeros(1, dtype=np.int64)
        nat.call_nostream("linear_wgrad_plan", int(R), int(K_in), int(N), int(z.stride(0)), int(du.stride(0)), nslab.ctypes.data,
                          rps.ctypes.data, need.ctypes.data)
        if int(nslab[0]) > 0 and nb <= 256 and z.data_ptr() % 16 == 0 and du.data_ptr() % 16 == 0:
            ws = _f32(int(need[0]), device=du.device)
            dw = _f32(K_in, N, device=du.device)
            db = _f32(N, device=du.device) if want_db else None
            nat.call("linear_wgrad_du_f32", z, z.stride(0), du, du.stride(0), R, int(K_in), int(N), int(nslab[0]), int(rps[0]), ws,
                     dw, db, part, int(nb), int(F_du), dws, dbs)
            return dw, db
        nat.call("sag_du_reduce_f32", part, int(nb), int(F_du), dws, dbs)
    nslab = np.zeros(1, dtype=np.int32)
    rps = np.zeros(1, dtype=np.int64)
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("linear_wgrad_plan", int(R), int(K_in), int(N), int(z.stride(0)), int(du.stride(0)), nslab.ctypes.data,
                      rps.ctypes.data, need.ctypes.data)
    if int(nslab[0]) > 0 and z.data_ptr() % 16 == 0 and du.data_ptr() % 16 == 0:
        ws = _f32(int(need[0]), device=du.device)
        dw = _f32(K_in, N, device=du.device)
        db = _f32(N, device=du.device) if want_db else None
        nat.call("linear_wgrad_f32", z, z.stride(0), du, du.stride(0), R, int(K_in), int(N), int(nslab[0]), int(rps[0]), 0, ws,
                 dw, db)
        return dw, db
    # wider than one slab pass (<= 128 x 128): the same MFMA slab kernel on 128-wide blocks, ONE fixed-order reduction for all
    # of them — rows of dW when K_in > 128 (GAT layer 2: 256 -> 64), columns when N > 128 through the transposed product
    # dW^T = dU^T Z (GAT layer 1: 92 -> 4 x 64).  The split-K fallback below spends 23 + 17 us on the DD batch for either.
    if z.data_ptr() % 16 == 0 and du.data_ptr() % 16 == 0 and z.stride(0) % 4 == 0 and du.stride(0) % 4 == 0:
        if K_in > 128 and N <= 128 and N % 4 == 0 and K_in % 4 == 0 and K_in <= 512:
            dw = _f32(K_in, N, device=du.device)
            db = _f32(N, device=du.device) if want_db else None
            sets = []
            for k0 in range(0, K_in, 128):
                kc = min(128, K_in - k0)
                got = linear_wgrad_slabs(z[:, k0:], kc, du)
                if got is None:
                    sets = None
                    break
                sets.append((got[0], got[1], kc, N, dw[k0:k0 + kc], db if k0 == 0 else None))
            if sets:
                wgrad_reduce_multi(sets)
                return dw, db
        elif N > 128 and K_in <= 128 and K_in % 4 == 0 and N % 4 == 0 and N <= 512 and z.size(1) >= K_in:
            dwt = _f32(N, K_in, device=du.device)
            sets = []
            for n0 in range(0, N, 128):
                nc = min(128, N - n0)
                got = linear_wgrad_slabs(du[:, n0:], nc, z[:, :K_in] if z.size(1) != K_in else z)
                if got is None:
                    sets = None
                    break
                sets.append((got[0], got[1], nc, K_in, dwt[n0:n0 + nc], None))
            if sets:
                wgrad_reduce_multi(sets)
                return dwt.t(), (colsum(du) if want_db else None)
    if K_in <= 4:                          # narrow input (IMDB's single constant column): dW and db from one pass over du
        need = np.zeros(1, dtype=np.int64)
        nat.call_nostream("wgrad_narrow_plan", int(R), int(K_in), int(N), need.ctypes.data)
        ws = _f32(max(int(need[0]), 1), device=du.device)
        dwb = _f32(K_in + 1, N, device=du.device)
        nat.call("wgrad_narrow_f32", z, z.stride(0), du, du.stride(0), int(R), int(K_in), int(N), ws, dwb)
        return dwb[:K_in], (dwb[K_in] if want_db else None)
    return gemm_tn_splitk(z, K_in, du), (colsum(du) if want_db else None)


def colsum(x):
    R, F = x.size(0), x.size(1)
    out = _f32(F, device=x.device)
    ws = _f32(max(1, (R + 127) // 128) * F, device=x.device)
    nat.call("colsum_f32", x, x.stride(0), int(R), int(F), out, ws, 0)
    return out


def rowgemm_ok(a, lda, b, ldb, K, N, trans_b):
    return bool(nat.lib().tsgnn_rowgemm_supported(a.data_ptr(), int(lda), b.data_ptr(), int(ldb), None, 0, int(K), int(N),
                                                  int(trans_b)))


class _LinearL2Norm(torch.autograd.Function):
    """v = normalize(z W + b)  — encoders.py:36-40."""

    @staticmethod
    def forward(ctx, z, weight, bias, normalize):
        z = _check(z)
        w = weight.contiguous()
        K, N = w.size(0), w.size(1)
        if z.size(1) < K:
            raise ValueError("input has %d features, weight expects %d" % (z.size(1), K))
        R = z.size(0)
        v = _f32(R, N, device=z.device)
        rinv = _f32(R, device=z.device) if normalize else None
        if rowgemm_ok(z, z.stride(0), w, w.stride(0), K, N, False):
            nat.call("rowgemm_f32", z, z.stride(0), w, w.stride(0), 0, bias, v, v.stride(0), rinv, R, K, N, int(normalize), 0)
        else:
            nat.call("linear_l2norm_f32", z, z.stride(0), w, w.stride(0), bias, v, v.stride(0), rinv, R, K, N,
                     int(normalize))
        ctx.save_for_backward(z, w, v if normalize else None, rinv)
        ctx.has_bias = bias is not None
        ctx.normalize = normalize
        return v

    @staticmethod
    def backward(ctx, dv):
        z, w, v, rinv = ctx.saved_tensors
        dv = _check(dv)
        R, N, K = dv.size(0), w.size(1), w.size(0)
        if ctx.normalize:
            du = torch.empty_like(dv)
            nat.call("l2norm_bwd_f32", v, v.stride(0), dv, dv.stride(0), rinv, du, du.stride(0), R, N)
        else:
            du = dv
        dz = dw = db = None
        if ctx.needs_input_grad[0]:
            ldz = z.size(1)
            dz = _f32(R, ldz, device=dv.device, zero=(ldz > K))
            if rowgemm_ok(du, du.stride(0), w, w.stride(0), N, K, True):                     # dZ = dU W^T
                nat.call("rowgemm_f32", du, du.stride(0), w, w.stride(0), 1, None, dz, dz.stride(0), None, R, N, K, 0, 0)
            else:
                gemm(du, du.stride(0), 1, w, 1, w.stride(0), dz, dz.stride(0), 1, R, K, N)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = linear_wgrad(z, K, du, want_db)                                         # dW = Z^T dU (+ db)
        elif want_db:
            db = colsum(du)
        return dz, dw, db, None


def linear_l2norm(z, weight, bias=None, normalize=True):
    return _LinearL2Norm.apply(z, weight, bias, bool(normalize))


class _LinearOI(torch.autograd.Function):
    """y = x W^T + b with W stored [out, in] (torch.nn.Linear's layout) on the fp32 MFMA row-panel kernels: no transposed
    copy of the weight, dW and db from one pass over the rows."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _check(x)
        w = weight.contiguous()
        N, K = w.size(0), w.size(1)
        R = x.size(0)
        y = _f32(R, N, device=x.device)
        if rowgemm_ok(x, x.stride(0), w, w.stride(0), K, N, True):
            nat.call("rowgemm_f32", x, x.stride(0), w, w.stride(0), 1, bias, y, y.stride(0), None, R, K, N, 0, 0)
        else:
            gemm(x, x.stride(0), 1, w, 1, w.stride(0), y, y.stride(0), 1, R, N, K)
            if bias is not None:
                nat.call("broadcast_add_f32", y, y.stride(0), R, 1, N, None, bias, N, None, 0, 0, 1.0)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _check(dy)
        R, N, K = dy.size(0), w.size(0), w.size(1)
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and LINEAR_MERGED_BWD:
            got = linear_bwd_products(x, K, dy, w, want_db)      # (dW, db) slabs and dx = dy W side by side in one launch
            if got is not None:
                return got[2], got[0], got[1]
        if ctx.needs_input_grad[0]:
            dx = _f32(R, K, device=dy.device)
            if rowgemm_ok(dy, dy.stride(0), w, w.stride(0), N, K, False):                   # dX = dY W
                nat.call("rowgemm_f32", dy, dy.stride(0), w, w.stride(0), 0, None, dx, dx.stride(0), None, R, N, K, 0, 0)
            else:
                gemm(dy, dy.stride(0), 1, w, w.stride(0), 1, dx, dx.stride(0), 1, R, K, N)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            got = linear_wgrad_oi(x, K, dy, want_db)       # dW [out, in] contiguous: AccumulateGrad keeps it (a .t() view is copied)
            if got is not None:
                dw, db = got
            else:
                dwt, db = linear_wgrad(x, K, dy, want_db)                                    # (dW)^T = X^T dY [in, out]
                dw = dwt.t()
        elif want_db:
            db = colsum(dy)
        return dx, dw, db


LINEAR_MERGED_BWD = os.environ.get("TSGNN_LINEAR_MERGED_BWD", "1") != "0"   # nn.Linear backward: weight-gradient slabs beside dx = dy W


def linear_bwd_products(x, K_in, dy, w, want_db):
    """(dW[N, K_in], db[N] or None, dx[R, K_in]) of y = x W^T + b in two launches: the blocked weight-gradient slabs beside the
    input-gradient product (tsgnn_linear_bwd_products_f32) and the slabs' reduction; None when the shape is not taken"""
    R, N = int(dy.size(0)), int(dy.size(1))
    if not (R >= 256 and K_in <= 512 and N <= 512 and K_in % 4 == 0 and N % 4 == 0 and x.size(1) == K_in and x.stride(0) % 4 == 0
            and dy.stride(0) % 4 == 0 and w.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0
            and w.data_ptr() % 16 == 0):
        return None
    nslab = np.zeros(1, dtype=np.int32)
    rps = np.zeros(1, dtype=np.int64)
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("wgrad_blocks_plan", R, int(K_in), N, int(x.stride(0)), int(dy.stride(0)), nslab.ctypes.data, rps.ctypes.data,
                      need.ctypes.data)
    if int(nslab[0]) <= 0:
        return None
    ws = _f32(int(need[0]), device=dy.device)
    dx = _f32(R, int(K_in), device=dy.device)
    if not nat.try_call("linear_bwd_products_f32", x, x.stride(0), dy, dy.stride(0), R, int(K_in), N, w, w.stride(0), dx, dx.stride(0),
                        int(nslab[0]), int(rps[0]), ws):
        return None
    dw = _f32(N, int(K_in), device=dy.device)
    db = _f32(N, device=dy.device) if want_db else None
    nat.call("wgrad_blocks_reduce_oi_f32", ws, int(nslab[0]), int(K_in), N, dw, dw.stride(0), db)
    return dw, db, dx


def linear_oi(x, weight, bias=None):
    return _LinearOI.apply(x, weight, bias)


MLP3_TWO_LAUNCH_BWD = True      # rows kernel + weights kernel (False: the single-launch backward, kept for comparison)


class _Mlp3LogSoftmax(torch.autograd.Function):
    """log_softmax(lin3(relu(lin2(dropout(relu(lin1(x))))))) — Code/sag/network.py:48-53 — forward and backward in one launch each"""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, keep, keep_scale, drop=None):
        x = _check(x)
        _w_in = (w1, w2, w3)                              # the parameters as given (the gradient sink is keyed by their storage)
        w1, w2, w3 = w1.contiguous(), w2.contiguous(), w3.contiguous()
        B, D0, D1, D2, C = x.size(0), w1.size(1), w1.size(0), w2.size(0), w3.size(0)
        a1, a2, logp = _f32(B, D1, device=x.device), _f32(B, D2, device=x.device), _f32(B, C, device=x.device)
        if drop is not None:                              # (p, seed, device state, used): the mask is made inside the launch
            p_, seed, state, used = drop
            nat.call("mlp3_fwd_drop_f32", x, x.stride(0), w1, b1, float(p_), int(seed), state, used, w2, b2, w3, b3, B, D0, D1, D2, C,
                     a1, a2, logp)
            keep_scale, keep = 1.0 / (1.0 - float(p_)), True
        else:
            nat.call("mlp3_fwd_f32", x, x.stride(0), w1, b1, keep, float(keep_scale), w2, b2, w3, b3, B, D0, D1, D2, C, a1, a2, logp)
        ctx.save_for_backward(x, w1, w2, w3, a1, a2, logp)
        ctx.keep_scale = float(keep_scale) if keep is not None else 1.0
        ctx.has_b = (b1 is not None, b2 is not None, b3 is not None)
        ctx.params = (_w_in[0], b1, _w_in[1], b2, _w_in[2], b3)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        x, w1, w2, w3, a1, a2, logp = ctx.saved_tensors
        dev = x.device
        B, D0, D1, D2, C = x.size(0), w1.size(1), w1.size(0), w2.size(0), w3.size(0)
        # gradients straight into the trainer's flat bucket when one is installed (no AccumulateGrad copies, no concatenation)
        pw1, pb1, pw2, pb2, pw3, pb3 = ctx.params
        hb = ctx.has_b
        dw1, s1 = _sink_or_new(pw1, (D1, D0), dev)
        dw2, s2 = _sink_or_new(pw2, (D2, D1), dev)
        dw3, s3 = _sink_or_new(pw3, (C, D2), dev)
        db1, t1 = _sink_or_new(pb1, (D1,), dev) if hb[0] else (_f32(D1, device=dev), False)
        db2, t2 = _sink_or_new(pb2, (D2,), dev) if hb[1] else (_f32(D2, device=dev), False)
        db3, t3 = _sink_or_new(pb3, (C,), dev) if hb[2] else (_f32(C, device=dev), False)
        dx = _f32(B, D0, device=dev) if ctx.needs_input_grad[0] else None
        nll = take_deferred_nll(dlogp)                    # deferred F.nll_loss: this backward forms its gradient and the loss
        if nll is None:
            dlogp = dlogp.contiguous()
        parts = None
        if GRAD_SINK is not None and s1 and s2 and s3 and all(hb) and t1 and t2 and t3 and MLP3_TWO_LAUNCH_BWD:
            # ... and their shares of |grad|^2 for the barrier-free optimiser
            parts = GRAD_SINK.norm_slots(int(nat.lib().tsgnn_mlp3_bwd2_norm_blocks(int(D1), int(D2))))
            if parts is not None:
                for p_ in ctx.params:
                    GRAD_SINK.normed.add(p_.data_ptr())
        if parts is not None:
            ws = _f32(B * (C + D2 + D1), device=dev)
            nat.call("mlp3_bwd2_np_f32", x, x.stride(0), w1, w2, w3, a1, a2, logp, None if nll is not None else dlogp,
                     nll[0] if nll is not None else None, nll[1] if nll is not None else None, ctx.keep_scale, B, D0, D1, D2, C,
                     dw1, db1, dw2, db2, dw3, db3, dx, D0, ws, parts)
        elif nll is not None:
            ws = _f32(B * (C + D2 + D1), device=dev)
            nat.call("mlp3_bwd2_nll_f32", x, x.stride(0), w1, w2, w3, a1, a2, logp, nll[0], nll[1], ctx.keep_scale, B, D0, D1, D2, C,
                     dw1, db1, dw2, db2, dw3, db3, dx, D0, ws)
        elif MLP3_TWO_LAUNCH_BWD:
            ws = _f32(B * (C + D2 + D1), device=dev)
            nat.call("mlp3_bwd2_f32", x, x.stride(0), w1, w2, w3, a1, a2, logp, dlogp, ctx.keep_scale, B, D0, D1, D2, C,
                     dw1, db1, dw2, db2, dw3, db3, dx, D0, ws)
        else:
            nat.call("mlp3_bwd_f32", x, x.stride(0), w1, w2, w3, a1, a2, logp, dlogp, ctx.keep_scale, B, D0, D1, D2, C,
                     dw1, db1, dw2, db2, dw3, db3, dx, D0)
        return (dx, None if s1 else dw1, (None if t1 else db1) if hb[0] else None, None if s2 else dw2,
                (None if t2 else db2) if hb[1] else None, None if s3 else dw3, (None if t3 else db3) if hb[2] else None, None, None, None)


_deferred_nll = None


class _NllLoss(torch.autograd.Function):
    """F.nll_loss(logp, label) (mean) on the fused head's output.  Deferred mode (FlatTrainer(defer_loss=True)): nothing is
    launched here; the head's backward forms dlogits = (softmax - onehot) / B itself and writes the loss value."""

    @staticmethod
    def forward(ctx, logp, label):
        loss = _f32(1, device=logp.device)
        ctx.loss = loss
        ctx.save_for_backward(logp, label.contiguous())
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        global _deferred_nll
        logp, label = ctx.saved_tensors
        if g is _unit.get(g.device):
            ph = torch.empty_like(logp)
            _deferred_nll = (ph, label, ctx.loss)
            return ph, None
        B = logp.size(0)                                    # arbitrary upstream gradient: plain formulas, late
        ctx.loss.copy_(-logp.gather(1, label.view(-1, 1)).mean().view(1))
        d = torch.zeros_like(logp)
        d.scatter_(1, label.view(-1, 1), -1.0 / B)
        return d * g, None


class deferred_loss:
    """`with mp.deferred_loss(): loss = model.loss(...) / mp.nll_loss(...); loss.backward(gradient=mp.unit_seed(dev))` — what
    FlatTrainer(defer_loss=True) arranges for its own steps, for a hand-written loop: inside the block a cross-entropy / nll on the
    fused head's output launches nothing and the head's backward forms its gradient and writes the loss value (valid after the
    backward).  The backward MUST be seeded with unit_seed (d loss = 1): any other upstream gradient falls back to the late,
    launch-by-launch formulas."""

    def __enter__(self):
        global CE_DEFER
        self.prev, CE_DEFER = CE_DEFER, True
        return self

    def __exit__(self, *exc):
        global CE_DEFER
        CE_DEFER = self.prev
        return False


def take_deferred_nll(dlogp):
    global _deferred_nll
    d = _deferred_nll
    if d is not None and dlogp is d[0]:
        _deferred_nll = None
        return d[1], d[2]
    return None


def nll_loss(logp, label):
    """F.nll_loss(logp, label); folded into the fused head's backward when a FlatTrainer(defer_loss=True) step is running and
    logp came from mlp3_log_softmax"""
    if CE_DEFER and getattr(logp, "_tsgnn_defer_nll", False) and label.dtype == torch.int64 and logp.is_contiguous():
        return _NllLoss.apply(logp, label)
    return torch.nn.functional.nll_loss(logp, label)


def mlp3_ok(x, lin1, lin2, lin3):
    return bool(x.is_cuda and x.dim() == 2 and lin1.weight.data_ptr() % 16 == 0 and lin2.weight.data_ptr() % 16 == 0
                and nat.lib().tsgnn_mlp3_supported(int(x.size(0)), int(lin1.in_features), int(lin1.out_features),
                                                   int(lin2.out_features), int(lin3.out_features)))


MLP3_DROP_IN_KERNEL = os.environ.get("TSGNN_MLP3_DROP_IN_KERNEL", "1") != "0"
_mlp3_drop = {}             # device -> (process seed, uint64[2] device state of tsgnn_mlp3_fwd_drop_f32)
last_mlp3_dropout = None    # (p, seed, used) of the most recent in-kernel dropout launch: tests regenerate its mask from it


def mlp3_dropout_mask(p, seed, used, B, D1):
    """the 0 / 1 keep mask [B, D1] of the launch that left `used` (tsgnn_mlp3_dropout_mask_f32)"""
    out = _f32(B, D1, device=used.device)
    nat.call("mlp3_dropout_mask_f32", float(p), int(seed), int(used.item()), int(B), int(D1), out)
    return out


def mlp3_log_softmax(x, lin1, lin2, lin3, p=0.0, training=False):
    """the SAGPool head on three nn.Linear modules.  Dropout (network.py:49): the mask is made INSIDE the forward launch (Philox keyed
    on a process seed — drawn from torch's CPU generator at first use, so torch.manual_seed governs it — and a device counter that
    the launch advances itself: a hipGraph replay draws a new mask every time).  TSGNN_MLP3_DROP_IN_KERNEL=0: the mask comes from
    torch's generator instead (one bernoulli launch, and two fill launches per hipGraph replay for its graph-safe state)."""
    global last_mlp3_dropout
    keep, scale, drop = None, 1.0, None
    in_kernel = MLP3_DROP_IN_KERNEL
    if in_kernel and _mlp3_drop.get(x.device) is None and x.is_cuda and torch.cuda.is_current_stream_capturing():
        in_kernel = False       # (the device counter cannot be created inside a capture — its zero fill would replay; torch's mask here)
    if training and p > 0.0 and in_kernel:
        st = _mlp3_drop.get(x.device)
        if st is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            st = _mlp3_drop[x.device] = (seed, torch.zeros(2, dtype=torch.int64, device=x.device))
        used = torch.empty(1, dtype=torch.int64, device=x.device)
        drop = (float(p), st[0], st[1], used)
        last_mlp3_dropout = (float(p), st[0], used)
    elif training and p > 0.0:
        keep = torch.empty(x.size(0), lin1.out_features, dtype=torch.float32, device=x.device).bernoulli_(1.0 - p)
        scale = 1.0 / (1.0 - p)
    logp = _Mlp3LogSoftmax.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, lin3.weight, lin3.bias, keep, scale, drop)
    logp._tsgnn_defer_nll = True          # mp.nll_loss on this output may be folded into the head's backward
    return logp


# ----------------------------------------------------------------------------- ReLU + per-slot batch norm
class _BnSlots(torch.autograd.Function):
    """y = bn_over_slots(relu(v))  — encoders.py:179-181 with apply_bn :134-138 (trap T2)."""

    @staticmethod
    def forward(ctx, v, g, relu, bn):
        v = _check(v, g.total_rows)
        F = v.size(1)
        dev = v.device
        mean = _f32(g.nmax, device=dev) if bn else None
        rstd = _f32(g.nmax, device=dev) if bn else None
        y = torch.empty_like(v)
        nat.call("bn_slots_fwd_f32", g.graph_ptr, g.slot_count, g.row_slot, g.B, g.nmax, g.n_rows, g.n_ghost, v,
                 v.stride(0), F, int(relu), int(bn), mean, rstd, y, y.stride(0))
        ctx.g, ctx.relu, ctx.bn = g, relu, bn
        ctx.save_for_backward(v, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        v, mean, rstd = ctx.saved_tensors
        g = ctx.g
        dy = _check(dy)
        F = v.size(1)
        m1 = _f32(g.nmax, device=v.device) if ctx.bn else None
        m2 = _f32(g.nmax, device=v.device) if ctx.bn else None
        dv = torch.empty_like(v)
        nat.call("bn_slots_bwd_f32", g.graph_ptr, g.slot_count, g.row_slot, g.B, g.nmax, g.n_rows, g.n_ghost, v,
                 v.stride(0), dy, dy.stride(0), F, int(ctx.relu), int(ctx.bn), mean, rstd, m1, m2, dv, dv.stride(0))
        return dv, None, None, None


class _LinearNormBn(torch.autograd.Function):
    """y = slot_bn(relu(normalize(z W + b))) — a hidden GraphConv's transform and the ReLU + apply_bn that follow it
    (encoders.py:36-40,179-181) as one node on the fused slot kernels: BN statistics and normalisation in one launch forward,
    BN + ReLU + L2-normalise backward in one launch (instead of stats / apply, stats / apply and the normalise backward)."""

    @staticmethod
    def forward(ctx, z, weight, bias, g):
        z = _check(z)
        w = weight.contiguous()
        K, N = w.size(0), w.size(1)
        R = z.size(0)
        dev = z.device
        v, rinv = _f32(R, N, device=dev), _f32(R, device=dev)
        if rowgemm_ok(z, z.stride(0), w, w.stride(0), K, N, False):
            nat.call("rowgemm_f32", z, z.stride(0), w, w.stride(0), 0, bias, v, v.stride(0), rinv, R, K, N, 1, 0)
        else:
            nat.call("linear_l2norm_f32", z, z.stride(0), w, w.stride(0), bias, v, v.stride(0), rinv, R, K, N, 1)
        mean, rstd = _f32(g.nmax, device=dev), _f32(g.nmax, device=dev)
        y = torch.empty_like(v)
        nat.call("slot_bn_fwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, v, v.stride(0), N, 1, mean, rstd,
                 y, y.stride(0), None, 0)
        ctx.save_for_backward(z, w, v, rinv, mean, rstd)
        ctx.g, ctx.has_bias = g, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        z, w, v, rinv, mean, rstd = ctx.saved_tensors
        g = ctx.g
        dy = _check(dy)
        R, N, K = dy.size(0), w.size(1), w.size(0)
        du = torch.empty_like(v)
        nat.call("slot_post_bwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, v, v.stride(0), dy, dy.stride(0),
                 None, 0, None, 0, None, N, 1, 1, mean, rstd, rinv, du, du.stride(0))
        dz = dw = db = None
        if ctx.needs_input_grad[0]:
            ldz = z.size(1)
            dz = _f32(R, ldz, device=dy.device, zero=(ldz > K))
            if rowgemm_ok(du, du.stride(0), w, w.stride(0), N, K, True):
                nat.call("rowgemm_f32", du, du.stride(0), w, w.stride(0), 1, None, dz, dz.stride(0), None, R, N, K, 0, 0)
            else:
                gemm(du, du.stride(0), 1, w, 1, w.stride(0), dz, dz.stride(0), 1, R, K, N)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = linear_wgrad(z, K, du, want_db)
        elif want_db:
            db = colsum(du)
        return dz, dw, db, None


def linear_norm_bn_ok(g, n_out):
    """the fused slot kernels take this batch (no ghost rows needed: they handle both layouts)"""
    return bool(nat.lib().tsgnn_slot_fused_supported(int(g.B), int(n_out)))


def linear_norm_bn(z, weight, bias, g):
    return _LinearNormBn.apply(z, weight, bias, g)


class _RowLn(torch.autograd.Function):
    """ReLU + apply_bn with per-graph (B = 1) statistics = per-row layer norm (tripletnet.py:36-38 semantics, batched)."""

    @staticmethod
    def forward(ctx, v, relu):
        v = _check(v)
        R, F = v.shape
        mean, rstd = _f32(R, device=v.device), _f32(R, device=v.device)
        y = torch.empty_like(v)
        nat.call("row_ln_fwd_f32", v, v.stride(0), R, F, int(relu), mean, rstd, y, y.stride(0))
        ctx.relu = relu
        ctx.save_for_backward(v, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        v, mean, rstd = ctx.saved_tensors
        dy = _check(dy)
        dv = torch.empty_like(v)
        nat.call("row_ln_bwd_f32", v, v.stride(0), dy, dy.stride(0), v.size(0), v.size(1), int(ctx.relu), mean, rstd, dv, dv.stride(0))
        return dv, None


def bn_slots(v, g, relu=True, bn=True, per_graph=False):
    """per_graph=True: every graph keeps the statistics it would have alone in its batch (B = 1)."""
    if per_graph and bn:
        return _RowLn.apply(v, bool(relu))
    return _BnSlots.apply(v, g, bool(relu), bool(bn))


# ----------------------------------------------------------------------------- max readout over node slots
def _readout_ws(g, F, dev):
    """workspace of tsgnn_readout_max_fwd_f32: zeroed ONCE per (batch, width, stream) — the launch leaves its counters at zero"""
    cache = g.__dict__.setdefault("_readout_ws", {})
    key = (int(F), torch.cuda.current_stream(dev).cuda_stream)
    ws = cache.get(key)
    if ws is None:
        words = int(nat.lib().tsgnn_readout_max_ws_words(int(g.B), int(g.nmax), int(F)))
        if words <= 0:
            raise ValueError("max readout: batch of %d graphs x %d slots x %d features is out of range" % (g.B, g.nmax, F))
        ws = cache[key] = register_clear_on_error(torch.zeros(words, dtype=torch.int64, device=dev))
    return ws


def readout_fwd_raw(x, g, into=None):
    """(out [B, F], arg [B, F]) of the max readout, no autograd; ``into``: the column block of a ReadoutColumns buffer to write"""
    x = _check(x, g.total_rows)
    F = x.size(1)
    out = into.t if into is not None else _f32(g.B, F, device=x.device)
    arg = torch.empty(g.B, F, dtype=torch.int32, device=x.device)
    nat.call("readout_max_fwd_f32", g.graph_ptr, g.slot_count, g.B, g.nmax, g.n_rows, g.n_ghost, x, x.stride(0),
             F, 0, _readout_ws(g, F, x.device), out, out.stride(0), arg)
    return out, arg


def readout_dout_in_place(dout):
    """the readout's gradient as the kernels read it: a column slice of the concatenated readouts' gradient stays where it is"""
    if dout.stride(1) != 1 or dout.stride(0) < dout.size(1):
        dout = dout.contiguous()
    return dout


def readout_bwd_raw(dout, arg, g, rows, ghost_unused, dpass=None):
    """gradient of the rows [rows, F] from the readout's (dout, arg), summed with ``dpass`` (a second gradient of the same rows)"""
    dout = readout_dout_in_place(dout)
    F = dout.size(1)
    dense_ok = (F % 4 == 0 and dout.stride(0) % 4 == 0 and dout.data_ptr() % 16 == 0 and g.row_graph is not None
                and (g.n_ghost == 0 or ghost_unused) and rows >= g.n_rows)
    if dpass is not None and dense_ok:
        dpass = dpass if (dpass.stride(1) == 1 and dpass.stride(0) % 4 == 0 and dpass.data_ptr() % 16 == 0) else dpass.contiguous()
    if dense_ok:
        dx = _f32(rows, F, device=dout.device)              # one dense pass writes every element
        nat.call("readout_max_bwd_rows_f32", dout, dout.stride(0), arg, g.row_graph, F, g.n_rows, rows, dpass,
                 dpass.stride(0) if dpass is not None else 0, dx, dx.stride(0))
        return dx
    dx = _f32(rows, F, device=dout.device, zero=True)
    nat.call("readout_max_bwd_f32", dout, dout.stride(0), arg, g.B, F, None, 0, 0, g.n_rows, dx, dx.stride(0))
    if dpass is not None:
        dx = dx + dpass
    return dx


class _ReadoutMax(torch.autograd.Function):
    """out[b] = max over ALL nmax node slots of graph b (ghost rows included) — encoders.py:183 (trap T5).
    passthrough: also returns x itself as a second differentiable output for the OTHER consumer of the same tensor (DiffPool's
    contraction reads the embeddings the readout reads): the backward then receives both gradients at once and sums them in the
    pass that scatters the readout's (no zero fill, no scatter launch of its own, no element-wise add by autograd).
    ghost_unused: the caller discards the gradient of the ghost rows (masked embeddings) — the dense pass may be used with them."""

    @staticmethod
    def forward(ctx, x, g, passthrough, ghost_unused, into=None):
        out, arg = readout_fwd_raw(x, g, into)
        ctx.g = g
        ctx.rows = x.size(0)
        ctx.ghost_unused = bool(ghost_unused)
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)
        if passthrough:
            return out, arg, x.view_as(x)
        return out, arg

    @staticmethod
    def backward(ctx, dout, _darg, dpass=None):
        (arg,) = ctx.saved_tensors
        if dout is None:
            return dpass, None, None, None, None
        return readout_bwd_raw(dout, arg, ctx.g, ctx.rows, ctx.ghost_unused, dpass), None, None, None, None


def readout_max(x, g, return_arg=False, into=None):
    out, arg = _ReadoutMax.apply(x, g, False, False, into)
    return (out, arg) if return_arg else out


def readout_max_pass(x, g, ghost_unused=False, into=None):
    """(max readout of x, x): the second output is x for its other consumer; see _ReadoutMax"""
    out, _arg, xp = _ReadoutMax.apply(x, g, True, bool(ghost_unused), into)
    return out, xp


class _Into:
    """a column block of a ReadoutColumns buffer (handed to the readout as a plain object: not an autograd input)"""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class _JoinColumns(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cols, *parts):
        ctx.widths = [p.size(1) for p in parts]
        return cols.buf

    @staticmethod
    def backward(ctx, d):
        out, c0 = [None], 0
        for w in ctx.widths:
            out.append(d[:, c0:c0 + w])              # read in place by the part's backward
            c0 += w
        return tuple(out)


class ReadoutColumns:
    """torch.cat(readouts, dim=1) (encoders.py:203,388-391) without the copy launch: ONE [B, total] buffer, every readout writes its
    column block (`take`), `join` hands the buffer on as the concatenation; backwards, a part's gradient is a column slice of the
    buffer's.  The blocks alias the buffer's storage without being autograd views of it (nothing here is modified in place by
    autograd's bookkeeping: each block is written by exactly one launch)."""

    def __init__(self, B, total, device):
        self.buf = _f32(int(B), int(total), device=device)
        self.c0, self.parts = 0, 0

    def take(self, F):
        """the next F columns (None: they do not fit, or are not 16-byte aligned — the caller then concatenates)"""
        F = int(F)
        if self.c0 < 0 or self.c0 + F > self.buf.size(1) or F % 4 or self.c0 % 4:
            self.c0 = -1
            return None
        t = torch.empty(0, dtype=torch.float32, device=self.buf.device)
        t.set_(self.buf.untyped_storage(), self.buf.storage_offset() + self.c0, (self.buf.size(0), F), (self.buf.stride(0), 1))
        self.c0 += F
        self.parts += 1
        return _Into(t)

    def join(self, parts):
        if self.c0 == self.buf.size(1) and self.parts == len(parts):
            return _JoinColumns.apply(self, *parts)
        return torch.cat(parts, dim=1)


# ----------------------------------------------------------------------------- padded <-> packed rows
class _PackRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, g, ld):
        if not xp.is_cuda:
            raise RuntimeError("two_stage_gnn_amd operators run on the GPU only (no CPU fallback)")
        xp = xp.contiguous().float()
        B, nmax, F = xp.shape
        if B != g.B or nmax != g.nmax:
            raise ValueError("padded features [%d,%d,*] do not match the graph batch [%d,%d]" % (B, nmax, g.B, g.nmax))
        out = _f32(g.total_rows, ld, device=xp.device, zero=True)       # ghost rows = zero padding
        nat.call("pack_rows_f32", xp, nmax, F, g.row_graph, g.row_slot, g.n_rows, out, out.stride(0))
        ctx.g, ctx.F = g, F
        return out

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        dout = _check(dout)
        dxp = _f32(g.B, g.nmax, ctx.F, device=dout.device)
        nat.call("unpack_rows_f32", g.graph_ptr, g.B, g.nmax, g.n_rows, 0, dout, dout.stride(0), ctx.F, 0.0, dxp)
        return dxp, None, None


def pack_rows(x_padded, g, ld=None):
    """[B,Nmax,F] (graph_sampler.py:110-114 layout) -> rows of the batch; ld >= F pads columns with zeros."""
    F = x_padded.size(2)
    if g.layout == "padded":
        if ld not in (None, F):
            raise ValueError("padded layout keeps the feature width")
        x = x_padded.contiguous().float().reshape(g.B * g.nmax, F)
        return x
    return _PackRows.apply(x_padded, g, F if ld is None else int(ld))


class _UnpackRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, fill):
        x = _check(x, g.total_rows)
        F = x.size(1)
        out = _f32(g.B, g.nmax, F, device=x.device)
        nat.call("unpack_rows_f32", g.graph_ptr, g.B, g.nmax, g.n_rows, g.n_ghost, x, x.stride(0), F, float(fill), out)
        ctx.g, ctx.F = g, F
        return out

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        dout = dout.contiguous()
        dx = _f32(g.total_rows, ctx.F, device=dout.device, zero=True)
        nat.call("pack_rows_f32", dout, g.nmax, ctx.F, g.row_graph, g.row_slot, g.n_rows, dx, dx.stride(0))
        if g.n_ghost:
            nat.call("unpack_rows_bwd_ghost_f32", g.graph_ptr, g.B, g.nmax, g.n_rows, dout, ctx.F, dx, dx.stride(0))
        return dx, None, None


def unpack_rows(x, g, fill=0.0):
    """rows of the batch -> [B,Nmax,F] padded tensor (ghost slots take the ghost rows' values)."""
    if g.layout == "padded":
        return x.reshape(g.B, g.nmax, x.size(1))
    return _UnpackRows.apply(x, g, fill)


class _MaskGhost(torch.autograd.Function):
    """x * embedding_mask (encoders.py:165-166): zero every ghost row, pass real rows through."""

    @staticmethod
    def forward(ctx, x, n_real):
        y = x.clone()
        y[n_real:].zero_()
        ctx.n_real = n_real
        return y

    @staticmethod
    def backward(ctx, dy):
        d = dy.clone()
        d[ctx.n_real:].zero_()
        return d, None


def mask_ghost_rows(x, g):
    return _MaskGhost.apply(x, g.n_rows) if g.n_ghost else x


# ----------------------------------------------------------------------------- gradient sink / unit seed
class GradSink:
    """Lets the fused backward nodes write a parameter's gradient straight into its slice of a flat bucket
    (data_parallel.FlatTrainer) instead of handing autograd a fresh tensor that is concatenated afterwards.  A node that
    used the sink returns None for that parameter; ``written`` records which slices were produced this step."""

    def __init__(self):
        self.views = {}             # parameter data_ptr -> view of the flat gradient buffer, shaped like the parameter
        self.written = set()
        # optional: the producers also leave shares of |grad|^2 (and advance the step counter) for a barrier-free optimiser
        self.norm_parts = None      # float32[capacity]
        self.step_state = None
        self.norm_enabled = False   # set per step by the trainer (single GPU only)
        self.ready_cb = None        # trainer hook: called with the parameters whose gradients a launch just finalised
        self.reset_norm()

    def ready(self, params):
        """a fused backward node reports: the launch just issued on the current stream finalised the gradients of `params`
        (all of them written through the sink) — lets the data-parallel trainer start their all-reduce early"""
        if self.ready_cb is not None:
            self.ready_cb(params)

    def reset_norm(self):
        self.norm_used = 0
        self.normed = set()         # parameters whose |grad|^2 is accounted for in norm_parts[:norm_used]
        self.stepped = False
        self.reused = False         # a slice was asked for twice this step (see take)

    def norm_slots(self, n):
        if not self.norm_enabled or self.norm_parts is None or self.norm_used + n > self.norm_parts.numel():
            return None
        v = self.norm_parts[self.norm_used:self.norm_used + n]
        self.norm_used += n
        return v

    def take(self, param, shape):
        """The parameter's slice of the flat bucket — ONCE per step.  The fused backward kernels store ('=') into the slice, so
        a parameter that feeds a second fused node in the same backward (tripletnet.py:36-38 calls the shared encoder three
        times; two forwards of one model under one loss) must not get the slice again: the second node is handed None, returns
        an ordinary gradient tensor, autograd accumulates it in p.grad and FlatTrainer.gather_grads adds p.grad onto the slice."""
        key = param.data_ptr()
        v = self.views.get(key)
        if v is None or tuple(v.shape) != tuple(shape):
            return None
        if key in self.written:
            self.reused = True          # the |grad|^2 shares the first producer left no longer describe the gradient
            return None
        self.written.add(key)
        return v


DEVICE_ERRORS = []                  # (flag tensor, message, state to clear): raised by kernels with a bounded device-wide barrier
_err_words = {}


def _dev_key(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def device_error_word(device):
    """THE error word of a device: one float, zero while all is well.  Kernels with a bounded device-wide barrier store a non-zero
    value when a barrier was not completed (their results are invalid); the optimiser kernels read it as `poison` and skip their
    update while it is set (csrc/optim.hip), and check_device_errors() raises and clears it."""
    device = _dev_key(device)
    w = _err_words.get(device)
    if w is None:
        flag = torch.zeros(1, dtype=torch.float32, device=device)
        w = _err_words[device] = (flag, [])
        DEVICE_ERRORS.append((flag, "a bounded device-wide barrier timed out on %s (a kernel's workgroups were not all resident): the "
                                    "results of that launch are invalid; the optimiser skipped every update since" % (device,), w[1]))
    return w[0]


def register_barrier_words(device, words):
    """barrier words that must be zeroed when the device's error word fires (a failed launch leaves arrival counts behind)"""
    device_error_word(device)
    _err_words[_dev_key(device)][1].append(words)


_clear_on_error = {}                # device -> [weakref to a tensor that a COMPLETED launch leaves zero and an aborted one may not]


def register_clear_on_error(t):
    """Workspaces that launches re-arm themselves (the packed max-readout buffers and ticket counters of tsgnn_readout_max_fwd_f32, the
    integer batch-norm sums / packed maxima of the fused stacks, the readout accumulators of the SAGEConv stack): a launch that does not
    complete — the case the device error word reports — may leave stale values behind, and every later launch on that batch would be
    silently wrong.  Registered here (weakly: the workspace lives and dies with its batch), they are zeroed when check_device_errors()
    finds the error word set."""
    import weakref
    lst = _clear_on_error.setdefault(_dev_key(t.device), [])
    lst.append(weakref.ref(t))
    if len(lst) > 4096:                                     # prune dead entries now and then
        lst[:] = [r for r in lst if r() is not None]
    return t


def check_device_errors(synchronize=True):
    """Raises if a kernel reported a failure it could not return through its status (a bounded device-wide barrier that was not
    completed: the launch's results are invalid).  Reads one float per device that ever armed such a kernel, so it belongs where
    the host synchronises anyway: GraphedStep.synchronize() / loss_value(), IngestPipeline.run(), bench.py after its timed
    region, test fixtures.  Clears the flag and the barrier words of the failed launch (start clean) before raising."""
    for flag, msg, state in DEVICE_ERRORS:
        if float(flag.item()) != 0.0:
            flag.zero_()
            for t in (state if isinstance(state, list) else [state]):
                if t is not None:
                    t.zero_()               # the barrier words of the failed launch
            for ref in _clear_on_error.get(_dev_key(flag.device), []):
                t = ref()
                if t is not None:
                    t.zero_()               # self-clearing workspaces an aborted launch may have left dirty
            raise RuntimeError(msg)


GRAD_SINK = None                    # installed by FlatTrainer between zero_grad() and gather_grads()
_unit = {}


def _sink_or_new(param, shape, device):
    """(buffer, came_from_sink)"""
    v = GRAD_SINK.take(param, shape) if (GRAD_SINK is not None and param is not None) else None
    if v is not None:
        return v, True
    return _f32(*shape, device=device), False


def unit_seed(device):
    """cached scalar 1.0 to seed ``loss.backward(gradient=unit_seed(dev))``: no fill launch, and the fused loss node
    recognises the object and skips the multiply by 1."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:       # the key the backward looks up is a tensor's device ("cuda:0")
        device = torch.device("cuda", torch.cuda.current_device())
    t = _unit.get(device)
    if t is None:
        t = _unit[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


# ----------------------------------------------------------------------------- loss
HEAD_TAIL = os.environ.get("TSGNN_HEAD_TAIL", "1") != "0"     # DiffPool: the last level's max readout inside the head's launches
CE_DEFER = False        # set by FlatTrainer(defer_loss=True) between zero_grad() and the backward: see _SoftmaxCE
_deferred_ce = None     # (placeholder gradient, logits, label, loss) handed from _SoftmaxCE.backward to the head node's backward


class _SoftmaxCE(torch.autograd.Function):
    """F.cross_entropy(pred, label) (encoders.py:221-224) with the logits gradient produced in the same launch.

    Deferred mode (trainer option, logits produced by the fused stack + head node): the forward launches nothing and the head's
    backward kernel rebuilds the logits gradient itself and writes the loss value (tsgnn_head2_bwd_ce_f32) — one launch less
    per step; the returned loss tensor holds its value once the backward has run."""

    @staticmethod
    def forward(ctx, logits, label, defer=False):
        B, C = logits.shape
        ctx.deferred = bool(defer and logits.is_contiguous() and logits.dtype == torch.float32)
        loss = _f32(1, device=logits.device)
        if ctx.deferred:
            ctx.loss = loss
            ctx.save_for_backward(logits, label.contiguous())
            return loss[0]
        logits = logits.contiguous().float()
        dlogits = _f32(B, C, device=logits.device)
        nat.call("softmax_ce_f32", logits, logits.stride(0), label.contiguous(), B, C, loss, dlogits)
        ctx.save_for_backward(dlogits)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        global _deferred_ce
        if ctx.deferred:
            logits, label = ctx.saved_tensors
            if g is _unit.get(g.device):
                ph = torch.empty_like(logits)               # never read: the head's backward recognises it and takes over
                _deferred_ce = (ph, logits, label, ctx.loss)
                return ph, None, None
            B, C = logits.shape                             # arbitrary upstream gradient: the ordinary kernel, late
            dlogits = _f32(B, C, device=logits.device)
            nat.call("softmax_ce_f32", logits, logits.stride(0), label, B, C, ctx.loss, dlogits)
            return dlogits * g, None, None
        (dlogits,) = ctx.saved_tensors
        if g is _unit.get(g.device):                        # seeded with unit_seed(): d loss = 1 by construction
            return dlogits, None, None
        return dlogits * g, None, None


def take_deferred_ce(dy):
    """(logits, label, loss) if dy is the placeholder a deferred cross-entropy left for the head's backward, else None"""
    global _deferred_ce
    d = _deferred_ce
    if d is not None and dy is d[0]:
        _deferred_ce = None
        return d[1], d[2], d[3]
    return None


def cross_entropy(logits, label):
    if logits.is_cuda and label.dtype == torch.int64:
        return _SoftmaxCE.apply(logits, label, bool(CE_DEFER and getattr(logits, "_tsgnn_defer_ce", False)))
    return torch.nn.functional.cross_entropy(logits, label, reduction="mean")


# ----------------------------------------------------------------------------- graph-level head
class _Head2(torch.autograd.Function):
    """(vec, y) = (W1 out + b1, W2 vec + b2): the chained nn.Linear pair after the readout, 1 + 1 launches."""

    @staticmethod
    def forward(ctx, out, w1, b1, w2, b2):
        out = out.contiguous()
        w1c, w2c = w1.contiguous(), w2.contiguous()
        B, P = out.shape
        E, C = w1c.size(0), w2c.size(0)
        vec = _f32(B, E, device=out.device)
        y = _f32(B, C, device=out.device)
        nat.call("head2_fwd_f32", out, out.stride(0), w1c, b1, w2c, b2, B, P, E, C, vec, y)
        ctx.save_for_backward(out, w1c, w2c, vec)
        ctx.has_b = (b1 is not None, b2 is not None)
        ctx.params = (w1, b1, w2, b2)
        ctx.set_materialize_grads(False)                    # an unused output must not cost a zero-fill launch
        return vec, y

    @staticmethod
    def backward(ctx, dvec, dy):
        out, w1, w2, vec = ctx.saved_tensors
        B, P = out.shape
        E, C = w1.size(0), w2.size(0)
        dev = out.device
        if dy is None and dvec is None:
            return None, None, None, None, None
        ce = take_deferred_ce(dy) if dy is not None else None      # deferred cross-entropy: this backward rebuilds dy
        if ce is None:
            dy = dy.contiguous() if dy is not None else torch.zeros(B, C, device=dev)
        dvec = dvec.contiguous() if dvec is not None else None
        dout = _f32(B, P, device=dev)
        pw1, pb1, pw2, pb2 = ctx.params
        dw1, s1 = _sink_or_new(pw1, (E, P), dev)
        dw2, s2 = _sink_or_new(pw2, (C, E), dev)
        db1, s3 = _sink_or_new(pb1, (E,), dev) if ctx.has_b[0] else (None, False)
        db2, s4 = _sink_or_new(pb2, (C,), dev) if ctx.has_b[1] else (None, False)
        parts = head_norm_slots((s1, s2, s3, s4), ctx.has_b, (pw1, pb1, pw2, pb2), E)
        if ce is not None:
            nat.call("head2_bwd_ce_f32", out, out.stride(0), vec, ce[0], ce[1], ce[2], dvec, w1, w2, B, P, E, C, dout, dout.stride(0),
                     dw1, db1, dw2, db2, parts)
        else:
            nat.call("head2_bwd_f32", out, out.stride(0), vec, dy, dvec, w1, w2, B, P, E, C, dout, dout.stride(0), dw1, db1, dw2, db2, parts)
        return dout, None if s1 else dw1, None if s3 else db1, None if s2 else dw2, None if s4 else db2


class _Head2Tail(torch.autograd.Function):
    """_Head2 whose launches also make the max readout of the LAST uniform level: forward(cols, z, N, w1, b1, w2, b2, *parts) — `cols`
    is the ReadoutColumns buffer whose leading column blocks `parts` were written by their readouts; the launch fills the remaining
    F = z.size(1) columns with max_n z[b * N + n, :] and runs the head; the backward's launch writes dz (tsgnn_head2_*_ro_f32)."""

    @staticmethod
    def forward(ctx, cols, z, N, w1, b1, w2, b2, *parts):
        out = cols.buf
        w1c, w2c = w1.contiguous(), w2.contiguous()
        z = _check(z)
        B, P = out.shape
        F = z.size(1)
        E, C = w1c.size(0), w2c.size(0)
        widths = [p_.size(1) for p_ in parts]
        c0 = sum(widths)
        if c0 + F != P or z.size(0) != B * N:
            raise ValueError("head tail: %d + %d columns for a %d-wide input, %d rows for %d graphs x %d" % (c0, F, P, z.size(0), B, N))
        vec, y = _f32(B, E, device=out.device), _f32(B, C, device=out.device)
        arg = torch.empty(B, F, dtype=torch.int32, device=out.device)
        nat.call("head2_fwd_ro_f32", out, out.stride(0), w1c, b1, w2c, b2, B, P, E, C, vec, y, z, z.stride(0), int(N), c0, F, arg)
        ctx.save_for_backward(out, w1c, w2c, vec, arg)
        ctx.has_b = (b1 is not None, b2 is not None)
        ctx.params = (w1, b1, w2, b2)
        ctx.geom = (int(N), c0, F, widths)
        ctx.set_materialize_grads(False)
        return vec, y

    @staticmethod
    def backward(ctx, dvec, dy):
        out, w1, w2, vec, arg = ctx.saved_tensors
        N, c0, F, widths = ctx.geom
        B, P = out.shape
        E, C = w1.size(0), w2.size(0)
        dev = out.device
        if dy is None and dvec is None:
            return (None,) * (7 + len(widths))
        ce = take_deferred_ce(dy) if dy is not None else None      # deferred cross-entropy: this backward rebuilds dy
        if ce is None:
            dy = dy.contiguous() if dy is not None else torch.zeros(B, C, device=dev)
        dvec = dvec.contiguous() if dvec is not None else None
        dout = _f32(B, P, device=dev)
        dz = _f32(B * N, F, device=dev)
        pw1, pb1, pw2, pb2 = ctx.params
        dw1, s1 = _sink_or_new(pw1, (E, P), dev)
        dw2, s2 = _sink_or_new(pw2, (C, E), dev)
        db1, s3 = _sink_or_new(pb1, (E,), dev) if ctx.has_b[0] else (None, False)
        db2, s4 = _sink_or_new(pb2, (C,), dev) if ctx.has_b[1] else (None, False)
        parts = head_norm_slots((s1, s2, s3, s4), ctx.has_b, (pw1, pb1, pw2, pb2), E)
        cy, cl, closs = ce if ce is not None else (None, None, None)
        if not nat.try_call("head2_bwd_ro_f32", out, out.stride(0), vec, None if ce is not None else dy, dvec, w1, w2, B, P, E, C, dout,
                            dout.stride(0), dw1, db1, dw2, db2, parts, arg, c0, F, N, dz, dz.stride(0), cy, cl, closs):
            if ce is not None:
                nat.call("head2_bwd_ce_f32", out, out.stride(0), vec, cy, cl, closs, dvec, w1, w2, B, P, E, C, dout, dout.stride(0),
                         dw1, db1, dw2, db2, parts)
            else:
                nat.call("head2_bwd_f32", out, out.stride(0), vec, dy, dvec, w1, w2, B, P, E, C, dout, dout.stride(0), dw1, db1, dw2, db2,
                         parts)
            nat.call("readout_max_bwd_f32", dout[:, c0:], dout.stride(0), arg, B, F, None, 0, 0, B * N, dz.zero_(), dz.stride(0))
        grads, o = [], 0
        for w in widths:
            grads.append(dout[:, o:o + w])                # read in place by the part's backward
            o += w
        return (None, dz, None, None if s1 else dw1, None if s3 else db1, None if s2 else dw2, None if s4 else db2) + tuple(grads)


def head2_tail_ok(cols, z, N, lin1, lin2):
    """the last level's readout may ride in the head's launches (tsgnn_head2_fwd_ro_f32 / _bwd_ro_f32)"""
    return (HEAD_TAIL and cols is not None and cols.c0 >= 0 and cols.c0 + z.size(-1) == cols.buf.size(1) and int(N) <= 64
            and z.size(-1) % 4 == 0 and head2_ok(cols.buf, lin1, lin2) and cols.buf.size(1) // 4 <= 512)


def head2_tail(cols, parts, z, N, lin1, lin2):
    """(lin1(out), lin2(lin1(out))) with out = [parts | max readout of the uniform level z (N rows per graph)]"""
    vec, y = _Head2Tail.apply(cols, z, int(N), lin1.weight, lin1.bias, lin2.weight, lin2.bias, *parts)
    y._tsgnn_defer_ce = True              # a cross-entropy on these logits may be folded into this node's backward (mp._SoftmaxCE)
    return vec, y


def head_norm_slots(sunk, has_b, params, E):
    """|grad|^2 shares of the head's gradients for the installed sink's optimiser, when ALL of them went to the sink"""
    s1, s2, s3, s4 = sunk
    if GRAD_SINK is None or not (s1 and s2 and (s3 or not has_b[0]) and (s4 or not has_b[1])):
        return None
    parts = GRAD_SINK.norm_slots((E + 3) // 4 + 1)
    if parts is not None:
        for p_ in params:
            if p_ is not None:
                GRAD_SINK.normed.add(p_.data_ptr())
    return parts


def head2_ok(out, lin1, lin2):
    return (out.is_cuda and isinstance(lin1, torch.nn.Linear) and isinstance(lin2, torch.nn.Linear) and out.dim() == 2
            and out.size(1) % 4 == 0 and out.size(1) <= 2048 and lin1.out_features <= 4096 and out.size(0) <= 1024
            and lin1.weight.data_ptr() % 16 == 0)


def head2(out, lin1, lin2):
    """lin2(lin1(out)) returning (lin1 output, lin2 output)."""
    vec, y = _Head2.apply(out, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    y._tsgnn_defer_ce = True              # a cross-entropy on these logits may be folded into this node's backward (mp._SoftmaxCE)
    return vec, y
