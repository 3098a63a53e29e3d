"""Edge-softmax attention aggregation (forward + explicit backward) over a GraphBatch.

Two normalisation modes, same kernels (csrc/attention.hip):
  * ``by_column=True``  — the reference's DGATHead (encoders_GAT.py:29-49): softmax over the ROW index i
    for every column j (trap T3); columns without any edge become uniform 1/N and add (1/N)*h_j to every
    row of their graph (the -9e15 mask makes an all-masked column's softmax uniform, :38-41).
  * ``by_column=False`` — standard per-target softmax (PyG GATConv).
"""
import torch

from . import _native as nat


def _f32(*shape, device, zero=False):
    return (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=device)


def node_scores(h, a, H, Fh):
    s = _f32(h.size(0), H, device=h.device)
    nat.call("node_scores_f32", h, h.stride(0), h.size(0), H, Fh, a, a.stride(0), s)
    return s


def node_scores2(h, a1, a2, H, Fh):
    """(h . a1, h . a2) per node and head from one pass over the rows"""
    s1, s2 = _f32(h.size(0), H, device=h.device), _f32(h.size(0), H, device=h.device)
    nat.call("node_scores2_f32", h, h.stride(0), h.size(0), H, Fh, a1, a1.stride(0), s1, a2, a2.stride(0), s2)
    return s1, s2


def segment_wsum2(x, w1, w2, H, Fh):
    """(sum_r w1[r,h] x[r,:], sum_r w2[r,h] x[r,:]) over all rows, x read once"""
    out1, out2 = _f32(1, H * Fh, device=x.device), _f32(1, H * Fh, device=x.device)
    nchunk = max(1, (int(x.size(0)) + 127) // 128)
    ws = _f32(2 * nchunk * H * Fh, device=x.device)
    nat.call("segment_wsum2_f32", x, x.stride(0), w1, w2, H, Fh, None, 1, x.size(0), int(x.size(0)), 1.0, ws, out1, out2, out1.stride(0))
    return out1, out2


def segment_wsum(x, w, H, Fh, seg_ptr, nseg, scale=1.0, mean=False, max_seg=None):
    """max_seg: longest segment (host int); defaults to all rows (always a valid bound)."""
    out = _f32(nseg, H * Fh, device=x.device)
    max_seg = int(x.size(0) if max_seg is None else max_seg)
    nchunk = max(1, (max_seg + 127) // 128)
    ws = _f32(nchunk * nseg * H * Fh, device=x.device)
    nat.call("segment_wsum_f32", x, x.stride(0), w, H, Fh, seg_ptr, nseg, x.size(0), max_seg, float(scale), int(mean), ws, out,
             out.stride(0))
    return out


def _row_seg(g):
    """row -> graph map for ragged batches (None: every graph has g.nmax rows, the kernels divide)"""
    return g.row_graph if getattr(g, "row_mult", None) is not None else None


def _inverse_entry_map(g, src_e_t):
    """inv[e] = p with src_e_t[p] = e: where A's entry e sits in A^T's entry order (cached on the graph)"""
    inv = getattr(g, "_inv_e_t", None)
    if inv is None:
        n = src_e_t.numel()
        inv = torch.empty(n, dtype=torch.int32, device=src_e_t.device)
        inv[src_e_t.long()] = torch.arange(n, dtype=torch.int32, device=src_e_t.device)
        g._inv_e_t = inv
    return inv


def _isolated_list(g, iso, force=False):
    """(row ids, weights, per-graph pointer) of the rows with a non-zero uniform-term weight — a few per graph; built once
    per graph (one host round trip at graph build time) so that the forward sums those rows only.  None: nothing to list, or
    (unless ``force``) so many that the all-rows scan is the better kernel."""
    c = getattr(g, "_iso_list", None)
    if c is None:
        w = iso[:, 0]
        idx = torch.nonzero(w).view(-1)
        rows_graph = g.row_graph[: g.n_rows].long() if g.n_ghost == 0 else None
        if rows_graph is None:                              # per-slot ghost layout: not used with the uniform term
            rows_graph = torch.div(torch.arange(w.numel(), device=w.device), max(g.nmax, 1), rounding_mode="floor")
        cnt = torch.bincount(rows_graph[idx], minlength=g.B)
        ptr = torch.zeros(g.B + 1, dtype=torch.int32, device=w.device)
        ptr[1:] = torch.cumsum(cnt, 0)
        n_listed = int(idx.numel())
        c = g._iso_list = (idx.to(torch.int32).contiguous(), w[idx].contiguous(), ptr, n_listed)
    if c[3] == 0 or (c[3] > 64 * g.B and not force):
        return None
    return c[:3]


def isolated_count(g):
    """number of listed edge-less columns (after _isolated_list has run on g)"""
    return g._iso_list[3]


def _isolated_columns(g, rp_t, R, H):
    """[R, H] indicator of columns without any edge; depends on the graph only, cached on it"""
    cache = g.__dict__.setdefault("_iso_cols", {})
    iso = cache.get(H)
    if iso is None:
        deg_t = rp_t[1:] - rp_t[:-1]
        w = (deg_t == 0).to(torch.float32)
        mult = getattr(g, "row_mult", None)
        if mult is not None:
            w = w * mult                  # a ghost representative stands for Nmax - n_b identical all-masked columns
        iso = cache[H] = w.unsqueeze(1).expand(R, H).contiguous()
    return iso


class _AttentionAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, a_row, a_col, g, H, slope, by_column, uniform_isolated, drop_mult=None):
        """h [R, H*Fh]; a_row / a_col [H, Fh]: vectors dotted with the ROW node i / COLUMN node j of entry (i,j).
        returns pre-activation out[i] = sum_j alpha_ij h_j (+ uniform term).  drop_mult [nnz, H] (per-target mode only): the
        attention dropout of PyG GATConv as a multiplier per CSR entry and head (0 or 1 / (1 - p)), applied to the normalised
        coefficients before the aggregation."""
        if drop_mult is not None and (by_column or uniform_isolated):
            raise NotImplementedError("the per-entry dropout multiplier belongs to the per-target softmax (GATConv); the column-softmax "
                                      "layers draw their mask inside the fused kernels (gat_fused.py)")
        h = h.contiguous()
        a_row = a_row.contiguous()
        a_col = a_col.contiguous()
        R, C = h.shape
        Fh = C // H
        dev = h.device
        s_row, s_col = node_scores2(h, a_row, a_col, H, Fh)
        nnz = max(g.nnz, 1)
        rp_t, col_t, src_e_t = g.transpose_map() if by_column else (None, None, None)
        eperm = None
        if by_column:
            # groups = columns j = rows of A^T: alpha_G lives on A^T's entries and the aggregation over A's rows reads it
            # through the entry map (no permuted copy)
            alpha_g = _f32(nnz, H, device=dev)
            nat.call("edge_softmax_fwd_f32", rp_t, col_t, R, H, s_col, s_row, 0, float(slope), alpha_g)
            alpha = alpha_g
            eperm = _inverse_entry_map(g, src_e_t)
        else:
            alpha_g = None
            alpha = _f32(nnz, H, device=dev)
            nat.call("edge_softmax_fwd_f32", g.rowptr, g.col, R, H, s_row, s_col, 0, float(slope), alpha)
            if drop_mult is not None:
                ctx.alpha_soft = alpha                         # the softmax backward needs the un-dropped coefficients
                alpha = alpha * drop_mult
        out = _f32(R, C, device=dev)
        iso = None
        if uniform_isolated:
            # columns with no edge: softmax of an all-masked column is uniform 1/N over the N rows of the graph; the
            # per-graph sum u is added to every row in the aggregation's epilogue
            iso = _isolated_columns(g, rp_t, R, H)
            N = g.nmax
            lst = _isolated_list(g, iso)
            if lst is not None:                                # a few listed rows per graph (packed batch): sum those only
                u = _f32(g.B, C, device=dev)
                nat.call("gather_wsum_f32", h, h.stride(0), lst[0], lst[1], lst[2], g.B, C, 1.0 / N, u, u.stride(0))
            else:                                              # padded batch: most rows are edge-less columns, scan them all
                u = segment_wsum(h, iso, H, Fh, g.graph_ptr, g.B, scale=1.0 / N, max_seg=int(g.sizes.max()))
            nat.call("csr_spmm_heads_epi_f32", g.rowptr, g.col, alpha, H, Fh, h, h.stride(0), 0, out, out.stride(0), R,
                     None, None, 0, None, None, 0, u, u.stride(0), None, N, _row_seg(g), 1.0, eperm)
        elif eperm is not None:
            nat.call("csr_spmm_heads_epi_f32", g.rowptr, g.col, alpha, H, Fh, h, h.stride(0), 0, out, out.stride(0), R,
                     None, None, 0, None, None, 0, None, 0, None, 1, None, 1.0, eperm)
        else:
            nat.call("csr_spmm_heads_f32", g.rowptr, g.col, alpha, H, Fh, h, h.stride(0), 0, out, out.stride(0), R)
        ctx.g, ctx.H, ctx.Fh, ctx.slope, ctx.by_column = g, H, Fh, slope, by_column
        ctx.drop_mult = drop_mult
        ctx.save_for_backward(h, a_row, a_col, s_row, s_col, alpha, alpha_g, iso)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, a_row, a_col, s_row, s_col, alpha, alpha_g, iso = ctx.saved_tensors
        g, H, Fh, slope = ctx.g, ctx.H, ctx.Fh, ctx.slope
        dout = dout.contiguous()
        R, C = h.shape
        dev = h.device
        nnz = max(g.nnz, 1)
        rp_t, col_t, src_e_t = g.transpose_map()
        dh = _f32(R, C, device=dev)
        ds_row = _f32(R, H, device=dev)
        ds_col = _f32(R, H, device=dev)
        if ctx.by_column:
            # everything stays in A^T's entry order: d alpha_ij = <dout_i, h_j> computed over A^T's rows (operands swapped),
            # the softmax backward there, and ds_row (a sum over A's rows) read through the entry map — no permuted copies
            dalpha_g = _f32(nnz, H, device=dev)
            nat.call("csr_sddmm_heads_f32", rp_t, col_t, H, Fh, h, h.stride(0), dout, dout.stride(0), 0, dalpha_g, R)
            dt_g = _f32(nnz, H, device=dev)
            nat.call("edge_softmax_bwd_f32", rp_t, col_t, R, H, s_col, s_row, 0, float(slope), alpha_g, dalpha_g, dt_g, ds_col)
            nat.call("csr_row_sum_perm_f32", g.rowptr, dt_g, _inverse_entry_map(g, src_e_t), R, H, ds_row)
            alpha_t = alpha_g
        else:
            dalpha = _f32(nnz, H, device=dev)
            nat.call("csr_sddmm_heads_f32", g.rowptr, g.col, H, Fh, dout, dout.stride(0), h, h.stride(0), 0, dalpha, R)
            dt = _f32(nnz, H, device=dev)
            alpha_s = alpha
            if ctx.drop_mult is not None:                      # out = sum_j (alpha_ij m_ij) h_j: d alpha = m * d(alpha m)
                dalpha = dalpha * ctx.drop_mult
                alpha_s = ctx.alpha_soft
            nat.call("edge_softmax_bwd_f32", g.rowptr, g.col, R, H, s_row, s_col, 0, float(slope), alpha_s, dalpha, dt, ds_row)
            dt_t = _f32(nnz, H, device=dev)
            nat.call("edge_permute_f32", dt, src_e_t, g.nnz, H, 0, dt_t)
            nat.call("csr_row_sum_f32", rp_t, dt_t, R, H, ds_col)
            alpha_t = _f32(nnz, H, device=dev)
            nat.call("edge_permute_f32", alpha, src_e_t, g.nnz, H, 0, alpha_t)
        # dh_j = sum_i alpha_ij dout_i  (transposed aggregation)  + ds_row (x) a_row + ds_col (x) a_col
        # (the three element-wise terms ride in the aggregation's epilogue: one pass over dh instead of four)
        du, N = None, max(g.nmax, 1)
        if iso is not None:
            du = segment_wsum(dout, None, H, Fh, g.graph_ptr, g.B, scale=1.0, max_seg=int(g.sizes.max()))
        nat.call("csr_spmm_heads_epi_f32", rp_t, col_t, alpha_t, H, Fh, dout, dout.stride(0), 0, dh, dh.stride(0), R,
                 ds_row, a_row, a_row.stride(0), ds_col, a_col, a_col.stride(0), du, du.stride(0) if du is not None else 0,
                 iso, N, _row_seg(g), 1.0 / N, None)
        da_row, da_col = segment_wsum2(h, ds_row, ds_col, H, Fh)
        da_row, da_col = da_row.view(H, Fh), da_col.view(H, Fh)
        return dh, da_row, da_col, None, None, None, None, None, None


def attention_aggregate(h, a_row, a_col, g, heads, slope=0.2, by_column=True, uniform_isolated=True, drop_mult=None):
    return _AttentionAggregate.apply(h, a_row, a_col, g, int(heads), float(slope), bool(by_column), bool(uniform_isolated), drop_mult)


class _EluHeads(torch.autograd.Function):
    """ELU per head (concat) or mean over heads (+ELU)  — encoders_GAT.py:47 / :78-83."""

    @staticmethod
    def forward(ctx, x, H, mean_heads, apply_elu):
        x = x.contiguous()
        R, C = x.shape
        Fh = C // H
        y = _f32(R, Fh if mean_heads else C, device=x.device)
        nat.call("elu_heads_fwd_f32", x, R, H, Fh, int(mean_heads), int(apply_elu), y)
        ctx.save_for_backward(x)
        ctx.cfg = (H, Fh, mean_heads, apply_elu)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        H, Fh, mean_heads, apply_elu = ctx.cfg
        dx = torch.empty_like(x)
        nat.call("elu_heads_bwd_f32", x, dy.contiguous(), x.size(0), H, Fh, int(mean_heads), int(apply_elu), dx)
        return dx, None, None, None


def elu_heads(x, heads, mean_heads, apply_elu):
    return _EluHeads.apply(x, int(heads), bool(mean_heads), bool(apply_elu))
