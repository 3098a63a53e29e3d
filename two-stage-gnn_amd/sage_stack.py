"""The GraphSage-style conv stack of GcnEncoderGraph (encoders.py:177-217) as ONE autograd node with a minimal launch
sequence (the b = 32 shape is bound by launch boundaries and per-kernel latency chains).

  forward, layer 0   : gather_rowgemm (aggregate + .W + bias + L2 normalise; z kept for dW)  -> slot_bn_fwd (ReLU + slot BN)
  forward, layer l>0 : sage_layer_fwd (the same product || max-readout partial of the layer's input)  [-> slot_bn_fwd]
  forward, tail      : readout_head_fwd (last layer's readout from its rows + decode of the others + both nn.Linear), or
                       readout_partial + readout_decode_layers when the head is not part of the node
  backward           : [head2_bwd ->] per layer slot_post_bwd (readout scatter + BN + ReLU + normalise backward -> dU), then
                       sage_layer_bwd (dW/db slabs || dX = (A dU) W^T) for hidden layers / linear_wgrad slabs for layer 0,
                       ONE slab reduction for all layers (straight into the trainer's flat gradient bucket when installed)

Ghost rows: they aggregate nothing, so the products skip them and a filler block writes their constant output;
only the ghost slots up to the largest graph can influence anything, so the slot kernels run on those.

Used when the batch qualifies (slot-BN on, sum aggregation without self term, <= 128 graphs, widths <= 128); any other
configuration runs the operator-by-operator path in dense_encoders.py: same results, more launches.  Every fusion has a switch
(environment / module attribute) that selects the launch sequence it replaces.
"""
import torch

from . import _native as nat
from . import message_passing as mp


def eligible(g, convs, bn, x):
    if not bn or g.B > 128 or len(convs) < 2:
        return False
    hid = convs[0].output_dim
    for i, c in enumerate(convs):
        if c.add_self or not c.normalize_embedding or c.dropout > 0.001:
            return False
        if c.output_dim % 4 or c.output_dim > 128:
            return False
        if i < len(convs) - 1 and c.output_dim != hid:
            return False
    return x.dim() == 2 and x.size(1) % 4 == 0 and x.is_cuda and x.stride(0) % 4 == 0


def _aggregate_raw(g, x, transposed=False, rows=None):
    """rows: produce only the first `rows` output rows (ghost rows aggregate nothing and, on the fused path, nobody reads
    their aggregate: the products leave them out)."""
    n = g.total_rows if rows is None else rows
    if g.val is None and mp.ell_ok(x) and g.total_rows <= mp.ELL_MAX_ROWS and (not transposed or g.symmetric):
        return mp.spmm_ell(g, x, rows=n)
    rp, col, val = g.transposed() if transposed else (g.rowptr, g.col, g.val)
    return mp.spmm_raw(rp, col, val, x, n)


import os

FUSED_TAIL = os.environ.get("TSGNN_FUSED_TAIL", "1") != "0"        # last readout + decode + the two Linear layers in one launch
MERGED_FWD = os.environ.get("TSGNN_MERGED_FWD", "1") != "0"        # a layer's product + the readout partial of its input in one launch
MERGED_BWD = os.environ.get("TSGNN_MERGED_BWD", "1") != "0"        # weight-gradient slabs + input-gradient product in one launch
GATHER_MAX_ROWS = int(os.environ.get("TSGNN_GATHER_MAX_ROWS", 65536))   # above: stand-alone row-batched aggregation + lean product
GATHER_FUSED = os.environ.get("TSGNN_GATHER_FUSED", "1") != "0"     # aggregate inside the `.W` product when the neighbour table has no CSR tail
EPILOGUE_READOUT = os.environ.get("TSGNN_EPILOGUE_READOUT", "1") != "0"   # the last layer's max readout in its product's epilogue
LAST_LAYER_ROWS = os.environ.get("TSGNN_LAST_LAYER_ROWS", "1") != "0"     # the last layer's dU from a row-parallel kernel
RO_MAP = os.environ.get("TSGNN_RO_MAP", "1") != "0"                       # readout blocks placed on the XCD that holds their graph's rows
FUSED_BN = os.environ.get("TSGNN_FUSED_BN", "1") != "0"                   # slot batch-norm without launches of its own (statistics in the
                                                                          # producing product's epilogue, normalisation in the consumers)
# layer 0: dU and its weight-gradient slabs in ONE launch, dU kept in LDS (tsgnn_slot_post_wgrad_f32).  Correct (tests run it) and one
# launch less, but measured SLOWER than the two launches it replaces (DD b32: 12.7 us at one slot per workgroup + 10.7 us for the
# reduction of 399 slabs, 15.3 + 5.1 at two slots, against 6.7 + 7.8 + 4.2): a slot is a full latency chain, and the slabs are per
# workgroup.  Off by default.
SLOT_WGRAD = os.environ.get("TSGNN_SLOT_WGRAD", "0") != "0"
HEAD_DU = os.environ.get("TSGNN_HEAD_DU", "1") != "0"                     # ... computed by extra workgroups of the head's backward launch


DU_MAP = os.environ.get("TSGNN_DU_MAP", "1") != "0"             # exact batches: the head backward's dU workgroups listed by the host
SLABS_BESIDE = os.environ.get("TSGNN_SLABS_BESIDE", "1") != "0"
NSLAB_MAX = int(os.environ.get("TSGNN_NSLAB_MAX", "0"))
_ncu = {}


def _cu_count(dev):
    n = _ncu.get(dev)
    if n is None:
        n = _ncu[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    return n


def _slabs_beside_panels(nslab, rps, need, rows, K, N, dev):
    """The merged backward launch hosts two slab blocks per slab AND one block per 32-row panel; a CU keeps two of these blocks
    (registers).  The plan sizes the slabs to one block per CU, which is right while the panels fit the CUs too; with more
    panels than CUs the launch would need a THIRD block on some CUs, which waits for a free slot (DD seed 6, 288 panels:
    13.7 -> 18.2 us).  Fewer, longer slabs keep the launch at two blocks per CU."""
    if not SLABS_BESIDE or nslab <= 0:
        return nslab, rps, need
    if NSLAB_MAX and nslab > NSLAB_MAX:                         # (experiment knob: fewer, longer slabs)
        rps = (-(-rows // NSLAB_MAX) + 7) // 8 * 8
        nslab = -(-rows // rps)
        need = nslab * (K + 1) * N
    ncu = _ncu.get(dev)
    if ncu is None:
        ncu = _ncu[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    npan = int(nat.lib().tsgnn_panel_blocks(int(rows)))         # (32-row panels, or 16-row units for the rows beyond one panel per CU)
    if npan <= ncu or 2 * nslab + npan <= 2 * ncu:
        return nslab, rps, need
    cap = max(32, (2 * ncu - npan) // 2)
    rps = -(-rows // cap)
    rps = (rps + 7) // 8 * 8
    nslab = -(-rows // rps)
    return nslab, rps, nslab * (K + 1) * N


def _gather_ok(g, x):
    if not GATHER_FUSED or g.val is not None or not mp.ell_ok(x) or g.total_rows > GATHER_MAX_ROWS:
        return False
    return True


def _flush_readout(g, B, sn, sg, pending):
    if pending is not None:
        y, pk = pending
        nat.call("readout_partial_f32", g.graph_ptr, B, sn, g.n_rows, sg, y, y.stride(0), y.size(1), pk)


# Per-graph statistics (B = 1 semantics: the 2stg triplet step runs anchor / positive / negative as one batch in which every graph
# keeps the batch-norm statistics it would have alone, tripletnet.py:36-38): the slot batch-norm degenerates to a per-row layer norm,
# so the two slot launches of a hidden layer are replaced by their row-local counterparts (tsgnn_row_ln_fwd_f32 forward,
# tsgnn_row_post_bwd_f32 backward) and everything else of the stack — aggregation inside the products, the readout partials riding
# in the next layer's launch, merged weight-gradient / input-gradient launches, one reduction for all layers — stays.  Selected by
# the caller around the node's forward (`with per_graph_stats(True):`); the statistics without launches of their own (FUSED_BN)
# are per slot across graphs and are not used in this mode.
_PER_GRAPH = [False]


class per_graph_stats:
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _PER_GRAPH[0]
        _PER_GRAPH[0] = self.on

    def __exit__(self, *exc):
        _PER_GRAPH[0] = self.prev
        return False


class _SageStack(torch.autograd.Function):
    """forward(x0, g, has_bias, n_head, nodes, *conv params[, w1, b1, w2, b2]).  n_head = 0: returns the concatenated readout
    [B, P].  n_head = 4: the two chained nn.Linear after the readout (encoders.py:207-217) are part of the node: the last
    layer's readout, the decode of the earlier layers and the head run as ONE launch and (vec, y) are returned.
    nodes = 1 / 2: no readouts; returns the per-layer NODE features concatenated on the feature axis [R, P] (gcn_forward,
    encoders.py:140-167), 2 = ghost rows zeroed (the embedding mask); layers write straight into the concatenated buffer."""

    @staticmethod
    def forward(ctx, x0, g, has_bias, n_head, nodes, *params):
        head = params[len(params) - n_head:] if n_head else None
        params = params[:len(params) - n_head] if n_head else params
        L = len(params) // 2
        Ws = [params[2 * l].contiguous() for l in range(L)]
        bs = [params[2 * l + 1] if has_bias else None for l in range(L)]
        dev = x0.device
        R, B = g.total_rows, g.B
        Fh, Fl = Ws[0].size(1), Ws[-1].size(1)
        total = B * ((L - 1) * Fh + Fl)
        # cleared by the first slot_bn_fwd launch; Fl spare words behind the last layer's segment take the readout of the dummy
        # graph that the padding rows of a capacity-padded batch (ingest.py) belong to: never cleared, never read
        packed = torch.empty(total + Fl, dtype=torch.int64, device=dev) if not nodes else None
        cat = torch.empty(R, (L - 1) * Fh + Fl, dtype=torch.float32, device=dev) if nodes else None
        x = mp._check(x0, R)
        # Ghost slots actually needed.  Every graph's padded rows at slots >= the largest graph are bitwise identical in
        # every layer (same bias row, same statistics), the max readout breaks ties towards the smallest row, and nothing
        # aggregates from a ghost row: only slots [0, max_size] can influence an output or a gradient.  The slot kernels,
        # the filler and the bias gradient therefore run on  gs = min(nmax, max_size + 1)  ghost rows (half of nmax on DD).
        gs = g.n_ghost
        if g.n_ghost > 0 and x.stride(0) % 4 == 0 and all(
                Ws[l].size(1) % 4 == 0 and Ws[l].data_ptr() % 16 == 0 and (bs[l] is None or bs[l].data_ptr() % 16 == 0)
                for l in range(L)):
            # (capacity-padded batches keep one shape for every batch: a fixed bound instead of this batch's largest graph)
            fixed = getattr(g, "ghost_slots_fixed", None)
            gs = min(g.nmax, (int(fixed) if fixed is not None else int(g.sizes.max()) + 1))
        if nodes == 1:
            gs = g.n_ghost                                   # unmasked node output: every ghost row is part of the result
        sn, sg = (gs, gs) if g.n_ghost else (g.nmax, 0)      # (slots, ghost rows) handed to the slot kernels
        saved = []
        off = 0
        pending_ro = None
        keep = []
        last_ro_done = False
        bnf = None
        per_graph = ctx.per_graph = bool(_PER_GRAPH[0])
        if per_graph and (nodes or head is not None):
            raise NotImplementedError("per-graph statistics: the readout form of the stack only")
        if (FUSED_BN and not per_graph and head is not None and not nodes and L >= 2 and g.n_ghost == g.nmax and sn == sg and sn <= 1024
                and Fh == 128 and Fl == 128 and Ws[0].size(0) <= 128 and x.size(1) % 4 == 0 and MERGED_FWD and EPILOGUE_READOUT
                and _gather_ok(g, x) and all(Ws[l].size(0) == 128 and Ws[l].stride(0) % 4 == 0 for l in range(1, L))
                and all(Ws[l].data_ptr() % 16 == 0 and (bs[l] is None or bs[l].data_ptr() % 16 == 0) for l in range(L))
                and mp.rowgemm_ok(x, x.stride(0), Ws[0], Ws[0].stride(0), Ws[0].size(0), Fh, False)
                and head[0].size(0) <= 128 and (L - 1) * Fh + Fl <= 2048 and g.row_graph is not None):
            ell_s = g.ell_slots()
            if ell_s is not None:
                bnf = g.bn_workspace(B, L, Fh, Fl, sn)
                if bnf["dirty"]:
                    bnf["sums"].zero_(); bnf["ghost"].zero_(); bnf["packed"].zero_()
                bnf["dirty"] = True
                packed = bnf["packed"]
        if bnf is not None:
            # ---- slot batch-norm without launches of its own (L launches for the conv stack instead of 2L - 1)
            ell, ell_w, tail = g.ell()
            tp, tc = tail if tail is not None else (None, None)
            ell_s, tc_s = ell_s
            sums, ghost = bnf["sums"], bnf["ghost"]
            ro_map, ro_ch = g.readout_map(sn, gs) if RO_MAP else (None, 0)
            for l in range(L):
                K, N = Ws[l].size(0), Ws[l].size(1)
                v = torch.empty(R, N, dtype=torch.float32, device=dev)
                rinv = torch.empty(R, dtype=torch.float32, device=dev)
                z = torch.empty(R, x.size(1) if l == 0 else Fh, dtype=torch.float32, device=dev)
                s_out = sums[l * 2 * sn:(l + 1) * 2 * sn] if l < L - 1 else None
                g_out = ghost[2 * l:2 * l + 2] if l < L - 1 else None
                if l == 0:
                    nat.call("gather_rowgemm_st_f32", ell, ell_w, tp, tc, x, x.stride(0), Ws[0], Ws[0].stride(0), bs[0], v, v.stride(0), rinv, z,
                             z.stride(0), g.n_rows, K, N, gs, g.row_slot, s_out, g_out)
                    mean = rstd = None
                else:
                    pm, pr_ = saved[l - 1][3], saved[l - 1][4]
                    last = l == L - 1
                    nat.call("sage_layer_fwd_bn_f32", ell_s, ell_w, tp, tc_s, saved[l - 1][1], saved[l - 1][1].stride(0), Ws[l], Ws[l].stride(0), bs[l],
                             v, v.stride(0), rinv, z, z.stride(0), g.n_rows, K, gs, g.graph_ptr, g.slot_count, B, sn, sg,
                             packed[(l - 1) * B * Fh:(l - 1) * B * Fh + B * Fh],
                             packed[l * B * Fh:l * B * Fh + (B + 1) * N] if last else None, g.row_graph,
                             sums[(l - 1) * 2 * sn:l * 2 * sn], ghost[2 * (l - 1):2 * l], pm, pr_,
                             None if last else g.row_slot, s_out, g_out, ro_map, ro_ch)
                if l < L - 1:
                    mean = torch.empty(g.nmax, dtype=torch.float32, device=dev)     # written by the NEXT launch's readout blocks
                    rstd = torch.empty(g.nmax, dtype=torch.float32, device=dev)
                else:
                    mean = rstd = None
                saved.append((z, v, rinv, mean, rstd, True))
            last_ro_done = True
        for l in (range(L) if bnf is None else ()):
            K, N = Ws[l].size(0), Ws[l].size(1)
            if nodes and l == L - 1:
                v = cat[:, (L - 1) * Fh:]                    # the last layer's output IS its block of the concatenation
            else:
                v = torch.empty(R, N, dtype=torch.float32, device=dev)
            rinv = torch.empty(R, dtype=torch.float32, device=dev)
            # Ghost rows aggregate nothing (z = 0): their output is the normalised bias, written by a filler block, and
            # their z is neither produced nor read (255 row panels + 1 filler = one block per CU on the DD batch).
            lean = (g.n_ghost > 0 and x.stride(0) % 4 == 0 and N % 4 == 0 and Ws[l].data_ptr() % 16 == 0
                    and (bs[l] is None or bs[l].data_ptr() % 16 == 0))
            fused = lean and N <= 128 and _gather_ok(g, x) and mp.rowgemm_ok(x, x.stride(0), Ws[l], Ws[l].stride(0), K, N, False)
            if (fused and MERGED_FWD and pending_ro is not None and K == 128 and N == 128 and x.size(1) == 128
                    and Ws[l].stride(0) % 4 == 0):
                # this layer's product and the max-readout partial of its input (the previous layer's output) in one launch
                ell, ell_w, tail = g.ell()
                tp, tc = tail if tail is not None else (None, None)
                z = torch.empty(R, x.size(1), dtype=torch.float32, device=dev)
                if EPILOGUE_READOUT and l == L - 1 and not nodes and l > 0 and (not g.n_ghost or gs > 0):
                    # last layer: no slot batch-norm follows, so its own max readout is folded into the product's epilogue
                    # (packed was cleared by layer 0's slot_bn_fwd launch): no pass over v for it
                    nat.call("sage_layer_fwd_ro_f32", ell, ell_w, tp, tc, x, x.stride(0), Ws[l], Ws[l].stride(0), bs[l], v, v.stride(0),
                             rinv, z, z.stride(0), g.n_rows, K, gs, g.graph_ptr, B, sn, sg, pending_ro[1],
                             packed[off:off + (B + 1) * N], g.row_graph)
                    last_ro_done = True
                else:
                    nat.call("sage_layer_fwd_f32", ell, ell_w, tp, tc, x, x.stride(0), Ws[l], Ws[l].stride(0), bs[l], v, v.stride(0), rinv,
                             z, z.stride(0), g.n_rows, K, gs, g.graph_ptr, B, sn, sg, pending_ro[1])
                pending_ro = None
            elif fused:
                _flush_readout(g, B, sn, sg, pending_ro)
                pending_ro = None
                # aggregation fused into the product: the neighbour rows are summed while the A panel is staged
                ell, ell_w, tail = g.ell()
                tp, tc = tail if tail is not None else (None, None)
                z = torch.empty(R, x.size(1), dtype=torch.float32, device=dev)
                nat.call("gather_rowgemm_f32", ell, ell_w, tp, tc, x, x.stride(0), Ws[l], Ws[l].stride(0), 0, bs[l], v, v.stride(0), rinv,
                         z, z.stride(0), g.n_rows, K, N, 1, gs)
            else:
                _flush_readout(g, B, sn, sg, pending_ro)
                pending_ro = None
                z = _aggregate_raw(g, x, rows=g.n_rows if lean else None)
                if lean and mp.rowgemm_ok(z, z.stride(0), Ws[l], Ws[l].stride(0), K, N, False):
                    nat.call("rowgemm_f32", z, z.stride(0), Ws[l], Ws[l].stride(0), 0, bs[l], v, v.stride(0), rinv, g.n_rows, K, N, 1,
                             gs)
                elif mp.rowgemm_ok(z, z.stride(0), Ws[l], Ws[l].stride(0), K, N, False):
                    if lean:
                        z[g.n_rows:].zero_()
                        lean = False
                    nat.call("rowgemm_f32", z, z.stride(0), Ws[l], Ws[l].stride(0), 0, bs[l], v, v.stride(0), rinv, R, K, N, 1, 0)
                else:
                    if lean:
                        z[g.n_rows:].zero_()
                        lean = False
                    nat.call("linear_l2norm_f32", z, z.stride(0), Ws[l], Ws[l].stride(0), bs[l], v, v.stride(0), rinv, R, K, N, 1)
            pk = packed[off:off + B * N] if not nodes else None
            if l < L - 1 and per_graph:
                mean = torch.empty(R, dtype=torch.float32, device=dev)      # per ROW
                rstd = torch.empty(R, dtype=torch.float32, device=dev)
                y = torch.empty_like(v)
                if l == 0:
                    packed[:total].zero_()                    # (what the first slot_bn_fwd launch does on its way)
                nat.call("row_ln_fwd_f32", v, v.stride(0), g.n_rows + sg, N, 1, mean, rstd, y, y.stride(0))
                pending_ro = (y, pk)
                keep.append(y)
                x = y
            elif l < L - 1:
                mean = torch.empty(g.nmax, dtype=torch.float32, device=dev)
                rstd = torch.empty(g.nmax, dtype=torch.float32, device=dev)
                y = torch.empty_like(v) if not nodes else cat[:, l * Fh:(l + 1) * Fh]
                nat.call("slot_bn_fwd_f32", g.graph_ptr, g.slot_count, B, sn, g.n_rows, sg, v, v.stride(0), N, 1,
                         mean, rstd, y, y.stride(0), packed if (l == 0 and not nodes) else None, total)
                if not nodes:
                    pending_ro = (y, pk)                    # rides along with the next layer's product (or is flushed before it)
                keep.append(y)
                x = y
            else:
                mean = rstd = None
                if head is None and not nodes and not last_ro_done:
                    nat.call("readout_partial_f32", g.graph_ptr, B, sn, g.n_rows, sg, v, v.stride(0), N, pk)
            saved.append((z, v, rinv, mean, rstd, lean))
            off += B * N
        ctx.nodes = nodes
        if nodes:
            if nodes == 2 and g.n_ghost:
                nat.defer_zero(cat[g.n_rows:])               # embedding mask: ghost rows of the node output are zero
            ctx.g, ctx.L, ctx.has_bias, ctx.dims = g, L, has_bias, (Fh, Fl)
            ctx.slots = (sn, sg)
            ctx.Ws, ctx.saved, ctx.arg = Ws, saved, None
            ctx.params = params
            ctx.head = None
            return cat
        out = torch.empty(B, (L - 1) * Fh + Fl, dtype=torch.float32, device=dev)
        arg = torch.empty(total, dtype=torch.int32, device=dev)
        ctx.g, ctx.L, ctx.has_bias, ctx.dims = g, L, has_bias, (Fh, Fl)
        ctx.slots = (sn, sg)
        ctx.Ws, ctx.saved, ctx.arg = Ws, saved, arg
        ctx.params = params
        ctx.x0_ld = x.size(1) if L == 0 else x0.size(1)
        ctx.head = None
        if head is None:
            nat.call("readout_decode_layers_f32", packed, B, L, Fh, Fl, out, out.stride(0), arg)
            return out
        # last layer's readout (straight from its rows) + decode of the earlier layers + both Linear layers: one launch
        w1, b1, w2, b2 = head
        w1c, w2c = w1.contiguous(), w2.contiguous()
        E, C = w1c.size(0), w2c.size(0)
        vec = torch.empty(B, E, dtype=torch.float32, device=dev)
        y = torch.empty(B, C, dtype=torch.float32, device=dev)
        v_last = saved[-1][1]
        if bnf is not None:
            # decode + both Linear layers + the step's housekeeping (packed and the integer sums zeroed for the next step)
            nat.call("packed_head_fwd_z_f32", packed, B, L, Fh, Fl, out, out.stride(0), arg, w1c, b1, w2c, b2, E, C, vec, y,
                     bnf["sums"], (L - 1) * 2 * sn)
            bnf["dirty"] = False
        elif last_ro_done and E <= 128 and out.size(1) <= 2048:
            # every layer's maxima are in `packed`: decode + both Linear layers, one memory round trip per block
            nat.call("packed_head_fwd_f32", packed, B, L, Fh, Fl, out, out.stride(0), arg, w1c, b1, w2c, b2, E, C, vec, y)
        else:
            nat.call("readout_head_fwd_f32", packed, B, L, Fh, Fl, v_last, v_last.stride(0), g.graph_ptr, g.n_rows, sn, sg, out,
                     out.stride(0), arg, w1c, b1, w2c, b2, E, C, vec, y)
        ctx.head = (out, vec, w1c, w2c, head)
        ctx.set_materialize_grads(False)
        return vec, y

    @staticmethod
    def backward(ctx, *gouts):
        g, L = ctx.g, ctx.L
        Fh, Fl = ctx.dims
        R, B = g.total_rows, g.B
        head_grads = ()
        du_last = None                                # the last layer's dU when the head's backward launch produced it
        if ctx.head is None:
            dout = gouts[0].contiguous()
            dev = dout.device
        else:
            out, vec, w1c, w2c, (pw1, pb1, pw2, pb2) = ctx.head
            dvec, dy = gouts
            dev = out.device
            P, E, C = out.size(1), w1c.size(0), w2c.size(0)
            if dy is None and dvec is None:
                return (None,) * (5 + 2 * L + 4)
            ce = mp.take_deferred_ce(dy) if dy is not None else None      # deferred cross-entropy: this backward rebuilds dy
            if ce is None:
                dy = dy.contiguous() if dy is not None else torch.zeros(B, C, device=dev)
            dvec = dvec.contiguous() if dvec is not None else None
            dout = torch.empty(B, P, dtype=torch.float32, device=dev)
            dw1, s1 = mp._sink_or_new(pw1, (E, P), dev)
            dw2, s2 = mp._sink_or_new(pw2, (C, E), dev)
            db1, s3 = mp._sink_or_new(pb1, (E,), dev) if pb1 is not None else (None, False)
            db2, s4 = mp._sink_or_new(pb2, (C,), dev) if pb2 is not None else (None, False)
            parts = mp.head_norm_slots((s1, s2, s3, s4), (pb1 is not None, pb2 is not None), (pw1, pb1, pw2, pb2), E)
            sn_, sg_ = ctx.slots
            v_l, rinv_l, lean_l = ctx.saved[L - 1][1], ctx.saved[L - 1][2], ctx.saved[L - 1][5]
            if (HEAD_DU and LAST_LAYER_ROWS and not ctx.nodes and L > 1 and lean_l and Fl % 4 == 0 and Fl <= 128 and g.n_ghost == g.nmax
                    and sn_ == sg_ and g.n_ghost >= B and ctx.needs_input_grad[5 + 2 * (L - 1)] and v_l.stride(0) % 4 == 0):
                # the last layer's dU (a row-wise function of the readout gradient: it has no batch-norm) rides in this launch
                du_l = torch.empty(R, Fl, dtype=torch.float32, device=dev)
                argl = ctx.arg[(L - 1) * B * Fh:(L - 1) * B * Fh + B * Fl]
                # (the non-empty (graph, chunk) pairs listed by the host for an exact batch: no workgroup that only returns)
                dmap, ndmap, dchunk = g.du_map(B + (E + 3) // 4 + 1) if DU_MAP else (None, 0, 64)
                if nat.try_call("head2_bwd_du_map_f32", out, out.stride(0), vec, ce[0] if ce is not None else None,
                                ce[1] if ce is not None else None, ce[2] if ce is not None else None, None if ce is not None else dy, dvec,
                                w1c, w2c, B, P, E, C, dout, dout.stride(0), dw1, db1, dw2, db2, parts, g.graph_ptr, g.n_rows, sg_,
                                sn_, v_l, v_l.stride(0), rinv_l, argl, (L - 1) * Fh, Fl, du_l, du_l.stride(0), dmap, ndmap, dchunk):
                    du_last = du_l
            if du_last is not None:
                pass
            elif ce is not None:
                nat.call("head2_bwd_ce_f32", out, out.stride(0), vec, ce[0], ce[1], ce[2], dvec, w1c, w2c, B, P, E, C, dout,
                         dout.stride(0), dw1, db1, dw2, db2, parts)
            else:
                nat.call("head2_bwd_f32", out, out.stride(0), vec, dy, dvec, w1c, w2c, B, P, E, C, dout, dout.stride(0), dw1, db1, dw2,
                         db2, parts)
            head_grads = (None if s1 else dw1, None if s3 else db1, None if s2 else dw2, None if s4 else db2)
            if mp.GRAD_SINK is not None and s1 and s2 and (s3 or pb1 is None) and (s4 or pb2 is None):
                mp.GRAD_SINK.ready((pw1, pb1, pw2, pb2))       # final already: their all-reduce may overlap the conv backward
        sn, sg = ctx.slots
        grads = [None] * (2 * L)
        dxs = None
        dx0 = None
        keep = []
        pending = []
        pend_sunk, pend_layers = [], []
        for l in range(L - 1, -1, -1):
            z, v, rinv, mean, rstd, lean = ctx.saved[l]
            W = ctx.Ws[l]
            K, N = W.size(0), W.size(1)
            last = l == L - 1
            du = du_last if (last and du_last is not None) else torch.empty(R, N, dtype=torch.float32, device=dev)
            bo = sg                                   # rows behind the real ones that feed the bias gradient only
            if ctx.nodes:
                dsl = argl = None
                dnode = dout[:, l * Fh:l * Fh + N]           # gradient of this layer's block of the node output
            else:
                dsl = dout[:, l * Fh:l * Fh + N]
                argl = ctx.arg[l * B * Fh:l * B * Fh + B * N]
                dnode = None
            if last and du_last is not None:
                bo = B                                # B ghost CONTRIBUTION rows stand for the sg ghost rows (tsgnn_head2_bwd_du_f32)
            elif (LAST_LAYER_ROWS and last and not ctx.nodes and dxs is None and N % 4 == 0 and N <= 128 and g.n_ghost == g.nmax
                    and sn == sg and dout.stride(0) % 4 == 0 and dsl.data_ptr() % 16 == 0 and g.row_graph is not None):
                # the last layer has no batch-norm: its dU is a row-wise function of the readout gradient (no slot structure)
                nat.call("readout_l2_bwd_f32", g.graph_ptr, g.row_graph, B, g.n_rows, sg, v, v.stride(0), dsl, dout.stride(0), argl, N, rinv,
                         du, du.stride(0))
            elif (SLOT_WGRAD and l == 0 and L > 1 and not ctx.nodes and not ctx.needs_input_grad[0] and lean and B <= 32 and N == 128
                  and K <= 128 and sn == sg and g.n_ghost == g.nmax and ctx.needs_input_grad[5] and z.stride(0) % 4 == 0
                  and z.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0 and (not ctx.has_bias or ctx.needs_input_grad[6])):
                # layer 0's dU has ONE consumer, its own weight / bias gradient: both in one launch, the rows of dU stay in LDS
                per = int(os.environ.get("TSGNN_SLOT_WGRAD_PER", "2"))      # slots per workgroup (each slot is a full latency chain)
                nblk = max(1, -(-sn // per))
                ws0 = torch.empty(nblk * (K + 1) * N, dtype=torch.float32, device=dev)
                nat.call("slot_post_wgrad_f32", g.graph_ptr, g.slot_count, B, sn, g.n_rows, sg, v, v.stride(0), dxs,
                         dxs.stride(0) if dxs is not None else 0, dsl, dout.stride(0) if dsl is not None else 0, argl, N, 1, 1, mean, rstd,
                         rinv, z, z.stride(0), K, ws0, nblk)
                dw, sw = mp._sink_or_new(ctx.params[0], (K, N), dev)
                db, sb = mp._sink_or_new(ctx.params[1], (N,), dev) if ctx.has_bias else (None, False)
                pending.append((ws0, nblk, K, N, dw, db))
                pend_sunk.append(sw and (sb or not ctx.has_bias)); pend_layers.append(0)
                grads[0], grads[1] = (None if sw else dw), (None if sb else db)
                continue
            elif ctx.per_graph:
                # per-graph statistics: readout winners + row layer norm + ReLU + normalise backward, row by row
                nat.call("row_post_bwd_f32", g.row_graph, B, g.n_rows, g.n_rows + sg, v, v.stride(0), dxs,
                         dxs.stride(0) if dxs is not None else 0, dsl, dout.stride(0), argl, N, 0 if last else 1, 0 if last else 1, mean,
                         rstd, rinv, du, du.stride(0))
            else:
                nat.call("slot_post_bwd_f32", g.graph_ptr, g.slot_count, B, sn, g.n_rows, sg, v, v.stride(0), dxs,
                         dxs.stride(0) if dxs is not None else 0, dnode, dnode.stride(0) if dnode is not None else 0, dsl,
                         dout.stride(0) if dsl is not None else 0, argl, N, 0 if last else 1, 0 if last else 1, mean, rstd, rinv, du,
                         du.stride(0))
            want_w = ctx.needs_input_grad[5 + 2 * l]
            want_b = ctx.has_bias and ctx.needs_input_grad[6 + 2 * l]
            merged = False
            if (MERGED_BWD and want_w and lean and l > 0 and K == 128 and N == 128 and g.symmetric and z.size(1) == K
                    and _gather_ok(g, du) and z.data_ptr() % 16 == 0 and du.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0
                    and W.stride(0) % 4 == 0):
                nslab, rps, need = mp.wgrad_plan(g.n_rows, K, N, z.stride(0), du.stride(0))
                nslab, rps, need = _slabs_beside_panels(nslab, rps, need, g.n_rows, K, N, dev)
                if 0 < nslab < 512:
                    # weight-gradient slabs and dX = (A dU) W^T side by side in one launch (both only need dU)
                    ell, ell_w, tail = g.ell()
                    tp, tc = tail if tail is not None else (None, None)
                    ws = torch.empty(need, dtype=torch.float32, device=dev)
                    dxs = torch.empty(R, K, dtype=torch.float32, device=dev)
                    nat.call("sage_layer_bwd_f32", ell, ell_w, tp, tc, du, du.stride(0), W, W.stride(0), dxs, dxs.stride(0), z, z.stride(0),
                             g.n_rows, nslab, rps, bo, ws)
                    dw, sw = mp._sink_or_new(ctx.params[2 * l], (K, N), dev)
                    db, sb = mp._sink_or_new(ctx.params[2 * l + 1], (N,), dev) if want_b else (None, False)
                    pending.append((ws, nslab, K, N, dw, db))
                    pend_sunk.append(sw and (sb or not want_b)); pend_layers.append(l)
                    grads[2 * l], grads[2 * l + 1] = (None if sw else dw), (None if sb else db)
                    keep.append(du)
                    merged = True
            if merged:
                continue
            if want_w:
                if not lean and sg < g.n_ghost:
                    du[g.n_rows + sg:].zero_()          # rows no slot kernel wrote
                sl = mp.linear_wgrad_slabs(z, K, du[:g.n_rows + bo] if lean else du, bias_only_rows=bo if lean else 0)
                if sl is not None:                      # slabs now, ONE reduction for all layers at the end
                    dw, sw = mp._sink_or_new(ctx.params[2 * l], (K, N), dev)     # straight into the flat bucket if one is installed
                    db, sb = mp._sink_or_new(ctx.params[2 * l + 1], (N,), dev) if want_b else (None, False)
                    pending.append((sl[0], sl[1], K, N, dw, db))
                    pend_sunk.append(sw and (sb or not want_b)); pend_layers.append(l)
                    dw, db = (None if sw else dw), (None if sb else db)
                else:
                    if lean:
                        z[g.n_rows:].zero_()
                        du[g.n_rows + sg:].zero_()
                    dw, db = mp.linear_wgrad(z, K, du, want_b)
                grads[2 * l], grads[2 * l + 1] = dw, db
            elif want_b:
                if sg < g.n_ghost:
                    du[g.n_rows + sg:].zero_()
                grads[2 * l + 1] = mp.colsum(du)
            keep.append(du)
            need_dx = l > 0 or ctx.needs_input_grad[0]
            if need_dx:
                ldz = z.size(1)
                if (l > 0 and g.n_ghost > 0 and g.symmetric and ldz == K and K <= 128 and _gather_ok(g, du)
                        and mp.rowgemm_ok(du, du.stride(0), W, W.stride(0), N, K, True)):
                    # dX = A^T (dU W^T) = (A dU) W^T for a symmetric A: the same fused gather + product; only real rows
                    ell, ell_w, tail = g.ell()
                    tp, tc = tail if tail is not None else (None, None)
                    dxs = torch.empty(R, ldz, dtype=torch.float32, device=dev)
                    nat.call("gather_rowgemm_f32", ell, ell_w, tp, tc, du, du.stride(0), W, W.stride(0), 1, None, dxs, dxs.stride(0), None,
                             None, 0, g.n_rows, N, K, 0, 0)
                else:
                    dz = torch.zeros(R, ldz, dtype=torch.float32, device=dev) if ldz > K else torch.empty(R, ldz, dtype=torch.float32, device=dev)
                    if mp.rowgemm_ok(du, du.stride(0), W, W.stride(0), N, K, True):
                        # ghost rows have no edges: their dz is never gathered, so only the real rows go through the product
                        nat.call("rowgemm_f32", du, du.stride(0), W, W.stride(0), 1, None, dz, dz.stride(0), None, g.n_rows, N, K, 0, 0)
                    else:
                        mp.gemm(du, du.stride(0), 1, W, 1, W.stride(0), dz, dz.stride(0), 1, R, K, N)
                    # ... and nothing reads the ghost rows of the aggregated gradient (slot_post_bwd takes 0 for them)
                    dxs = _aggregate_raw(g, dz, transposed=True, rows=g.n_rows if (l > 0 and g.n_ghost) else None)
                if l == 0:
                    dx0 = dxs
        if pending:
            sink = mp.GRAD_SINK
            all_sunk = sink is not None and len(pending) <= 4 and all(pend_sunk)
            mp.wgrad_reduce_multi(pending, norm_sink=sink if all_sunk else None)
            if all_sunk and sink.stepped:
                for l_ in pend_layers:
                    sink.normed.add(ctx.params[2 * l_].data_ptr())
                    if ctx.has_bias:
                        sink.normed.add(ctx.params[2 * l_ + 1].data_ptr())
        del keep
        return (dx0, None, None, None, None) + tuple(grads) + head_grads


def sage_stack_readouts(x, g, convs):
    """concatenated max readouts [B, hidden*(L-1)+embedding] of the conv stack (encoders.py:177-203)."""
    has_bias = convs[0].bias is not None
    params = []
    for c in convs:
        params.append(c.weight)
        params.append(c.bias if has_bias else c.weight.new_zeros(1))
    return _SageStack.apply(x, g, has_bias, 0, 0, *params)


def sage_stack_nodes(x, g, convs, mask_ghost):
    """per-layer node features concatenated on the feature axis [R, sum F] (gcn_forward, encoders.py:140-167)."""
    has_bias = convs[0].bias is not None
    params = []
    for c in convs:
        params.append(c.weight)
        params.append(c.bias if has_bias else c.weight.new_zeros(1))
    return _SageStack.apply(x, g, has_bias, 0, 2 if (mask_ghost and g.n_ghost) else 1, *params)


# ----------------------------------------------------------------------------- two stacks on one graph, launches shared
PAIR_LAUNCHES = os.environ.get("TSGNN_STACK_PAIRS", "1") != "0"
ZERO_RIDER = os.environ.get("TSGNN_ZERO_RIDER", "1") != "0"        # the embedding mask's clearing inside the last paired product launch


def _multi(tn, gs, zero=None):
    """one launch for the recorded argument tuples of <= 2 tsgnn_linear_wgrad_f32 (slab form) and <= 2 tsgnn_gather_rowgemm_f32
    calls; False when the entry point does not take the combination (csrc/multi.hip).  zero: two contiguous tensors that the
    products' filler blocks clear after their own rows (the deferred `_zero` records that follow the products)."""
    import numpy as np
    words = [len(tn), len(gs)]
    for a in tn:
        words += [nat._arg(v) or 0 for v in a[:11]]
    for a in gs:
        words += [(nat._arg(v) or 0) if not isinstance(v, bool) else int(v) for v in a[:20]]
    d = np.asarray(words, dtype=np.int64)
    if zero is not None:
        za, zb = zero
        if not (za.is_contiguous() and zb.is_contiguous()):
            return False
        return nat.try_call("sage_multi_zero_f32", d.ctypes.data, za, za.numel(), zb, zb.numel())
    return nat.try_call("sage_multi_f32", d.ctypes.data)


def _zero_after(q, i):
    return q[i + 1][1][0] if i + 1 < len(q) and q[i + 1][0] == "_zero" else None


def _slab_form(rec):
    return rec[0] == "linear_wgrad_f32" and len(rec[1]) >= 13 and rec[1][11] is None and rec[1][12] is None


def run_paired(qa, qb):
    """issue two launch records of INDEPENDENT computations (nat.deferred) in lockstep; where both are at the same kind of
    step, the two problems share a launch: gather products pairwise, weight-gradient slabs pairwise and together with the
    gather products that follow them (all four only read dU).  Any interleaving that keeps each record's order is valid."""
    i = j = 0
    while i < len(qa) and j < len(qb):
        a, b = qa[i], qb[j]
        if a[0] == b[0] == "gather_rowgemm_f32":
            za, zb = _zero_after(qa, i), _zero_after(qb, j)
            if ZERO_RIDER and za is not None and zb is not None and _multi([], [a[1], b[1]], zero=(za, zb)):
                i += 2; j += 2                      # the ghost rows' clearing rode in the products' filler blocks
                continue
            if _multi([], [a[1], b[1]]):
                i += 1; j += 1
                continue
        if _slab_form(a) and _slab_form(b):
            na = qa[i + 1] if i + 1 < len(qa) else (None,)
            nb = qb[j + 1] if j + 1 < len(qb) else (None,)
            if na[0] == nb[0] == "gather_rowgemm_f32" and _multi([a[1], b[1]], [na[1], nb[1]]):
                i += 2; j += 2
                continue
            if _multi([a[1], b[1]], []):
                i += 1; j += 1
                continue
        if a[0] == b[0] == "_zero":
            torch._foreach_zero_([a[1][0], b[1][0]])
            i += 1; j += 1
            continue
        if a[0] == b[0] and a[0] in _PAIRED and _PAIRED[a[0]](a[1], b[1]):
            i += 1; j += 1
            continue
        nat.run([a]); nat.run([b])
        i += 1; j += 1
    nat.run(qa[i:]); nat.run(qb[j:])


def _same(x, y):
    if torch.is_tensor(x) or torch.is_tensor(y):
        return torch.is_tensor(x) and torch.is_tensor(y) and x.data_ptr() == y.data_ptr()
    return x == y


def _pair_slot_bn(a, b):
    # (graph_ptr, slot_count, B, sn, n_rows, sg, v, ldv, N, relu, mean, rstd, y, ldy, zero_ptr, total)
    if a[14] is not None or b[14] is not None or not all(_same(a[k], b[k]) for k in (0, 1, 2, 3, 4, 5, 7, 8, 9, 13)):
        return False
    return nat.try_call("slot_bn_fwd_pair_f32", a[0], a[1], a[2], a[3], a[4], a[5], a[6], b[6], a[7], a[8], a[9], a[10], b[10], a[11], b[11],
                        a[12], b[12], a[13])


def _pair_slot_post_bwd(a, b):
    # (graph_ptr, slot_count, B, sn, n_rows, sg, v, ldv, dxs, lddxs, dxs2, lddxs2, dout, ldo, arg, N, relu, bn, mean, rstd, rinv, du, lddu)
    if any(t[12] is not None or t[14] is not None for t in (a, b)):
        return False
    if (a[8] is None) != (b[8] is None) or (a[10] is None) != (b[10] is None):
        return False
    if not all(_same(a[k], b[k]) for k in (0, 1, 2, 3, 4, 5, 7, 9, 11, 15, 16, 17, 22)):
        return False
    return nat.try_call("slot_post_bwd_pair_f32", a[0], a[1], a[2], a[3], a[4], a[5], a[6], b[6], a[7], a[8], b[8], a[9], a[10], b[10], a[11],
                        a[15], a[16], a[17], a[18], b[18], a[19], b[19], a[20], b[20], a[21], b[21], a[22])


def _pair_reduce(a, b):
    # 4 x (ws, nslab, K, N, dw, db), normparts, step
    import numpy as np
    if a[24] is not None or a[25] is not None or b[24] is not None or b[25] is not None:
        return False
    sets = [t[6 * k:6 * k + 6] for t in (a, b) for k in range(4) if t[6 * k] is not None]
    if not sets or len(sets) > 8:
        return False
    words = [len(sets)]
    for st in sets:
        words += [nat._arg(st[0]), int(st[1]), int(st[2]), int(st[3]), nat._arg(st[4]), nat._arg(st[5]) or 0]
    d = np.asarray(words, dtype=np.int64)
    return nat.try_call("wgrad_reduce_sets_f32", d.ctypes.data)


_PAIRED = {"slot_bn_fwd_f32": _pair_slot_bn, "slot_post_bwd_f32": _pair_slot_post_bwd, "wgrad_reduce_multi_f32": _pair_reduce}


class _StackCtx:
    """what _SageStack.forward / backward need of an autograd context, for the two halves of _SageStackPair"""
    needs_input_grad = ()

    def set_materialize_grads(self, v):
        pass


class _SageStackPair(torch.autograd.Function):
    """two _SageStack nodes (node outputs) on the SAME graph whose launches are recorded and issued in pairs (run_paired): the
    embedding and the assignment stack of DiffPool's first level.  forward(xa, xb, g, has_bias, nodes_a, nodes_b, n_a, *params)."""

    @staticmethod
    def forward(ctx, xa, xb, g, has_bias, nodes_a, nodes_b, n_a, *params):
        ca, cb = _StackCtx(), _StackCtx()
        with nat.deferred() as qa:
            oa = _SageStack.forward(ca, xa, g, has_bias, 0, nodes_a, *params[:n_a])
        with nat.deferred() as qb:
            ob = _SageStack.forward(cb, xb, g, has_bias, 0, nodes_b, *params[n_a:])
        run_paired(qa, qb)
        ctx.ca, ctx.cb, ctx.n_a = ca, cb, n_a
        return oa, ob

    @staticmethod
    def backward(ctx, da, db):
        n, n_a = ctx.needs_input_grad, ctx.n_a
        ctx.ca.needs_input_grad = (n[0], False, False, False, False) + tuple(n[7:7 + n_a])
        ctx.cb.needs_input_grad = (n[1], False, False, False, False) + tuple(n[7 + n_a:])
        with nat.deferred() as qa:
            ga = _SageStack.backward(ctx.ca, da)
        with nat.deferred() as qb:
            gb = _SageStack.backward(ctx.cb, db)
        run_paired(qa, qb)
        return (ga[0], gb[0], None, None, None, None, None) + tuple(ga[5:]) + tuple(gb[5:])


def sage_stack_nodes_pair(xa, xb, g, convs_a, convs_b, mask_ghost):
    """sage_stack_nodes of two stacks on one graph, launches shared where both are at the same step"""
    has_bias = convs_a[0].bias is not None
    if (convs_b[0].bias is not None) != has_bias:
        return sage_stack_nodes(xa, g, convs_a, mask_ghost), sage_stack_nodes(xb, g, convs_b, mask_ghost)
    params = []
    for c in list(convs_a) + list(convs_b):
        params.append(c.weight)
        params.append(c.bias if has_bias else c.weight.new_zeros(1))
    nodes = 2 if (mask_ghost and g.n_ghost) else 1
    # lazily built structures of the batch are built NOW: a build launch recorded inside one stack's launch record shifts it
    # against the other's, and the first step on a batch would run every launch singly (a different — split-K — product
    # kernel, so also slightly different numbers than every later step)
    if mp.ell_ok(xa) and g.val is None:
        g.ell()
        if not g.symmetric:
            g.transposed()
    return _SageStackPair.apply(xa, xb, g, has_bias, nodes, nodes, 2 * len(convs_a), *params)


def head_ok(g, convs, lin1, lin2):
    """the fused readout + head tail covers these shapes (else: sage_stack_readouts + message_passing.head2)"""
    P = sum(c.output_dim for c in convs)
    Fl = convs[-1].output_dim
    return (FUSED_TAIL and isinstance(lin1, torch.nn.Linear) and isinstance(lin2, torch.nn.Linear) and P % 4 == 0 and P <= 2048
            and Fl % 4 == 0 and Fl <= 128 and lin1.out_features <= 4096 and g.B <= 1024 and lin1.in_features == P
            and lin1.weight.data_ptr() % 16 == 0)


def sage_stack_head(x, g, convs, lin1, lin2):
    """(lin1(readout), lin2(lin1(readout))) with the readout tail and the head fused into the stack node."""
    has_bias = convs[0].bias is not None
    params = []
    for c in convs:
        params.append(c.weight)
        params.append(c.bias if has_bias else c.weight.new_zeros(1))
    vec, y = _SageStack.apply(x, g, has_bias, 4, 0, *params, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    y._tsgnn_defer_ce = True          # a cross-entropy on these logits may be folded into this node's backward (mp._SoftmaxCE)
    return vec, y
