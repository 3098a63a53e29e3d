"""Batch-of-graphs container: CSR adjacency in HBM + the row bookkeeping the kernels consume.

Replaces the reference's dense padded ``adj[B,Nmax,Nmax]`` (graph_sampler.py:102-114, shipped
host->device every step at train.py:114) and PyG's COO ``edge_index`` (Code/sag/network.py:31).

Row layouts
-----------
``packed``  rows [0, n_rows) = real nodes only, graph after graph (``graph_ptr``); the padded "ghost"
            rows of the reference (slots n >= n_b) are represented by ONE row per node slot, rows
            [n_rows, n_rows + nmax): all ghost rows of a slot carry the same value in the reference,
            so a single representative with multiplicity ``B - slot_count[n]`` reproduces the
            reference's per-slot BatchNorm statistics and max readout exactly.
``padded``  rows = B*Nmax exactly as the reference lays them out, no ghost representatives (used by the
            drop-in ``GraphConv.forward(x[B,N,F], adj[B,N,N])`` whose output must be padded too).
Feature matrices have ``total_rows = n_rows + n_ghost`` rows; ghost rows have empty neighbour lists.
"""
import os
import numpy as np
import torch

from . import _native as nat


def _i32(n, device):
    return torch.empty(int(n), dtype=torch.int32, device=device)


def _scan_ws(n, device):
    need = np.zeros(1, dtype=np.int64)
    nat.call_nostream("scan_workspace_ints", int(n), need.ctypes.data)
    return _i32(need[0], device)


def exclusive_scan(counts):
    """int32[n] -> int32[n+1] exclusive prefix sums (device)."""
    n = counts.numel()
    out = _i32(n + 1, counts.device)
    nat.call("exclusive_scan_i32", counts, n, out, _scan_ws(n, counts.device))
    return out


class GraphBatch:
    def __init__(self):
        self.B = 0
        self.nmax = 0
        self.n_rows = 0            # real rows
        self.n_ghost = 0           # ghost-slot representative rows (nmax when packed, 0 when padded)
        self.layout = "packed"
        self.sizes = None          # np.int64[B] rows of each graph in this layout
        self.graph_ptr = None      # int32[B+1]
        self.row_graph = None      # int32[n_rows]
        self.row_slot = None       # int32[n_rows]
        self.slot_count = None     # int32[nmax]: number of graphs that HAVE slot n
        self.rowptr = None         # int32[total_rows+1]
        self.col = None            # int32[nnz]
        self.val = None            # float32[nnz] or None (unit weights)
        self.nnz = 0
        self.symmetric = False
        self._t = None
        self.device = None

    @property
    def total_rows(self):
        return self.n_rows + self.n_ghost

    # ------------------------------------------------------------------ bookkeeping shared by builders
    def _set_sizes(self, sizes, nmax, device, ghosts=True):
        sizes = np.asarray(sizes, dtype=np.int64).reshape(-1)
        if sizes.size == 0 or (sizes < 0).any() or (sizes > nmax).any():
            raise ValueError("graph sizes must lie in [0, nmax]")
        self.B = int(sizes.size)
        self.nmax = int(nmax)
        self.sizes = sizes
        self.n_rows = int(sizes.sum())
        self.n_ghost = self.nmax if ghosts else 0
        gp = np.zeros(self.B + 1, dtype=np.int32)
        np.cumsum(sizes, out=gp[1:])
        hist = np.bincount(sizes, minlength=self.nmax + 1)
        slot_count = (self.B - np.cumsum(hist)[: self.nmax]).astype(np.int32)   # graphs with size > n
        self.device = device
        self.graph_ptr = torch.from_numpy(gp).to(device, non_blocking=True)
        self.slot_count = torch.from_numpy(slot_count).to(device, non_blocking=True)
        self.row_graph = _i32(max(self.n_rows, 1), device)
        self.row_slot = _i32(max(self.n_rows, 1), device)
        nat.call("row_maps", self.graph_ptr, self.B, self.n_rows, self.row_graph, self.row_slot)

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_dense_ghost1(cls, adj, sizes, assume_symmetric=False):
        """Packed rows with ONE ghost representative per graph (the row right after its n_b real rows, multiplicity
        ``row_mult = Nmax - n_b``; graphs that fill all Nmax slots get none).  For models without per-slot statistics (the
        GAT encoder): all padded rows of one graph carry the same value in every layer, so one row stands for them
        (DESIGN.md §4d).  ``g.real_sizes`` keeps n_b; ``g.sizes`` counts the representative."""
        sizes = np.asarray(sizes, dtype=np.int64).reshape(-1)
        nmax = adj.size(1)
        if (sizes > nmax).any() or (sizes < 0).any():
            raise ValueError("graph sizes must lie in [0, nmax]")
        has_ghost = sizes < nmax
        g = cls.from_dense(adj, sizes + has_ghost, layout="packed", assume_symmetric=assume_symmetric, ghosts=False)
        mult = np.ones(g.n_rows, dtype=np.float32)
        ends = np.cumsum(sizes + has_ghost) - 1
        mult[ends[has_ghost]] = (nmax - sizes[has_ghost]).astype(np.float32)
        g.row_mult = torch.from_numpy(mult).to(adj.device)
        g.real_sizes = sizes
        return g

    @classmethod
    def from_dense(cls, adj, sizes=None, layout="packed", assume_symmetric=False, ghosts=None):
        """adj: float32 [B,Nmax,Nmax] on the GPU (the tensor train.py:114 uploads).
        layout='packed' needs ``sizes`` (= batch_num_nodes) and assumes zero padding outside
        [:n_b,:n_b] (what GraphSampler produces); layout='padded' keeps all B*Nmax rows."""
        if adj.dim() != 3 or adj.size(1) != adj.size(2):
            raise ValueError("adj must be [B,N,N]")
        adj = adj.contiguous().float()
        B, nmax = adj.size(0), adj.size(1)
        g = cls()
        g.layout = layout
        if layout == "padded":
            sizes = np.full(B, nmax, dtype=np.int64)
        elif sizes is None:
            raise ValueError("packed layout needs the per-graph node counts")
        g._set_sizes(sizes, nmax, adj.device, ghosts=(layout != "padded") if ghosts is None else bool(ghosts))
        if g.B != B:
            raise ValueError("len(sizes) != batch size of adj")
        R = g.n_rows
        cnt = torch.zeros(g.total_rows, dtype=torch.int32, device=adj.device)
        nat.call("dense_adj_count", adj, B, nmax, g.graph_ptr, g.row_graph, R, cnt)
        g.rowptr = exclusive_scan(cnt)
        g.nnz = int(g.rowptr[-1].item())
        g.col = _i32(max(g.nnz, 1), adj.device)
        g.val = torch.empty(max(g.nnz, 1), dtype=torch.float32, device=adj.device)
        nat.call("dense_adj_fill", adj, B, nmax, g.graph_ptr, g.row_graph, R, g.rowptr, g.col, g.val)
        g.symmetric = bool(assume_symmetric)
        return g

    @classmethod
    def from_edge_index(cls, edge_index, num_nodes, sizes=None, nmax=None, assume_symmetric=False, ghosts=False):
        """edge_index: int64 [2,E] (PyG: row 0 = source j, row 1 = target i).  CSR row = target."""
        g = cls()
        g.layout = "packed"
        dev = edge_index.device
        num_nodes = int(num_nodes)
        if sizes is None:
            sizes = np.array([num_nodes], dtype=np.int64)
        sizes = np.asarray(sizes, dtype=np.int64)
        if int(sizes.sum()) != num_nodes:
            raise ValueError("sizes must sum to num_nodes")
        g._set_sizes(sizes, int(nmax) if nmax is not None else int(max(1, sizes.max())), dev, ghosts=ghosts)
        ei = edge_index.contiguous()
        E = int(ei.size(1))
        R = num_nodes
        cnt = torch.zeros(g.total_rows, dtype=torch.int32, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        key, other = ei[1], ei[0]
        nat.call("coo_count", key, E, R, cnt, bad)
        g.rowptr = exclusive_scan(cnt)
        g.nnz = E
        g.col = _i32(max(E, 1), dev)
        g.eid = _i32(max(E, 1), dev)
        nat.call("coo_fill", key, other, E, R, g.rowptr, _i32(max(R, 1), dev), g.col, g.eid)
        if int(bad.item()) != 0:
            raise IndexError("edge_index contains node ids outside [0, num_nodes)")
        g.val = None
        g.symmetric = bool(assume_symmetric)
        return g

    _uniform_cache = {}

    @classmethod
    def uniform(cls, B, K, device):
        """cached structure-only batch of B graphs with K rows each, no ghost rows (pooled DiffPool levels, apply_bn):
        built once per (B, K, device), so steady-state steps do no host->device copies (hipGraph-capturable)."""
        key = (int(B), int(K), str(device))
        g = cls._uniform_cache.get(key)
        if g is None:
            g = cls._uniform_cache[key] = cls.structure_only(np.full(int(B), int(K), dtype=np.int64), int(K), device, ghosts=False)
        return g

    @classmethod
    def structure_only(cls, sizes, nmax, device, ghosts=True):
        """Row bookkeeping without an adjacency (slot batch-norm / readout of stand-alone tensors)."""
        g = cls()
        g.layout = "packed" if ghosts else "padded"
        g._set_sizes(sizes, nmax, device, ghosts=ghosts)
        g.rowptr = torch.zeros(g.total_rows + 1, dtype=torch.int32, device=device)
        g.col = _i32(1, device)
        g.symmetric = True
        return g

    @classmethod
    def from_csr(cls, rowptr, col, val, sizes, nmax, assume_symmetric=True):
        """CSR-native ingest (synthetic loader / preprocessed datasets): rowptr int32[n_rows + nmax + 1]
        (the nmax ghost-slot rows are empty: trailing entries repeat nnz), col int32[nnz] in packed row ids."""
        g = cls()
        g._set_sizes(sizes, nmax, rowptr.device)
        if rowptr.numel() != g.total_rows + 1:
            raise ValueError("rowptr must have n_rows + nmax + 1 entries (empty ghost rows included)")
        g.rowptr, g.col, g.val = rowptr, col, val
        g.nnz = int(col.numel())
        g.symmetric = bool(assume_symmetric)
        return g

    # ------------------------------------------------------------------ row slabs per graph (ragged contractions)
    def row_slabs(self, rows_per_slab=64):
        """(slab_row_ptr int32[nslab+1], seg_slab_ptr int32[B+1], nslab): every graph's real rows cut into slabs of at
        most ``rows_per_slab`` rows (one workgroup of MFMA work each)."""
        cache = getattr(self, "_slabs", None)
        if cache is None or cache[3] != rows_per_slab:
            starts, seg = [], [0]
            off = 0
            for n in self.sizes:
                n = int(n)
                starts.extend(range(off, off + n, rows_per_slab))
                seg.append(len(starts))
                off += n
            # consecutive slabs are contiguous inside a graph and graphs are contiguous too, so the slab starts followed
            # by the total row count are exactly the [begin, end) boundaries the kernel reads
            srp = np.asarray(starts + [off], dtype=np.int32)
            cache = self._slabs = (torch.from_numpy(srp).to(self.device),
                                   torch.from_numpy(np.asarray(seg, dtype=np.int32)).to(self.device), len(starts), rows_per_slab)
        return cache[0], cache[1], cache[2]

    # ------------------------------------------------------------------ fixed-width (ELL) view
    def ell(self):
        """(ell_col int32[R,W], W, tail) for the unit-weight aggregation fast path, or None when the graph is
        weighted.  tail = (tail_ptr, tail_col) CSR of the entries beyond W per row, or None."""
        if self.val is not None:
            return None
        if getattr(self, "_ell", None) is None:
            R = self.total_rows
            deg = self.rowptr[1:] - self.rowptr[:-1]
            maxdeg = int(deg.max().item()) if R > 0 else 0
            W = 4 if maxdeg <= 4 else (8 if maxdeg <= 8 else 16)
            ell = _i32(max(R * W, 1), self.device)
            tail_cnt = torch.zeros(R, dtype=torch.int32, device=self.device) if maxdeg > W else None
            nat.call("csr_to_ell", self.rowptr, self.col, R, W, ell, tail_cnt)
            tail = None
            if tail_cnt is not None:
                tail_ptr = exclusive_scan(tail_cnt)
                tail_col = _i32(max(int(tail_ptr[-1].item()), 1), self.device)
                nat.call("csr_tail_fill", self.rowptr, self.col, tail_ptr, R, W, tail_col)
                tail = (tail_ptr, tail_col)
            self._ell = (ell, W, tail)
        return self._ell

    def ell_slots(self):
        """(table, tail_col) of the neighbour table with the neighbour's SLOT beside its row: entry = slot << 20 | row (empty
        entries stay -1), or None (>= 2^20 rows, > 1024 slots).  Operand of the layers whose input's slot batch-norm is formed on
        the fly (tsgnn_sage_layer_fwd_bn_f32).  Built once per batch structure, outside the step."""
        if getattr(self, "_ell_slots", None) is None:
            e = self.ell()
            if e is None or self.total_rows >= (1 << 20) or self.nmax > 1024 or self.row_slot is None:
                self._ell_slots = False
            else:
                ell, W, tail = e
                slot_of = self.row_slot.to(torch.int64)

                def pack(ids32, n):
                    ids = ids32[:n].to(torch.int64)
                    ok = (ids >= 0) & (ids < self.n_rows)              # (nothing aggregates from a ghost row)
                    packed = torch.where(ok, (slot_of[ids.clamp(0, max(self.n_rows - 1, 0))] << 20) | ids, torch.full_like(ids, -1))
                    out = _i32(max(ids32.numel(), 1), self.device)
                    out[:n] = packed.to(torch.int32)
                    return out
                table = pack(ell, self.total_rows * W)
                tcol = pack(tail[1], int(tail[0][-1].item())) if tail is not None else None
                self._ell_slots = (table, tcol)
        return self._ell_slots if self._ell_slots is not False else None

    def du_map(self, other_blocks):
        """(du_map int32[n], n, chunk rows) for tsgnn_head2_bwd_du_map_f32, or (None, 0, 64): the non-empty (graph, chunk) pairs of the
        last layer's dU role — chunk 0 of every graph, then one entry per further 64 (128) rows —, 64-row chunks while the launch
        (other_blocks = the head's own workgroups) stays within one workgroup per compute unit, else 128.  Host side, once per batch
        structure; batches whose sizes live on the device only (ingest slots) use the dense grid."""
        if self.sizes is None or getattr(self, "ghost_slots_fixed", None) is not None:
            # sizes on the device only (a capacity-padded ingest slot): n = -1 asks the kernel to resolve the same compact order itself
            return None, (-1 if self.B <= 64 else 0), 64
        cache = self.__dict__.setdefault("_du_maps", {})
        key = int(other_blocks)
        if key not in cache:
            sizes = np.asarray(self.sizes, dtype=np.int64)
            ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
            chunk = 64
            if key + int(np.maximum(1, -(-sizes // 64)).sum()) > ncu:
                chunk = 128
            if int(sizes.max()) > 255 * chunk or self.B >= (1 << 22):
                cache[key] = (None, 0, 64)
            else:
                ent = [(b << 8) | c for b in range(self.B) for c in range(max(1, -(-int(sizes[b]) // chunk)))]
                cache[key] = (torch.from_numpy(np.asarray(ent, dtype=np.int32)).to(self.device), len(ent), chunk)
        return cache[key]

    def readout_map(self, nslots, fill_rows):
        """(ro_map int32[B * chunks], chunk size) for tsgnn_sage_layer_fwd_bn_f32, or (None, 0): readout block k of that launch is
        workgroup n_gemm + k and therefore sits on XCD (n_gemm + k) % 8; the row panels of XCD x are a contiguous range of the batch's
        rows (xcd_remap, csrc/common.h), i.e. whole graphs — so the blocks that scan graph b are given slots on the XCD whose L2 already
        holds b's rows.  Host side, once per batch structure; batches whose structure lives on the device only (ingest slots) get the
        plain order."""
        if self.sizes is None:
            return None, 0
        key = (int(nslots), int(fill_rows))
        cache = self.__dict__.setdefault("_ro_maps", {})
        if key not in cache:
            ch = np.zeros(1, dtype=np.int32)
            ng = np.zeros(1, dtype=np.int32)
            nat.call_nostream("sage_layer_fwd_bn_plan", int(self.n_rows), int(fill_rows), int(self.B), int(nslots), ch.ctypes.data, ng.ctypes.data)
            ch, ng = int(ch[0]), int(ng[0])
            chunks = -(-int(nslots) // ch)
            npan = -(-self.n_rows // 32)
            ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
            if npan > ncu and npan - ncu <= ncu // 2 and os.environ.get("TSGNN_HALF_PANELS", "1") != "0":
                npan = ncu & ~7            # (rowgemm_body.h panel_split: the full panels; the rows behind them go as 16-row units on any XCD)
            q, r = divmod(npan, 8)
            start = np.zeros(9, dtype=np.int64)
            for x in range(8):
                start[x + 1] = start[x] + (q + 1 if x < r else q)
            gp = np.concatenate([[0], np.cumsum(np.asarray(self.sizes, dtype=np.int64))])
            mid_panel = ((gp[:-1] + gp[1:]) // 2) // 32
            xcd_of_graph = np.clip(np.searchsorted(start, mid_panel, side="right") - 1, 0, 7)
            want = [[] for _ in range(8)]                       # work items per XCD
            for b in range(self.B):
                for c in range(chunks):
                    want[int(xcd_of_graph[b])].append(b * chunks + c)
            items = self.B * chunks
            slots = [[k for k in range(items) if (ng + k) % 8 == x] for x in range(8)]
            ro = np.full(items, -1, dtype=np.int32)
            left = []
            for x in range(8):
                n = min(len(want[x]), len(slots[x]))
                ro[slots[x][:n]] = want[x][:n]
                left += want[x][n:]
                slots[x] = slots[x][n:]
            free = [k for x in range(8) for k in slots[x]]
            ro[free] = left
            assert (np.sort(ro) == np.arange(items)).all()
            cache[key] = (torch.from_numpy(ro).to(self.device), ch)
        return cache[key]

    def bn_workspace(self, B, L, Fh, Fl, nslots):
        """persistent device buffers of the fused slot batch-norms of a stack on this batch (sage_stack.py): the integer sums
        [L-1][2 * nslots] and the ghost rows' numbers [L-1][2] — zero between steps, the step's own head launch clears them —, and
        the packed max-readout buffer (zero between steps as well).  `dirty`: a forward that did not reach its head left them
        in an undefined state; the next forward clears them explicitly."""
        key = (B, L, Fh, Fl, nslots)
        ws = getattr(self, "_bn_ws", None)
        if ws is None or ws["key"] != key:
            words = (L - 1) * 2 * nslots
            from . import message_passing as mp        # (zeroed again if a launch is reported aborted: mp.register_clear_on_error)
            ws = {"key": key, "sums": mp.register_clear_on_error(torch.zeros(max(words, 2), dtype=torch.int64, device=self.device)),
                  "ghost": mp.register_clear_on_error(torch.zeros(max(2 * (L - 1), 2), dtype=torch.float32, device=self.device)),
                  "packed": mp.register_clear_on_error(torch.zeros(B * ((L - 1) * Fh + Fl) + Fl, dtype=torch.int64, device=self.device)),
                  "dirty": False}
            self._bn_ws = ws
        return ws

    # ------------------------------------------------------------------ transpose (for dX = A^T dY)
    def _ensure_transpose(self):
        if self._t is None:
            R1 = self.total_rows
            dev = self.device
            rowptr_t = _i32(R1 + 1, dev)
            col_t = _i32(max(self.nnz, 1), dev)
            val_t = torch.empty(max(self.nnz, 1), dtype=torch.float32, device=dev) if self.val is not None else None
            src_e = _i32(max(self.nnz, 1), dev)
            nat.call("csr_transpose", self.rowptr, self.col, self.val, R1, R1, self.nnz, rowptr_t, col_t, val_t,
                     src_e, _i32(max(R1, 1), dev), _scan_ws(R1, dev))
            self._t = (rowptr_t, col_t, val_t)
            self.src_e_t = src_e
        return self._t

    def transpose_map(self):
        """explicit CSR of A^T plus src_e_t[p] = entry of A that transposed entry p came from
        (always computed, even for symmetric graphs: edge-softmax needs the entry pairing)."""
        rp, col, _ = self._ensure_transpose()
        return rp, col, self.src_e_t

    def transposed(self, val="graph"):
        """CSR of A^T (rowptr, col, val) for dX = A^T dY; ``val`` = per-entry weights aligned with
        self.col (default: the graph's own).  Symmetric graphs (with symmetric weights) reuse A."""
        if isinstance(val, str):
            val = self.val
        if self.symmetric:
            return self.rowptr, self.col, val
        t = self._ensure_transpose()
        if val is self.val:
            return t
        return t[0], t[1], (val[self.src_e_t.long()] if val is not None else None)

    # ------------------------------------------------------------------ feature (un)packing helpers
    def new_features(self, feat, zero=False):
        f = torch.zeros if zero else torch.empty
        return f(self.total_rows, feat, dtype=torch.float32, device=self.device)
