"""Mini-batch ingest overlapped with the training step (graph_sampler.py:102-114, train.py:110-119).

The reference builds ``adj[B, Nmax, Nmax]`` per item in python and ships it to the device every step.  Here:

  host    ``tsgnn_host_collate_tu`` (C, csrc/ingest.hip) writes the mini-batch of a CSR-resident dataset (``tu_data.TUDataset``)
          straight in device layout into a pinned staging buffer: graph pointers, slot counts, row maps, the fixed-width
          neighbour table (+ tail), node labels, graph labels.
  copy    ONE host->device copy per batch on a copy stream, plus one launch that expands the node labels into the one-hot
          feature rows (train.py:227-231) — 4 bytes per node cross PCIe, not 4 * F.
  step    a capacity-padded batch (``CapacityBatch``): the row count every kernel is launched with is the slot's capacity, the
          rows beyond the batch's own are padding that belongs to a dummy graph, so ONE hipGraph per slot replays every batch.
  overlap two slots: while the step of batch k replays from slot k % 2, batch k + 1 is collated and uploaded into the other.

No dense ``[Nmax, Nmax]`` tensor exists anywhere on this path.
"""
import numpy as np
import torch

from . import _native as nat
from .graph import GraphBatch
from .tu_data import TUDataset

ELL_W = 16


def layout(B, nmax, row_cap, ell_w=ELL_W, tail_cap=4096):
    off = np.zeros(10, dtype=np.int64)
    nat.call_nostream("ingest_layout", int(B), int(nmax), int(row_cap), int(ell_w), int(tail_cap), off.ctypes.data)
    return [int(v) for v in off]


def host_collate(ds, ids, B, nmax, row_cap, staging, ell_w=ELL_W, tail_cap=4096):
    """staging: int32 host array / tensor of layout(...)[9] words.  -> (rows, directed edges, tail entries, largest graph)"""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    if len(ids) != B:
        raise ValueError("a mini-batch of this slot has exactly %d graphs" % B)
    out = np.zeros(4, dtype=np.int64)
    ptr = staging.data_ptr() if isinstance(staging, torch.Tensor) else staging.ctypes.data
    nl = ds.node_label
    nat.call_nostream("host_collate_tu", ds.graph_ptr.ctypes.data, ds.rowptr.ctypes.data, ds.col.ctypes.data,
                      nl.ctypes.data if nl is not None else None, ds.graph_label.ctypes.data, ids.ctypes.data, int(B), int(nmax),
                      int(row_cap), int(ell_w), int(tail_cap), ptr, out.ctypes.data)
    return int(out[0]), int(out[1]), int(out[2]), int(out[3])


def synthetic_dataset(seed, n_graphs, shape="DD", nmax=1000):
    """a TU-style dataset (one CSR over all graphs, integer node labels) of synthetic graphs of the given shape"""
    from . import synthetic
    hb = synthetic.host_batch(seed, n_graphs, shape, nmax)
    n = int(hb["sizes"].sum())
    fin = hb["fin"]
    rng = np.random.default_rng(seed + 1)
    gp = np.zeros(n_graphs + 1, dtype=np.int64)
    np.cumsum(hb["sizes"], out=gp[1:])
    ds = TUDataset(gp, hb["rowptr"][: n + 1].astype(np.int64), hb["col"].astype(np.int64), hb["label"].astype(np.int64),
                   rng.integers(0, fin, size=n).astype(np.int64), None, fin)
    return ds


class CapacityBatch:
    """One in-flight mini-batch: fixed-capacity device buffers (one allocation, refreshed by one copy), the pinned host staging
    buffer of the same layout, and a ``GraphBatch`` view over the device buffer that the encoders take."""

    def __init__(self, B, nmax, row_cap, fin, device, ghost_slots=None, tail_cap=4096):
        self.B, self.nmax, self.row_cap, self.fin, self.tail_cap = int(B), int(nmax), int(row_cap), int(fin), int(tail_cap)
        off = layout(B, nmax, row_cap, ELL_W, tail_cap)
        self.words = off[9]
        self.dev = torch.zeros(self.words, dtype=torch.int32, device=device)
        self.host = torch.zeros(self.words, dtype=torch.int32).pin_memory() if device.type == "cuda" else torch.zeros(self.words, dtype=torch.int32)
        R = self.row_cap + self.nmax
        d = self.dev
        g = GraphBatch()
        g.B, g.nmax, g.n_rows, g.n_ghost, g.layout = self.B, self.nmax, self.row_cap, self.nmax, "packed"
        g.device = device
        g.graph_ptr = d[off[0]:off[0] + B + 2]
        g.slot_count = d[off[1]:off[1] + nmax]
        g.row_graph = d[off[2]:off[2] + row_cap]
        g.row_slot = d[off[3]:off[3] + row_cap]
        g._ell = (d[off[4]:off[4] + R * ELL_W], ELL_W, (d[off[5]:off[5] + R + 1], d[off[6]:off[6] + max(tail_cap, 1)]))
        g.rowptr = g.col = g.val = None                       # the step reads the neighbour table; CSR-only paths would fail loudly
        g.nnz = 0
        g.symmetric = True
        g.sizes = None
        g.ghost_slots_fixed = self.nmax if ghost_slots is None else int(ghost_slots)
        self.g = g
        self.node_label = d[off[7]:off[7] + row_cap]
        self.label = d[off[8]:off[8] + 2 * B].view(torch.int64)
        ld = (fin + 3) // 4 * 4
        self.x = torch.zeros(R, ld, dtype=torch.float32, device=device)
        self.rows = self.edges = 0
        if device.type == "cuda":
            self.uploaded = torch.cuda.Event()      # the copy + feature expansion of the current batch are enqueued / done
            self.consumed = torch.cuda.Event()      # the step that read this slot has finished
            self._pending = False

    def collate(self, ds, ids):
        """host: write the mini-batch ``ids`` of ``ds`` into the staging buffer (waits for the previous upload from it)"""
        if getattr(self, "_pending", False):
            self.uploaded.synchronize()
        self.rows, self.edges, self.tail, self.largest = host_collate(ds, ids, self.B, self.nmax, self.row_cap, self.host,
                                                                     ELL_W, self.tail_cap)
        if self.largest + 1 > self.g.ghost_slots_fixed:
            raise ValueError("a graph of %d nodes exceeds the slot's fixed ghost-slot bound %d" % (self.largest, self.g.ghost_slots_fixed))

    def collate_async(self, pool, ds, ids):
        """queue the collate of ``ids`` on a native worker (``CollatePool``); ``collate_wait`` before ``upload``"""
        self._ids = np.ascontiguousarray(ids, dtype=np.int64)
        if len(self._ids) != self.B:
            raise ValueError("a mini-batch of this slot has exactly %d graphs" % self.B)
        self._out = np.zeros(4, dtype=np.int64)
        self._ticket = np.zeros(1, dtype=np.int64)
        nl = ds.node_label
        after = self.uploaded.cuda_event if getattr(self, "_pending", False) else None
        nat.call_nostream("collate_pool_submit", pool.handle, ds.graph_ptr.ctypes.data, ds.rowptr.ctypes.data, ds.col.ctypes.data,
                          nl.ctypes.data if nl is not None else None, ds.graph_label.ctypes.data, self._ids.ctypes.data, int(self.B),
                          int(self.nmax), int(self.row_cap), ELL_W, int(self.tail_cap), self.host.data_ptr(), self._out.ctypes.data,
                          after, self._ticket.ctypes.data)
        self._pool = pool

    def collate_wait(self):
        nat.call_nostream("collate_pool_wait", self._pool.handle, int(self._ticket[0]))
        self.rows, self.edges, self.tail, self.largest = (int(v) for v in self._out)
        if self.largest + 1 > self.g.ghost_slots_fixed:
            raise ValueError("a graph of %d nodes exceeds the slot's fixed ghost-slot bound %d" % (self.largest, self.g.ghost_slots_fixed))

    def upload(self, stream):
        """enqueue on ``stream``: staging -> device (one copy), node labels -> one-hot feature rows"""
        with torch.cuda.stream(stream):
            nat.call("ingest_upload_f32", self.dev, self.host.data_ptr(), int(self.words), self.node_label, int(self.rows),
                     int(self.row_cap + self.nmax), int(self.fin), self.x, self.x.stride(0))
            self.uploaded.record(stream)
            self._pending = True


class CollatePool:
    """native worker threads for the host collate (csrc/ingest.hip): they run ahead of the thread that drives the GPU"""

    def __init__(self, threads=2):
        import ctypes
        h = ctypes.c_void_p()
        nat.call_nostream("collate_pool_create", int(threads), ctypes.addressof(h))
        self.handle = h.value

    def close(self):
        if self.handle:
            nat.call_nostream("collate_pool_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 - interpreter shutdown
            pass


class IngestPipeline:
    """Double-buffered training loop: every step consumes a NEW mini-batch drawn from ``ds``."""

    def __init__(self, model, trainer, ds, batch, nmax, device, schedule, depth=3, make_loss=None):
        from .data_parallel import GraphedStep
        self.ds, self.B, self.schedule = ds, int(batch), schedule
        sizes = ds.sizes
        rows_max = max(int(sizes[np.asarray(ids)].sum()) for ids in schedule)
        self.row_cap = (rows_max + 31) // 32 * 32
        ghost = min(int(nmax), int(sizes.max()) + 1)
        self.compute = torch.cuda.Stream()
        self.copy = torch.cuda.Stream()
        self.slots = [CapacityBatch(batch, nmax, self.row_cap, ds.num_node_labels, device, ghost_slots=ghost) for _ in range(depth)]
        self.steps = []
        if make_loss is None:
            make_loss = lambda s: (lambda: model.loss(model(s.x, s.g)[1], s.label))
        for s in self.slots:
            s.collate(ds, schedule[0])
            s.upload(self.compute)
            torch.cuda.synchronize()
            self.steps.append(GraphedStep(trainer, make_loss(s), warmup=2, stream=self.compute))

    def run(self, schedule=None, workers=2):
        """enqueue one step per entry of the schedule; returns after the last step is enqueued (caller synchronises).
        The host collate of the batches ahead runs on ``workers`` native threads, so the enqueueing thread only uploads and
        replays; workers = 0 collates inline."""
        sched = self.schedule if schedule is None else schedule
        depth = len(self.slots)
        pool = self._get_pool(workers) if workers else None
        if pool is not None:
            for k in range(min(depth, len(sched))):
                self.slots[k].collate_async(pool, self.ds, sched[k])
        for k, ids in enumerate(sched):
            s, gs = self.slots[k % depth], self.steps[k % depth]
            if pool is not None:
                s.collate_wait()                              # this batch is in the slot's staging buffer
            else:
                s.collate(self.ds, ids)
            self.copy.wait_event(s.consumed)                  # the step that last read this slot is done
            s.upload(self.copy)
            self.compute.wait_event(s.uploaded)
            gs.step()
            s.consumed.record(self.compute)
            if pool is not None and k + depth < len(sched):   # the slot's next batch (the worker waits for this upload first)
                s.collate_async(pool, self.ds, sched[k + depth])
        return len(sched)

    def _get_pool(self, workers):
        if getattr(self, "_cpool", None) is None or self._cpool_n != workers:
            if getattr(self, "_cpool", None) is not None:
                self._cpool.close()
            self._cpool, self._cpool_n = CollatePool(workers), workers
        return self._cpool
