"""Mini-batch ingest overlapped with the training step (graph_sampler.py:102-114, train.py:110-119).

The reference builds ``adj[B, Nmax, Nmax]`` per item in python and ships it to the device every step.  Here:

  host    native worker threads (``CollatePool``) run ``tsgnn_host_collate_compact`` (C, csrc/ingest.hip) for the batches AHEAD:
          the mini-batch of a CSR-resident dataset (``tu_data.TUDataset``) as a compact CSR batch in a pinned staging buffer
          (graph pointers, slot counts, row pointers, columns, node labels, graph labels: ~0.3 MB for 32 DD graphs).
  pull    kernels of the step's OWN hipGraph read the staging buffer over PCIe (a flat copy into a device mirror) and expand it on
          the device: row maps, the fixed-width neighbour table, the one-hot feature rows of the TU "node-label" mode
          (train.py:227-231).  The copy of the NEXT batch rides as extra workgroups of the current step's first hidden-layer
          product (csrc/ingest_rider.h), so its PCIe round trip is off the step's critical path.  No copy engine, no second
          stream, no event per step: on this stack the copy-engine -> shader hand-over and cross-stream events cost more than
          the transfer (measured, scripts/ingest_profile.py); the host learns that a staging buffer may be refilled from a
          sequence word the expand launch echoes into pinned memory.
  step    a capacity-padded batch (``CapacityBatch``): the row count every kernel is launched with is the slot's capacity, the
          rows beyond the batch's own are padding that belongs to a dummy graph, so ONE hipGraph per slot replays every batch.
  overlap three slots: while batch k replays from slot k % 3, the workers fill the other slots' staging buffers.

No dense ``[Nmax, Nmax]`` tensor exists anywhere on this path.
"""
import numpy as np
import torch

from . import _native as nat
from . import message_passing as mp
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


def compact_layout(B, nmax, row_cap, edge_cap, tail_cap=4096):
    off = np.zeros(10, dtype=np.int64)
    nat.call_nostream("ingest_compact_layout", int(B), int(nmax), int(row_cap), int(edge_cap), int(tail_cap), off.ctypes.data)
    return [int(v) for v in off]


def host_collate_compact(ds, ids, B, nmax, row_cap, edge_cap, staging, ell_w=ELL_W, tail_cap=4096):
    """the mini-batch in the compact (CSR) staging layout.  -> (rows, directed edges, tail entries, largest graph)"""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    if len(ids) != B:
        raise ValueError("a mini-batch of this slot has exactly %d graphs" % B)
    out = np.zeros(4, dtype=np.int64)
    ptr = staging.data_ptr() if isinstance(staging, torch.Tensor) else staging.ctypes.data
    nl = ds.node_label
    nat.call_nostream("host_collate_compact", ds.graph_ptr.ctypes.data, ds.rowptr.ctypes.data, ds.col.ctypes.data,
                      nl.ctypes.data if nl is not None else None, ds.graph_label.ctypes.data, ids.ctypes.data, int(B), int(nmax),
                      int(row_cap), int(edge_cap), int(ell_w), int(tail_cap), ptr, out.ctypes.data)
    return int(out[0]), int(out[1]), int(out[2]), int(out[3])


class CapacityBatch:
    """One in-flight mini-batch slot: the pinned host staging buffer (compact layout), its device mirror, the fixed-capacity
    device arrays the step reads, and a ``GraphBatch`` view over them.  ``pull()`` enqueues the two launches that bring the
    staged batch in; called at the head of the step (inside the slot's hipGraph), so replaying the graph IS the upload."""

    def __init__(self, B, nmax, row_cap, edge_cap, fin, device, ghost_slots=None, tail_cap=4096):
        self.B, self.nmax, self.row_cap, self.edge_cap = int(B), int(nmax), int(row_cap), int(edge_cap)
        self.fin, self.tail_cap = int(fin), int(tail_cap)
        off = compact_layout(B, nmax, row_cap, edge_cap, tail_cap)
        self.words = off[9]
        cuda = device.type == "cuda"
        self.host = torch.zeros(self.words, dtype=torch.int32).pin_memory() if cuda else torch.zeros(self.words, dtype=torch.int32)
        self.mirror = torch.zeros(self.words, dtype=torch.int32, device=device)
        R = self.row_cap + self.nmax
        i32 = lambda n: torch.zeros(int(n), dtype=torch.int32, device=device)
        m = self.mirror
        g = GraphBatch()
        g.B, g.nmax, g.n_rows, g.n_ghost, g.layout = self.B, self.nmax, self.row_cap, self.nmax, "packed"
        g.device = device
        g.graph_ptr, g.slot_count = m[off[1]:off[1] + B + 2], m[off[2]:off[2] + nmax]      # plain copies: views of the mirror
        g.row_graph, g.row_slot = i32(row_cap), i32(row_cap)
        self.tail_col = m[off[8]:off[8] + max(tail_cap, 1)]
        g._ell = (torch.full((R * ELL_W,), -1, dtype=torch.int32, device=device), ELL_W, (i32(R + 1), self.tail_col))
        g.rowptr = g.col = g.val = None                       # the step reads the neighbour table; CSR-only paths would fail loudly
        g.nnz = 0
        g.symmetric = True
        g.sizes = None
        g.ghost_slots_fixed = self.nmax if ghost_slots is None else int(ghost_slots)
        # the neighbour table of this batch is rewritten on the device every step, and so is its slot-annotated copy (the operand
        # of the fused slot batch-norm path, GraphBatch.ell_slots): the expansion writes both
        self.ell_slots = torch.full((R * ELL_W,), -1, dtype=torch.int32, device=device)
        self.tail_slots = i32(max(tail_cap, 1))
        g._ell_slots = (self.ell_slots, self.tail_slots) if (R < (1 << 20) and self.nmax <= 1024) else False
        self.g = g
        self.node_label = m[off[5]:off[5] + row_cap]
        self.label = m[off[3]:off[3] + 2 * B].view(torch.int64)
        ld = (fin + 3) // 4 * 4
        self.x = torch.zeros(R, ld, dtype=torch.float32, device=device)
        self.rows = self.edges = 0
        # hand-shake with the device without an event per step: whoever collates a batch stamps it with a sequence number
        # (header word 4); the expand launch echoes the number of the batch it has pulled into `ack` (pinned host memory); the
        # staging buffer is refilled only once the last stamped batch was echoed
        self._seq_word = off[0] + 4
        self.seq = 0                                # sequence number of the batch in the staging buffer
        self.ack = torch.zeros(1, dtype=torch.int64).pin_memory() if cuda else None
        self._ack_np = self.ack.numpy() if cuda else None
        self._replayed = True                       # the staged batch has been enqueued for pulling at least once

    # ------------------------------------------------------------------ host side
    def _check(self):
        if self.largest + 1 > self.g.ghost_slots_fixed:
            raise ValueError("a graph of %d nodes exceeds the slot's fixed ghost-slot bound %d" % (self.largest, self.g.ghost_slots_fixed))

    def _wait_pulled(self):
        """the batch in the staging buffer, if a pull of it was enqueued, has been pulled (bounded wait)"""
        if self.ack is None or self.seq == 0 or not self._replayed:
            return
        import time
        t0 = time.perf_counter()
        while int(self._ack_np[0]) < self.seq:
            if time.perf_counter() - t0 > 20.0:
                raise RuntimeError("ingest: the device did not acknowledge batch %d of this slot within 20 s" % self.seq)

    def collate(self, ds, ids):
        """write the mini-batch ``ids`` of ``ds`` into the staging buffer (after the pull of the batch it holds has finished)"""
        self._wait_pulled()
        self.rows, self.edges, self.tail, self.largest = host_collate_compact(ds, ids, self.B, self.nmax, self.row_cap, self.edge_cap,
                                                                             self.host, ELL_W, self.tail_cap)
        self.seq += 1
        self.host[self._seq_word] = self.seq
        self._replayed = False
        self._check()

    def collate_async(self, pool, ds, ids):
        """queue the collate of ``ids`` on a native worker (``CollatePool``); ``collate_wait`` before the step is replayed"""
        self._ids = np.ascontiguousarray(ids, dtype=np.int64)
        if len(self._ids) != self.B:
            raise ValueError("a mini-batch of this slot has exactly %d graphs" % self.B)
        self._out = np.zeros(4, dtype=np.int64)
        self._ticket = np.zeros(1, dtype=np.int64)
        nl = ds.node_label
        # the worker waits until the batch this buffer holds has been pulled (if a pull of it was ever enqueued)
        target = self.seq if self._replayed else 0
        self.seq += 1
        self._replayed = False
        nat.call_nostream("collate_pool_submit_ack", pool.handle, ds.graph_ptr.ctypes.data, ds.rowptr.ctypes.data, ds.col.ctypes.data,
                          nl.ctypes.data if nl is not None else None, ds.graph_label.ctypes.data, self._ids.ctypes.data, int(self.B),
                          int(self.nmax), int(self.row_cap), int(self.edge_cap), ELL_W, int(self.tail_cap), self.host.data_ptr(),
                          self._out.ctypes.data, self.ack.data_ptr(), int(target), int(self.seq), self._ticket.ctypes.data)
        self._pool = pool

    def collate_wait(self):
        nat.call_nostream("collate_pool_wait", self._pool.handle, int(self._ticket[0]))
        self.rows, self.edges, self.tail, self.largest = (int(v) for v in self._out)
        self._check()

    # ------------------------------------------------------------------ device side
    def pull(self):
        """enqueue (current stream; capturable) the pull of the staged batch and its expansion into the slot's arrays"""
        g = self.g
        ell, _, (tail_ptr, tail_col) = g._ell
        args = (self.host.data_ptr(), self.mirror, self.B, self.nmax, self.row_cap, self.edge_cap, ELL_W, self.tail_cap, g.row_graph,
                g.row_slot, ell, tail_ptr, self.ell_slots, self.tail_slots, self.fin, self.x, self.x.stride(0), self.ack.data_ptr())
        nat.call("ingest_pull_expand_ack_f32", *args)
        if not torch.cuda.is_current_stream_capturing():
            self._replayed = True

    def _expand_args(self):
        g = self.g
        ell, _, (tail_ptr, tail_col) = g._ell
        return (self.mirror, self.B, self.nmax, self.row_cap, self.edge_cap, ELL_W, self.tail_cap, g.row_graph, g.row_slot, ell, tail_ptr,
                self.ell_slots, self.tail_slots, self.fin, self.x, self.x.stride(0), self.ack.data_ptr())

    def expand(self):
        """the expansion alone, of a batch that is already in the mirror (pulled as passengers of the previous step, or by pull_only)"""
        nat.call("ingest_expand_ack_f32", *self._expand_args())
        if not torch.cuda.is_current_stream_capturing():
            self._replayed = True

    def pull_only(self):
        """the flat copy staging buffer -> mirror as a launch of its own (current stream)"""
        nat.call("ingest_pull_f32", self.host.data_ptr(), self.mirror, self.B, self.nmax, self.row_cap, self.edge_cap, self.tail_cap)

    def arm_pull_rider(self, parts=1, skip=0):
        """the same copy as passengers of this thread's layer-product launches (csrc/ingest_rider.h): an equal share in each of ``parts``
        carrier launches, after ``skip`` carriers that go without"""
        nat.call_nostream("ingest_arm_pull_rider_parts", self.host.data_ptr(), self.mirror, self.B, self.nmax, self.row_cap, self.edge_cap,
                          self.tail_cap, int(parts), int(skip))

    def arm_expand_rider(self):
        """the expansion (and the echo) as passengers of this thread's next packed-head launch (csrc/ingest_rider.h)"""
        nat.call_nostream("ingest_arm_expand_rider", *self._expand_args())

    def mark_consumed(self, stream=None):
        """a replay of the captured step (whose first launches are this slot's pull) has been enqueued"""
        self._replayed = True


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
    """Training loop in which every step consumes a NEW mini-batch drawn from ``ds``: native workers collate the batches ahead
    into the slots' pinned staging buffers, the enqueueing thread replays the slot's hipGraph.

    ride (default 2; env TSGNN_INGEST_RIDE=0 turns it off): the graph of position p steps on the batch that is already expanded
    in p's arrays and brings the NEXT position's batch in as passengers of its own launches (csrc/ingest_rider.h): the flat copy of
    that staging buffer over PCIe rides in the first hidden layer's product launch (``CapacityBatch.arm_pull_rider``: the ~10 us
    round trip is hidden inside the step), its expansion in the head's forward launch (``arm_expand_rider``: a few workgroups
    that leave most of the chip idle).  ride = 1: only the pull rides, the expansion is the step's first launch.  The first batch
    of a run comes in by launches of its own.  Without it (or with a custom ``make_loss``) the first two
    launches of the slot's graph pull and expand its own batch (``CapacityBatch.pull``)."""

    def __init__(self, model, trainer, ds, batch, nmax, device, schedule, depth=3, make_loss=None, ride=None):
        import os
        from .data_parallel import GraphedStep
        self.ds, self.B, self.schedule = ds, int(batch), schedule
        sizes = ds.sizes
        deg_per_graph = ds.rowptr[ds.graph_ptr[1:]] - ds.rowptr[ds.graph_ptr[:-1]]
        rows_max = max(int(sizes[np.asarray(ids)].sum()) for ids in schedule)
        edges_max = max(int(deg_per_graph[np.asarray(ids)].sum()) for ids in schedule)
        self.row_cap = (rows_max + 31) // 32 * 32
        self.edge_cap = (edges_max + 1023) // 1024 * 1024
        ghost = min(int(nmax), int(sizes.max()) + 1)
        self.compute = torch.cuda.Stream()
        self.slots = [CapacityBatch(batch, nmax, self.row_cap, self.edge_cap, ds.num_node_labels, device, ghost_slots=ghost)
                      for _ in range(depth)]
        if ride is None:
            ride = int(os.environ.get("TSGNN_INGEST_RIDE", "2"))
        self.ride = bool(ride) and make_loss is None and depth >= 2
        # 2 (default): the next batch's EXPANSION rides too (in the head's forward launch); 1: it is the step's first launch
        self.ride_expand = self.ride and int(ride) >= 2
        # Where the pull rides.  A read of pinned host memory by the GPU is ~10 us of latency + ~25 GB/s here: the ~340 KB staging buffer
        # of a DD batch takes longer than any one launch of the step, and a rider that outlives its carrier lengthens the step.  The
        # carriers are the conv layers' forward products, the later ones the longer: the copy is dealt over the LAST TWO of them in equal
        # shares (3 layers: the first carrier goes without, half in each of the other two).  Measured at 255 panels (scripts/dev/
        # ingest_rider_sweep.sh; the slot's step alone, ms): whole copy in layer 1's launch 0.1407, in layer 2's 0.1391, in layer 3's 0.1399,
        # halves in 1 + 2 0.1403, halves in 2 + 3 0.1368.  TSGNN_INGEST_PULL_PARTS / _SKIP override.
        L = int(getattr(model, "num_layers", 0) or 0)
        self.pull_parts = int(os.environ.get("TSGNN_INGEST_PULL_PARTS", "2" if L >= 3 else "1"))
        self.pull_skip = int(os.environ.get("TSGNN_INGEST_PULL_SKIP", str(max(0, L - 2)) if L >= 2 else "0"))
        self.steps = []
        if make_loss is None and not self.ride:
            def make_loss(s):
                def loss():
                    s.pull()                                  # first launches of the step: bring the staged batch in
                    return model.loss(model(s.x, s.g)[1], s.label)
                return loss
        for s in self.slots:
            s.collate(ds, schedule[0])
        if self.ride:
            with torch.cuda.stream(self.compute):
                for s in self.slots:
                    s.pull_only(); s.expand()                 # every position's arrays hold a batch before the first step

            def make_loss_ride(s, nxt):
                def loss():
                    if not self.ride_expand:
                        s.expand()                            # this position's batch: pulled by the previous step's passengers
                    nxt.arm_pull_rider(self.pull_parts, self.pull_skip)   # the next position's staging buffer rides in the layer products,
                    if self.ride_expand:
                        nxt.arm_expand_rider()                # its expansion in the head's forward launch
                    try:
                        out = model.loss(model(s.x, s.g)[1], s.label)
                    except BaseException:
                        nat.call_nostream("ingest_disarm_riders")     # nobody carried them: they must not ride an unrelated launch later
                        raise
                    nat.call("ingest_flush_pull_rider")       # (a model without such a launch: the riders as launches of their own)
                    return out
                return loss
        # capacity-padded batches keep plain 32-row panels: their rows beyond one panel per CU are mostly padding (csrc/rowgemm_body.h,
        # panel_split; the decision is taken when the launches are captured)
        nat.call_nostream("panel_split_hint", 0)
        try:
            for p, s in enumerate(self.slots):
                fn = make_loss_ride(s, self.slots[(p + 1) % depth]) if self.ride else make_loss(s)
                self.steps.append(GraphedStep(trainer, fn, warmup=2, stream=self.compute))
        finally:
            nat.call_nostream("panel_split_hint", 1)

    def run(self, schedule=None, workers=2, ticks=None):
        """enqueue one step per entry of the schedule; returns after the last step is enqueued (caller synchronises).
        workers: native collate threads running ahead (0: collate inline on the enqueueing thread).
        ticks: a dict that receives the host seconds spent waiting for collates / replaying / submitting (measurement)."""
        import time
        sched = self.schedule if schedule is None else schedule
        depth = len(self.slots)
        pool = self._get_pool(workers) if workers else None
        T = ticks if ticks is not None else {}

        def tick(name, t0):
            t1 = time.perf_counter()
            T[name] = T.get(name, 0.0) + (t1 - t0)
            return t1

        def stage(k):                                         # batch k is in its slot's staging buffer when this returns
            s = self.slots[k % depth]
            if pool is not None:
                s.collate_wait()
            else:
                s.collate(self.ds, sched[k])

        if self.ride:
            # the last replay of the previous run carried passengers for a batch that never came: they re-pull (and re-echo) whatever
            # its staging buffer holds.  Let them finish before this run's batches go into the staging buffers, or an echo of
            # theirs would be taken for the pull that feeds this run's step.
            self.compute.synchronize()
            mp.check_device_errors()                          # the host has synchronised anyway: did a step of the last run fail?
        if pool is not None:
            for k in range(min(depth, len(sched))):
                self.slots[k].collate_async(pool, self.ds, sched[k])
        if self.ride and len(sched):
            stage(0)
            with torch.cuda.stream(self.compute):
                self.slots[0].pull_only()                     # nobody rode for the first batch of the run
                if self.ride_expand:
                    self.slots[0].expand()
        for k, ids in enumerate(sched):
            s, gs = self.slots[k % depth], self.steps[k % depth]
            t = time.perf_counter()
            if self.ride:
                if k + 1 < len(sched):
                    stage(k + 1)                              # this step's passengers pull batch k + 1
            else:
                stage(k)
            t = tick("collate/wait", t)
            gs.step()                                         # [pull +] expand + forward + backward + optimiser, one replay
            t = tick("replay", t)
            (self.slots[(k + 1) % depth] if self.ride_expand else s).mark_consumed(self.compute)   # whose echo this replay emits
            if pool is not None and k + depth < len(sched):   # the slot's next batch (the worker waits for the echo first)
                s.collate_async(pool, self.ds, sched[k + depth])
            tick("submit", t)
        return len(sched)

    def _get_pool(self, workers):
        if getattr(self, "_cpool", None) is None or self._cpool_n != workers:
            if getattr(self, "_cpool", None) is not None:
                self._cpool.close()
            self._cpool, self._cpool_n = CollatePool(workers), workers
        return self._cpool
