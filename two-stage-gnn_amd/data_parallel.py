"""Data-parallel training step: one process per GPU, independent mini-batches of graphs per rank, ONE
RCCL all-reduce per step on a single flat fp32 gradient buffer over xGMI, then a fused clip+Adam.

The reference is single-process (train.py:509 pins one device, no torch.distributed anywhere); this is
the new capability BASELINE.json's north_star asks for.  Graphs never share edges, so
there is no feature exchange between ranks: the gradient all-reduce is the only collective.  The model
is tiny (SAGE-3L h=128 on DD: ~61 k parameters = 0.25 MB), so the collective is latency-bound: one
bucket, one call.  Slot batch-norm statistics stay local to each rank's batch (= the reference run on
each shard).
"""
import os

import torch
import torch.distributed as dist

from . import _native as nat
from . import message_passing as mp


class FlatTrainer:
    """Owns flat parameter / gradient / Adam-moment buffers; model parameters become views of the flat
    parameter buffer, so the HIP optimiser kernel updates the model in place."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip=2.0, group=None, direct_grads=True,
                 defer_loss=False, buckets=None):
        self.model = model
        # buckets = 2: the gradients that are final EARLY in the backward (the fused stack reports its head's gradients right
        # after the first backward launch: 49 k of the 94 k floats of the SAGE-3L model) are all-reduced on a side stream while
        # the rest of the backward runs; the remainder follows in all_reduce().  Takes effect where the collective can overlap:
        # eager steps and the one-graph step (parallel branch of the hipGraph); the two-graph step keeps one bucket.
        self.buckets = int(os.environ.get("TSGNN_AR_BUCKETS", 1)) if buckets is None else int(buckets)
        # defer_loss: a cross-entropy computed between zero_grad() and backward() on the fused stack's logits launches nothing;
        # the head's backward kernel rebuilds its gradient and fills in the loss value (one launch less per step).  The loss
        # tensor then holds its value only after backward() — what step() / GraphedStep return.
        self.defer_loss = bool(defer_loss)
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        # every parameter starts on a 16-byte boundary of the flat buffer (the float4 / MFMA kernels read weights with 16-byte
        # loads; a 1-element bias would otherwise misalign everything behind it).  The few padding floats stay zero in the
        # parameter, gradient and moment buffers, so the norm, the all-reduce and Adam are unaffected by them.
        self.views = []
        self._pads = []
        off = 0
        for p in self.params:
            pad = (-off) % 4
            self._pads.append(pad)
            off += pad
            self.views.append((off, p.numel()))
            off += p.numel()
        self.numel = off
        self.flat_param = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        # the gradient bucket with one spare slot behind it: at N > 1 every rank puts ITS device error word there before the collective,
        # so the SUM all-reduce hands every rank the same verdict and all of them skip (and later raise) together — a rank-local word
        # let the healthy ranks apply an update the failed rank skipped
        self._grad_store = torch.zeros(self.numel + 4, dtype=torch.float32, device=dev)
        self.flat_grad = self._grad_store[:self.numel]
        self._err_slot = self._grad_store[self.numel:self.numel + 1]
        for p, (o, n) in zip(self.params, self.views):
            self.flat_param[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + n].view_as(p.data)
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip
        self.on_gpu = dev.type == "cuda"
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)      # step, grad norm, applied scale, 1 = an update was skipped
        # the device's error word: the optimiser kernels skip their update while a kernel's bounded barrier has reported a time-out
        self.poison = mp.device_error_word(dev) if dev.type == "cuda" else None
        self.ws = torch.zeros(264, dtype=torch.float32, device=dev)       # norm partials + the optimiser kernel's sign-off counter
        self._zeros = [torch.zeros_like(p).reshape(-1) for p in self.params]     # stand-ins for parameters without a gradient
        self._pad_zeros = [torch.zeros(k, dtype=torch.float32, device=dev) if k else None for k in self._pads]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if self.world > 1:
            # replicas must start from the same bits whatever each rank's RNG did before constructing the model (DDP does the same)
            dist.broadcast(self.flat_param, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        # direct_grads: between zero_grad() and gather_grads() the fused backward nodes write parameter gradients straight
        # into their slice of flat_grad (message_passing.GradSink).  A slice is handed out once per step: a parameter that
        # feeds a second fused node (the reference's tripletnet calls one encoder three times) gets its further contributions
        # through autograd and gather_grads adds them (GradSink.take).
        self.sink = None
        self._dirty = set()                 # parameter indices whose slice of flat_grad may be non-zero (it starts all zero)
        if direct_grads and self.on_gpu:
            self.sink = mp.GradSink()
            for p, (o, n) in zip(self.params, self.views):
                self.sink.views[p.data_ptr()] = self.flat_grad[o:o + n].view_as(p.data)
            self.sink.norm_parts = torch.zeros(4096, dtype=torch.float32, device=dev)
            self.sink.step_state = self.state
        self.always_reduce = False          # issue the collective even in a one-rank group (single-GPU rehearsal of the N > 1 path)
        self._early = None                  # (start, end) of the early bucket, learnt from the first backward that reports one
        self._early_issued = False
        self._allow_early = True            # cleared by GraphedStep while it captures the two-graph step
        self._side = torch.cuda.Stream() if (self.on_gpu and self.buckets > 1) else None

    # ------------------------------------------------------------------ early bucket (overlapped all-reduce)
    def _reducing(self):
        return (self.world > 1 or self.always_reduce) and dist.is_available() and dist.is_initialized()

    def _range_of(self, params):
        """(start, end) of the flat buffer covered by `params` if they form one contiguous run of it (padding included)"""
        ptrs = {q.data_ptr() for q in params}
        idx = sorted(i for i, p in enumerate(self.params) if p.data_ptr() in ptrs)
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        start = self.views[idx[0]][0] - self._pads[idx[0]]
        last = idx[-1]
        end = self.numel if last == len(self.params) - 1 else self.views[last + 1][0] - self._pads[last + 1]
        return start, end

    def _on_ready(self, params):
        """called by a fused backward node right after the launch that finalised the gradients of `params` (on the compute
        stream, possibly under hipGraph capture): fork, all-reduce their range on the side stream"""
        if self._early_issued or not self._allow_early or self._side is None or not self._reducing():
            return
        rng = self._range_of([p for p in params if p is not None])
        if rng is None or rng[1] - rng[0] <= 0:
            return
        if self._early is None:
            self._early = rng
        elif rng != self._early:
            return
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            dist.all_reduce(self.flat_grad[rng[0]:rng[1]], op=dist.ReduceOp.SUM, group=self.group)
        self._early_issued = True

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self._norm_ready = False
        self._early_issued = False
        mp.CE_DEFER = self.defer_loss and self.on_gpu
        if self.sink is not None:
            self.sink.ready_cb = self._on_ready if (self._side is not None and self._reducing()) else None
            self.sink.written.clear()
            self.sink.reset_norm()
            self.sink.norm_enabled = self.world == 1 and not self.always_reduce     # local shares only describe a local gradient
            mp.GRAD_SINK = self.sink

    def backward(self, loss):
        """loss.backward() seeded with the cached unit scalar (no fill launch, no multiply by 1 in the loss node)."""
        loss.backward(gradient=mp.unit_seed(loss.device) if loss.is_cuda and loss.dim() == 0 else None)

    def gather_grads(self):
        """per-parameter gradients -> the flat bucket: nothing to do for slices the backward nodes wrote in place; one
        concatenation kernel when none were; a copy / clear per remaining parameter otherwise."""
        written = ()
        mp.CE_DEFER = False
        if self.sink is not None:
            mp.GRAD_SINK = None
            written = self.sink.written
            # barrier-free optimiser: every gradient of this step was written in place by a producer that also left its
            # share of |grad|^2 and the step counter was advanced; nothing arrived through autograd; single GPU
            self._norm_ready = (bool(written) and self.sink.stepped and written == self.sink.normed and not self.sink.reused
                                and all(p.grad is None for p in self.params) and self.world == 1 and not self.always_reduce)
        if not written:
            parts = []
            for p, z, pz in zip(self.params, self._zeros, self._pad_zeros):
                if pz is not None:
                    parts.append(pz)
                parts.append(p.grad.reshape(-1) if p.grad is not None else z)
            torch.cat(parts, out=self.flat_grad)
            self._dirty = {i for i, p in enumerate(self.params) if p.grad is not None}
            return self.flat_grad
        # the remaining parameters in THREE multi-tensor launches (copies, sums, clears) instead of one launch per parameter: a
        # model whose fused nodes place some gradients and whose other ops leave ~40 `.grad` tensors (DiffPool) paid ~3.5 us each
        cp_dst, cp_src, add_dst, add_src, zero_dst = [], [], [], [], []
        for i, (p, (o, n)) in enumerate(zip(self.params, self.views)):
            direct = p.data_ptr() in written
            if p.grad is not None:
                if direct and self._early_issued and self._early[0] <= o < self._early[1]:
                    # this slice is already being summed over the ranks: the late contribution is summed separately (the
                    # all-reduce is linear) and joins it after the side stream
                    extra = p.grad.reshape(-1).contiguous()
                    dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=self.group)
                    torch.cuda.current_stream().wait_stream(self._side)
                    self.flat_grad[o:o + n].add_(extra)
                elif direct:
                    add_dst.append(self.flat_grad[o:o + n]); add_src.append(p.grad.reshape(-1))   # used by a sink node AND an ordinary op
                else:
                    cp_dst.append(self.flat_grad[o:o + n]); cp_src.append(p.grad.reshape(-1))
                self._dirty.add(i)
            elif direct:
                self._dirty.add(i)
            elif i in self._dirty:                                          # no gradient this step, stale values from an earlier one
                zero_dst.append(self.flat_grad[o:o + n])
                self._dirty.discard(i)
        if cp_dst:
            torch._foreach_copy_(cp_dst, cp_src)
        if add_dst:
            torch._foreach_add_(add_dst, add_src)
        if zero_dst:
            torch._foreach_zero_(zero_dst)
        return self.flat_grad

    def all_reduce(self):
        """SUM over ranks on the flat bucket (+ the error-word slot behind it); the 1/world factor is applied inside the optimiser
        kernel."""
        if self.world > 1 or self.always_reduce:
            if self.poison is not None:
                self._err_slot.copy_(self.poison)                           # this rank's error word rides in the collective
            if self._early_issued:
                a, b = self._early
                for s_, e_ in ((0, a), (b, self.numel + 4)):
                    if e_ > s_:
                        dist.all_reduce(self._grad_store[s_:e_], op=dist.ReduceOp.SUM, group=self.group)
                torch.cuda.current_stream().wait_stream(self._side)        # join: the optimiser needs both buckets
                return
            dist.all_reduce(self._grad_store, op=dist.ReduceOp.SUM, group=self.group)

    def _poison_word(self):
        """what the optimiser kernels read as `poison`: the device's error word, or — after an all-reduce — the sum of every rank's"""
        return self._err_slot if ((self.world > 1 or self.always_reduce) and self.poison is not None) else self.poison

    def apply(self):
        scale = 1.0 / self.world
        if not self.on_gpu:
            raise RuntimeError("FlatTrainer.apply runs the HIP optimiser kernel; parameters must live on the GPU "
                               "(no CPU fallback)")
        if getattr(self, "_norm_ready", False):
            nat.call("adam_from_partials_f32", self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.numel,
                     float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), float(self.clip),
                     self.state, self.sink.norm_parts, int(self.sink.norm_used), self._poison_word())
            return
        if self.sink is not None and self.sink.stepped:
            self.state[0] -= 1.0                        # the gradient reduction advanced the counter for the barrier-free path
        nat.call("clip_adam_step_f32", self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.numel,
                 float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
                 float(self.clip), float(scale), self.state, self.ws, self._poison_word())

    def check(self):
        """synchronise and raise if a kernel of a step since the last check reported invalid results (a bounded device-wide
        barrier that timed out).  The optimiser has skipped every update since then (state[3] = 1), so parameters and moments
        are those of the last good step."""
        torch.cuda.synchronize()
        skipped = self.on_gpu and float(self.state[3]) != 0.0
        try:
            mp.check_device_errors()
            if skipped and (self.world > 1 or self.always_reduce):
                # this rank's own kernels are fine, but the optimiser skipped: ANOTHER rank's error word arrived with the gradients
                raise RuntimeError("a bounded device-wide barrier timed out on another rank: every rank skipped the optimiser update "
                                   "since (the replicas still hold identical parameters)")
        finally:
            if skipped:
                self.state[3] = 0.0
                self._err_slot.zero_()
                torch.cuda.synchronize()

    def step(self, loss_fn):
        """loss_fn() -> scalar loss.  fwd + bwd + all-reduce + clip + Adam."""
        self.zero_grad()
        loss = loss_fn()
        self.backward(loss)
        self.gather_grads()
        self.all_reduce()
        self.apply()
        return loss


class _capture:
    """torch.cuda.graph(...) with the cyclic garbage collector held off for the duration of the capture: a collection that happens to run
    between two captured launches finalises whatever unreachable device objects it finds (graphs, events, pinned buffers of a discarded
    trainer or pipeline) — runtime calls a capturing stream does not admit, which abort the process from inside a destructor (seen
    when several pipelines were built one after the other in one process).  The garbage is collected right before the capture instead."""

    def __init__(self, graph, **kw):
        self._ctx = torch.cuda.graph(graph, **kw)

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            return self._ctx.__enter__()
        except BaseException:
            if self._was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self._ctx.__exit__(*exc)
        finally:
            if self._was:
                gc.enable()


class GraphedStep:
    """One optimiser step on FIXED device buffers, replayed from hipGraphs (the b = 32 step is launch-bound: 15 launches).

        gs = GraphedStep(trainer, lambda: model.loss(model(x, g)[1], label))   # captures on its own stream
        for _ in range(steps): gs.step()                                        # refill x / label in place between steps

    Single GPU: forward + backward + gradient bucket + optimiser are ONE graph.  With a process group (world > 1 or
    trainer.always_reduce) the RCCL all-reduce cannot live inside the capture here, so the step is two graphs
    (forward/backward/bucket, optimiser) with ``trainer.all_reduce()`` issued between them on the same stream; the captures
    use ``capture_error_mode="thread_local"`` because the RCCL watchdog thread touches the HIP runtime concurrently.
    ``use_graph=False`` runs the same sequence eagerly (debugging)."""

    def __init__(self, trainer, loss_fn, warmup=3, use_graph=True, stream=None, steps_per_replay=1):
        self.trainer, self.loss_fn, self.use_graph = trainer, loss_fn, use_graph
        self.multi = trainer.world > 1 or trainer.always_reduce
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.loss = None
        self._fb = self._opt = None
        self._fbk, self.steps_per_replay = None, 1
        self.one_graph = False
        with torch.cuda.stream(self.stream):
            # allocator / lazy-init warm-up on the capture stream.  The warm-up steps are real optimiser steps on whatever the
            # input buffers hold, so the trainer's state is put back afterwards: constructing a GraphedStep leaves parameters,
            # Adam moments and the step counter exactly as it found them.
            snap = [t.clone() for t in (trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state)] if warmup else None
            for _ in range(warmup):
                self._fwd_bwd(); trainer.all_reduce(); trainer.apply()
            if snap is not None:
                for t, s_ in zip((trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state), snap):
                    t.copy_(s_)
            torch.cuda.synchronize()
            mp.check_device_errors()
            if not use_graph:
                return
            if self.multi and dist.is_initialized():
                dist.barrier(group=trainer.group)                  # no collective in flight while capturing
                torch.cuda.synchronize()
                # ... and none on the process group's watchdog list either: its thread polls the events of the collectives it still
                # holds every ~100 ms, and a poll that lands inside the capture was seen (once in some tens of runs, one-rank RCCL) to
                # fail with hipErrorCapturedEvent, invalidate the capture and take the process down with the watchdog's exception.
                # The completed collectives leave the list at its next pass: give it two.
                if dist.get_backend(trainer.group) == "nccl":
                    import time
                    time.sleep(0.25)
            mode = {"capture_error_mode": "thread_local"} if self.multi else {}
            # N > 1: TSGNN_GRAPH_ALLREDUCE = 1: the collective is captured too, the step is ONE graph (no graph boundary either
            # side of the all-reduce: 0.156 vs 0.1715 ms in the one-rank rehearsal of round 1); 0: two graphs around an eagerly
            # issued all-reduce; auto (default): capture the one-graph step, CHECK it — one replay from a snapshot must leave
            # the parameters the eager sequence leaves, on every rank (agreement by an eager MIN all-reduce) — and fall back
            # to two graphs otherwise.
            want = os.environ.get("TSGNN_GRAPH_ALLREDUCE", "auto") if self.multi else "0"
            self.mode_note = ""
            if want == "auto" and self.multi and dist.is_initialized() and dist.get_backend(trainer.group) != "nccl" \
                    and os.environ.get("TSGNN_ONE_GRAPH_ANY_BACKEND") != "1":
                want = "0"                  # gloo moves the data through the host: its collectives cannot be captured
            if self.multi and want != "0":
                self._snap = [t.clone() for t in self._bufs()]
                graph = self._capture_one(mode)                  # None: the capture raised
        # (the rest runs outside the block above: its stream may be one a failed capture has just poisoned)
        if self.multi and want != "0":
            ok = self._try_one_graph(graph, verify=(want != "1"))
            graph = None
            if not ok:
                self._fb = None
                if want == "1":
                    raise RuntimeError("TSGNN_GRAPH_ALLREDUCE=1: the all-reduce could not be captured into the step's hipGraph "
                                       "on every rank; use TSGNN_GRAPH_ALLREDUCE=auto (checked, falls back) or 0 (two graphs)")
            self.one_graph = ok
        if self._fb is None:
            with torch.cuda.stream(self.stream):                 # (self.stream is a fresh one if a capture was invalidated)
                self._capture_two(mode)
        self._snap = None
        self._loss1 = self._lossk = self.loss                    # the loss scalar the single-step graph(s) write
        if int(steps_per_replay) > 1 and not self.multi:
            # run(): k CONSECUTIVE optimiser steps on the same resident buffers as ONE graph launch — between two graph launches the
            # device idles for the launch's own latency (8.7 us behind a 0.131 ms step: scripts/replay_trace.py), once per replay
            # whatever the graph holds.  N = 1 only: at N > 1 a step is paced by its all-reduce, and run() replays step by step.
            self.steps_per_replay = int(steps_per_replay)
            with torch.cuda.stream(self.stream):
                snap = [t.clone() for t in (trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state)]
                self._fbk = torch.cuda.CUDAGraph()
                with _capture(self._fbk, stream=self.stream):
                    for _ in range(self.steps_per_replay):
                        self._fwd_bwd()
                        trainer.apply()
                for t, s_ in zip((trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state), snap):
                    t.copy_(s_)                                  # (capturing runs nothing, but keep the contract explicit)
                torch.cuda.synchronize()
            self._lossk, self.loss = self.loss, self._loss1      # (the k-step graph's LAST step writes its own scalar)

    def _bufs(self):
        tr = self.trainer
        return tr.flat_param, tr.exp_avg, tr.exp_avg_sq, tr.state

    def _agree(self, ok):
        """every rank takes the same decision about the one-graph step (eager MIN all-reduce of the local verdicts)"""
        tr = self.trainer
        if dist.is_initialized():
            with torch.cuda.stream(self.stream):                 # (the caller's current stream may be one a capture poisoned)
                flag = torch.tensor([1.0 if ok else 0.0], device=tr.flat_param.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=tr.group)
                ok = bool(flag.item() > 0.5)
                torch.cuda.synchronize()
        return ok

    def _capture_two(self, mode):
        trainer = self.trainer
        if self.multi and dist.is_initialized() and dist.get_backend(trainer.group) == "nccl":
            # (the one-graph attempt before this issued collectives: let the watchdog drop them before the capture, as in __init__)
            torch.cuda.synchronize()
            import time
            time.sleep(0.25)
        trainer._allow_early = not self.multi                # a fork inside the first of two graphs could not be joined
        self._fb = torch.cuda.CUDAGraph()
        with _capture(self._fb, stream=self.stream, **mode):
            self._fwd_bwd()
            if not self.multi:
                trainer.apply()
        if self.multi:
            self._opt = torch.cuda.CUDAGraph()
            with _capture(self._opt, stream=self.stream, **mode):
                trainer.apply()

    def _abandon_stream(self):
        """A capture that was invalidated half-way leaves its stream unusable on this stack (every later call on it reports
        hipErrorStreamCaptureInvalidated), and a side stream that was forked into it is in the same state: carry on with fresh
        ones, swallow the runtime's sticky "last error", put the trainer's state back."""
        tr = self.trainer
        self.stream = torch.cuda.Stream()
        if tr._side is not None:
            tr._side = torch.cuda.Stream()
        tr._early_issued = False
        with torch.cuda.stream(self.stream):
            for _ in range(4):                                   # the sticky error is reported once more
                try:
                    torch.cuda.synchronize()
                    break
                except Exception:                                # noqa: BLE001
                    pass
            for t, s_ in zip(self._bufs(), self._snap):
                t.copy_(s_)
            torch.cuda.synchronize()

    def _capture_one(self, mode):
        tr = self.trainer
        graph = torch.cuda.CUDAGraph()
        try:
            tr._allow_early = True
            with _capture(graph, stream=self.stream, **mode):
                self._fwd_bwd()
                if os.environ.get("TSGNN_TEST_BREAK_CAPTURE") in ("all", str(dist.get_rank() if dist.is_initialized() else 0)):
                    torch.cuda.current_stream().synchronize()   # test hook: an operation a capture does not admit — the
                tr.all_reduce()                                 # capture is invalidated and raises (rehearsal of the fall-back)
                tr.apply()
        except Exception as e:                                  # noqa: BLE001 - any capture failure selects the two-graph step
            print("GraphedStep: one-graph capture failed (%s: %s)" % (type(e).__name__, str(e).splitlines()[0]), flush=True)
            return None
        return graph

    def _try_one_graph(self, graph, verify):
        """`graph`: forward + backward + bucket + all-reduce + optimiser captured as ONE graph (_capture_one; None if the capture
        raised on this rank); verify=True: replay it once from a
        snapshot and compare with the eager sequence from the same snapshot.  Leaves the trainer's state as it found it.
        -> installed in self._fb.

        Every rank issues the SAME sequence of collectives whatever happened locally (a rank whose capture raised
        used to skip the verification replay — a captured SUM all-reduce of the bucket on its peers — and go straight to the
        1-element MIN, i.e. collectives mismatched in order and size):
          capture attempt (no communication: the collective is recorded, not run) -> [a failed rank recovers its stream] ->
          MIN all-reduce "did every rank capture?" -> only if all did: replay + eager step (both all-reduce the bucket) ->
          MIN all-reduce "did every rank's check pass?"."""
        tr = self.trainer
        bufs, snap = self._bufs(), self._snap
        captured = graph is not None
        if not captured:
            self._abandon_stream()
        all_captured = self._agree(captured)
        if not all_captured:
            self.mode_note = " (one-graph step failed its capture on %s: fell back)" % ("this rank" if not captured else "another rank")
            return False
        ok = True
        with torch.cuda.stream(self.stream):
            if verify:
                graph.replay()
                torch.cuda.synchronize()
                got = tr.flat_param.clone()
                for t, s_ in zip(bufs, snap):
                    t.copy_(s_)
                self._fwd_bwd(); tr.all_reduce(); tr.apply()
                torch.cuda.synchronize()
                scale = float(tr.flat_param.abs().max()) + 1e-30
                ok = bool(torch.isfinite(got).all()) and float((got - tr.flat_param).abs().max()) <= 1e-5 * scale
            for t, s_ in zip(bufs, snap):
                t.copy_(s_)
            torch.cuda.synchronize()
        if verify:
            ok = self._agree(ok)
            if not ok:
                self.mode_note = " (one-graph step failed its check: fell back)"
        if ok:
            self._fb = graph
        return ok

    def describe(self):
        if not self.use_graph:
            return "eager"
        if not self.multi:
            k = ("; run(): %d consecutive steps per graph launch" % self.steps_per_replay) if self._fbk is not None else ""
            return "one graph: forward + backward + bucket + optimiser" + k
        bk = ", early bucket [%d, %d) all-reduced on a side branch" % self.trainer._early if self.trainer._early else ""
        if self.one_graph:
            return "one graph incl. the captured all-reduce" + bk
        return "two graphs around the eagerly issued all-reduce" + getattr(self, "mode_note", "")

    def _fwd_bwd(self):
        tr = self.trainer
        tr.zero_grad()
        self.loss = self.loss_fn()
        tr.backward(self.loss)
        tr.gather_grads()

    def synchronize(self):
        """wait for the enqueued steps and raise if one of them reported invalid results (FlatTrainer.check): call it wherever
        the loop synchronises anyway — reading the loss, logging, the end of an epoch"""
        self.stream.synchronize()
        self.trainer.check()

    def loss_value(self):
        """the last step's loss as a float (synchronises; checked)"""
        self.synchronize()
        return float(self.loss.detach())

    def run(self, n):
        """enqueue n consecutive steps on self.stream: graphs of `steps_per_replay` steps while they fit, single steps for the rest
        (the same n optimiser steps as n calls of step(), bit for bit — tests/test_data_parallel.py)"""
        n = int(n)
        self._since_check = getattr(self, "_since_check", 0) + n
        if self._fbk is not None:
            with torch.cuda.stream(self.stream):
                while n >= self.steps_per_replay:
                    self._fbk.replay()
                    n -= self.steps_per_replay
                    self.loss = self._lossk
        for _ in range(n):
            self.step()
        if self._since_check >= self.CHECK_EVERY:
            # a loop that only ever calls run() / step() would never raise: look at the error word at a coarse interval (one
            # synchronisation per CHECK_EVERY enqueued steps)
            self._since_check = 0
            self.synchronize()
        return self.loss

    CHECK_EVERY = 4096

    def step(self):
        """enqueue one step on self.stream (returns immediately; self.loss is the device scalar of the last step; read it through
        loss_value(), which also checks the device's error word)"""
        with torch.cuda.stream(self.stream):
            if not self.use_graph:
                self._fwd_bwd(); self.trainer.all_reduce(); self.trainer.apply()
            elif not self.multi or self.one_graph:
                self._fb.replay()
                self.loss = self._loss1
            else:
                self._fb.replay(); self.trainer.all_reduce(); self._opt.replay()
                self.loss = self._loss1
        return self.loss
