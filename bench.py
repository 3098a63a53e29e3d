#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): graphs/sec, forward+backward(+all-reduce+clip+Adam) of the
3-layer h=128 GraphSage-style encoder (GcnEncoderGraph, `--method=base`) on DD-shaped synthetic batches
of 32 graphs per GPU, on 1/2/4/8 MI355X, plus the roofline of EVERY kernel the timed step launches, the
aggregation kernel's batch-size sweep and the reference's dense formulation timed on the host CPU.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (WORLD_SIZE unset: starts the N ranks itself as child processes)

One JSON line on stdout (rank 0).  A "step" = one optimiser step on one batch per rank: forward, CE loss,
backward, gradient bucket (+RCCL all-reduce when N>1), clip_grad_norm(2.0), Adam — exactly the body of the
reference's loop (train.py:110-131) minus the host->device copies (inputs are resident in HBM).

`roofline` describes the step that was timed:
  step_kernels   one row per launch group of the step, in launch order: the device kernel the library dispatched (reported by
                 the library, tsgnn_last_kernel), launches per step, average duration of ONE launch measured with HIP events
                 around a hipGraph burst of that very launch (same arguments, same buffers the step used: layer 0 runs at its
                 K = 92, not at a stand-in shape), algorithmic flops and bytes, and the fraction of the fp32-MFMA peak and of
                 the HBM peak that follows.
  (top level)    the kernel with the largest per-step total, with the fields the contract names.
  aggregation_in_fused   SURVEY §8(d)'s aggregation-only bytes / the duration of the fused kernel that contains the aggregation.
  sweep          the stand-alone aggregation kernel (tsgnn_ell_spmm_f32 / tsgnn_csr_spmm_f32, F = hidden) at
                 32 / 256 / 2,048 / 16,384 DD graphs (north_star: >= 40 % of the HBM roofline on DD-sized batched graphs).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36
MFMA_F32_PEAK_TF = 157.3       # dense fp32 MFMA, /opt/skills/guides/MI355X_MICROARCH.md:42
PROFILE_DIR = os.path.join("profiles", "r04")
HBM_COPY_CEILING_GBS = 6290.0    # measured float4 copy, /opt/skills/guides/MI355X_MICROARCH.md:36


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="graphs per GPU (BASELINE: 32)")
    ap.add_argument("--shape", default="DD")
    ap.add_argument("--nmax", type=int, default=1000, help="padded size of the reference (--max_nodes default)")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--steps-per-graph", type=int, default=1,
                    help="N = 1: consecutive optimiser steps captured into one hipGraph launch (GraphedStep.run).  Default 1 = one step per "
                         "launch, what a training loop with a new batch per step can use (`value`); the 4-steps-per-launch time of the same "
                         "resident batch is reported beside it as `ms_per_step_four_steps_per_graph`")
    ap.add_argument("--settle-steps", type=int, default=128,
                    help="untimed replays of the captured step right after capture, BEFORE the --warmup steps (the same count on every "
                         "rank): the first ~80 replays of a freshly captured step run 3-4 %% slower than every later one (0.131 -> "
                         "0.127 ms over ~10 ms; a matmul burst of the same length does not remove it, DESIGN.md §5), so a 20-step timed "
                         "region right after capture measures the settling, not the step.  0 disables; reported as `config.settle`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="0 = auto-size the sample to ~15 s")
    ap.add_argument("--no-sweep", action="store_true", help="skip the aggregation-kernel batch-size sweep")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel table of the step (rocprof runs of the bare step)")
    ap.add_argument("--sweep-sizes", default="32,256,2048,16384")
    ap.add_argument("--no-seeds", action="store_true", help="skip `value_over_seeds` (the same step on the batches of seeds 0-7)")
    ap.add_argument("--seeds", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--seed-base", type=int, default=0, help="rank r draws the batch of seed seed-base + r (default 0: the contract's batches)")
    ap.add_argument("--no-pyg", action="store_true", help="skip `pyg_surface` (the same batch through the torch_geometric-named layers)")
    ap.add_argument("--triplet", action="store_true", help="also time the 2stg triplet step (f3: one triplet per optimiser step)")
    ap.add_argument("--ingest", action="store_true", help="also time the step fed with a NEW host batch every step (collate + "
                                                          "upload on a copy stream, double-buffered): `ingest` in the JSON line")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (nothing in this process has
    touched the GPU yet), forward rank 0's JSON line, exit with the launcher's code.  Never falls back to one GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE)
    sys.stdout.write(r.stdout.decode())
    sys.stdout.flush()
    if r.returncode != 0:
        sys.stderr.write("bench.py: the %d-rank launch failed (exit %d); no single-GPU number is reported in its place\n"
                         % (a.gpus, r.returncode))
    raise SystemExit(r.returncode)


def hip_event_ms(fn, iters, stream):
    """average ms per call of fn() measured with HIP events recorded on `stream`."""
    import torch
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def burst_ms(fn, iters, stream, repeats=5):
    """ms per launch of fn() replayed back to back from one hipGraph on `stream` (the host's ~8 us per ctypes launch does not
    pace it); the best of `repeats` replays."""
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    burst = torch.cuda.CUDAGraph()
    with torch.cuda.graph(burst, stream=stream):
        for _ in range(iters):
            fn()
    burst.replay()
    torch.cuda.synchronize()
    return min(hip_event_ms(burst.replay, 1, stream) for _ in range(repeats)) / iters


# ----------------------------------------------------------------------------------------------- the step's kernels
def _numel(t):
    return 0 if t is None else int(t.numel())


def account(entry, a, nnz):
    """(flops, algorithmic bytes, what) of one launch of a training-step entry point, from its own arguments.
    Bytes follow SURVEY §8(d): every operand once (features read once however often they are gathered), indices, row
    pointers, outputs once; flops count the MFMA products only (2mnk)."""
    f4 = 4
    if entry == "gather_rowgemm_f32":
        rows, K, N, fill = int(a[15]), int(a[16]), int(a[17]), int(a[19])
        by = f4 * rows * K + 4 * nnz + 4 * (rows + 1) + f4 * rows * N + f4 * K * N + f4 * fill * N
        by += f4 * rows * K if a[13] is not None else 0          # z (the aggregate) kept for the weight gradient
        by += f4 * (rows + fill) if a[12] is not None else 0     # rinv
        return 2.0 * rows * K * N, by, "aggregate + .W + bias + L2 normalise (layer 0), K=%d N=%d" % (K, N)
    if entry == "gather_rowgemm_st_f32":
        rows, K, N, fill = int(a[14]), int(a[15]), int(a[16]), int(a[17])
        by = f4 * rows * K + 4 * nnz + 4 * (rows + 1) + f4 * rows * N + f4 * K * N + f4 * fill * N + 16 * rows     # + one pair of 64-bit atomics a row
        by += f4 * rows * K if a[12] is not None else 0
        by += f4 * (rows + fill) if a[11] is not None else 0
        return 2.0 * rows * K * N, by, "aggregate + .W + bias + L2 normalise + slot-BN statistics (layer 0), K=%d N=%d" % (K, N)
    if entry == "sage_layer_fwd_bn_f32":
        rows, K, gs, B = int(a[14]), int(a[15]), int(a[16]), int(a[19])
        ro, st = a[23] is not None, a[29] is not None
        by = (f4 * rows * K + 4 * nnz + 4 * (rows + 1) + 2 * f4 * rows * K + f4 * K * K + f4 * gs * K + f4 * (rows + gs)       # product half (x = the previous v)
              + f4 * gs * K + 8 * B * K + 16 * int(a[20]) * 2                                                                # readout half: ghost rows, packed maxima; the integer sums
              + ((8 * B * K + 4 * rows) if ro else 0) + (16 * rows if st else 0))
        return 2.0 * rows * K * K, by, ("[slot BN of the input on the fly + aggregate + .W + bias + normalise%s || max-readout partial of BN(input)], K=N=%d"
                                        % (" + max readout of the output" if ro else (" + slot-BN statistics" if st else ""), K))
    if entry == "packed_head_fwd_z_f32":
        B, L, Fh, Fl, E, C = int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[12]), int(a[13])
        P = (L - 1) * Fh + Fl
        return 2.0 * B * (P * E + E * C), 16 * B * P + f4 * (E * P + C * E) + 2 * f4 * B * P + 8 * int(a[17]), \
            "decode of all layers' packed maxima + Linear(%d,%d) + Linear(%d,%d) + zeroing of the step's accumulators" % (P, E, E, C)
    if entry in ("head2_bwd_du_f32", "head2_bwd_du_map_f32"):
        B, P, E, C, n, F = int(a[10]), int(a[11]), int(a[12]), int(a[13]), int(a[22]), int(a[30])
        return 4.0 * B * (P * E + E * C), 2 * f4 * (E * P + C * E) + 2 * f4 * B * P + 2 * f4 * n * F + f4 * n + 8 * B * F, \
            "CE loss + head backward (dW1, db1, dW2, db2, d readout) || last layer's dU rows from the readout gradient"
    if entry in ("sage_layer_fwd_f32", "sage_layer_fwd_ro_f32"):
        rows, K, gs = int(a[14]), int(a[15]), int(a[16])
        B = int(a[18])
        ro = entry.endswith("_ro_f32") and a[22] is not None
        by = (f4 * rows * K + 4 * nnz + 4 * (rows + 1) + 2 * f4 * rows * K + f4 * K * K + f4 * gs * K + f4 * (rows + gs)   # product half
              + f4 * gs * K + 8 * B * K                                                                                   # readout half: ghost rows, packed maxima
              + ((8 * B * K + 4 * rows) if ro else 0))                                                                    # own readout: packed maxima, row -> graph
        return 2.0 * rows * K * K, by, ("[aggregate + .W + bias + normalise%s || max-readout partial of the layer's input], K=N=%d"
                                        % (" + max readout of the output" if ro else "", K))
    if entry == "sage_layer_bwd_f32":
        rows, nslab, sg = int(a[12]), int(a[13]), int(a[15])
        K = N = 128
        by = (f4 * (rows + sg) * N + f4 * rows * K + 4 * nnz + 4 * (rows + 1) + f4 * rows * K + f4 * K * N + f4 * nslab * (K + 1) * N)
        return 4.0 * rows * K * N, by, "[dW/db slabs = z^T dU || dX = (A dU) W^T], K=N=128"
    if entry == "linear_wgrad_f32":
        rows, K, N, nslab, bo = int(a[4]), int(a[5]), int(a[6]), int(a[7]), int(a[9])
        return 2.0 * rows * K * N, f4 * rows * K + f4 * (rows + bo) * N + f4 * nslab * (K + 1) * N, "dW/db slabs of layer 0, K=%d N=%d" % (K, N)
    if entry == "wgrad_reduce_multi_f32":
        by = 0
        for t in range(4):
            ns, K, N = int(a[6 * t + 1]), int(a[6 * t + 2]), int(a[6 * t + 3])
            by += f4 * (ns + 1) * (K + 1) * N
        return 0.0, by, "fixed-order slab sums of all layers' dW/db -> flat bucket, |grad|^2 shares"
    if entry == "slot_bn_fwd_f32":
        sn, n, sg, F = int(a[3]), int(a[4]), int(a[5]), int(a[8])
        return 0.0, 2 * f4 * (n + sg) * F + 8 * int(a[15]) + 8 * sn, "ReLU + per-slot BatchNorm (fresh statistics), F=%d" % F
    if entry == "slot_post_bwd_f32":
        n, sg, F = int(a[4]), int(a[5]), int(a[15])
        nread = 1 + (a[8] is not None) + (a[10] is not None)
        by = f4 * (n + sg) * F * (nread + 1) + f4 * (n + sg) + (8 * int(a[2]) * F if a[12] is not None else 0)
        return 0.0, by, "backward of readout scatter + slot BN + ReLU + L2 normalise -> dU, F=%d" % F
    if entry == "slot_post_wgrad_f32":
        n, sg, F, K, nblk = int(a[4]), int(a[5]), int(a[13]), int(a[21]), int(a[23])
        by = f4 * (n + sg) * F * 2 + f4 * n * K + f4 * (n + sg) + 8 * int(a[2]) * F + f4 * nblk * (K + 1) * F
        return 2.0 * n * K * F, by, "layer 0: backward of readout scatter + slot BN + ReLU + normalise AND the dW/db slabs (dU stays in LDS), K=%d N=%d" % (K, F)
    if entry == "readout_head_fwd_f32":
        B, L, Fh, Fl, n, sg, E, C = int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[8]), int(a[10]), int(a[18]), int(a[19])
        P = (L - 1) * Fh + Fl
        return 2.0 * B * (P * E + E * C), f4 * (n + sg) * Fl + 8 * B * (L - 1) * Fh + f4 * (E * P + C * E) + 2 * f4 * B * P, \
            "last layer's max readout + decode + Linear(%d,%d) + Linear(%d,%d)" % (P, E, E, C)
    if entry == "packed_head_fwd_f32":
        B, L, Fh, Fl, E, C = int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[12]), int(a[13])
        P = (L - 1) * Fh + Fl
        return 2.0 * B * (P * E + E * C), 8 * B * P + f4 * (E * P + C * E) + 2 * f4 * B * P, \
            "decode of all layers' packed maxima + Linear(%d,%d) + Linear(%d,%d)" % (P, E, E, C)
    if entry in ("head2_bwd_ce_f32", "head2_bwd_f32"):
        o = 0 if entry == "head2_bwd_ce_f32" else -1
        B, P, E, C = int(a[9 + o]), int(a[10 + o]), int(a[11 + o]), int(a[12 + o])
        return 4.0 * B * (P * E + E * C), 2 * f4 * (E * P + C * E) + 2 * f4 * B * P, "CE loss + head backward (dW1, db1, dW2, db2, d readout)"
    if entry == "adam_from_partials_f32":
        n = int(a[4])
        return 0.0, 7 * f4 * n, "clip + Adam on the flat buffer (norm from the producers' shares)"
    if entry == "clip_adam_step_f32":
        n = int(a[4])
        return 0.0, 7 * f4 * n, "clip_grad_norm + Adam on the flat buffer (every block sums the norm)"
    if entry == "readout_l2_bwd_f32":
        B, n, sg, F = int(a[2]), int(a[3]), int(a[4]), int(a[10])
        return 0.0, 2 * f4 * (n + sg) * F + f4 * (n + sg) + 4 * n + 8 * B * F, "backward of the last layer's max readout + L2 normalise -> dU (row-parallel), F=%d" % F
    if entry == "readout_partial_f32":
        n, sg, F = int(a[3]), int(a[4]), int(a[7])
        return 0.0, f4 * (n + sg) * F + 8 * int(a[1]) * F, "max-readout partial"
    return 0.0, 0, entry


def step_kernel_table(gstep, trainer, stream, iters=100):
    """Record ONE eager step (every launch with its arguments and the kernel the library dispatched), then time each launch on
    its own as a hipGraph burst with the step's own operands.  The trainer's state is put back afterwards."""
    import torch
    from two_stage_gnn_amd import _native as nat
    snap = [t.clone() for t in (trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state)]
    with torch.cuda.stream(stream):
        torch.cuda.synchronize()
        nat.trace = []
        try:
            gstep._fwd_bwd()            # (no collective here: only rank 0 runs this table)
            trainer.apply()
        finally:
            rec, nat.trace = nat.trace, None
        torch.cuda.synchronize()
        rows = []
        for name, args, kernel in rec:
            fn = (lambda name=name, args=args: nat.call(name, *args))
            ms = burst_ms(fn, iters, stream, repeats=3)
            rows.append((name, args, kernel, ms * 1e3))
        for t, s_ in zip((trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state), snap):
            t.copy_(s_)
        torch.cuda.synchronize()
    return rows


def summarise_kernels(rows, nnz):
    """group consecutive / repeated launches of the same kernel with the same accounting into one table row"""
    table, index = [], {}
    for name, args, kernel, us in rows:
        flops, nbytes, what = account(name, args, nnz)
        key = (kernel or name, what)
        if key not in index:
            index[key] = len(table)
            table.append({"kernel": kernel or ("tsgnn_" + name), "entry": "tsgnn_" + name, "what": what, "launches_per_step": 0,
                          "us": [], "flops": flops, "bytes": nbytes})
        r = table[index[key]]
        r["launches_per_step"] += 1
        r["us"].append(us)
    for r in table:
        us = sum(r["us"]) / len(r["us"])
        r["us_per_launch"] = us
        r["us_per_step"] = us * r["launches_per_step"]
        del r["us"]
        tf = r["flops"] / (us * 1e-6) / 1e12
        gbs = r["bytes"] / (us * 1e-6) / 1e9
        r["tflops"], r["gbs"] = tf, gbs
        r["frac_mfma"], r["frac_hbm"] = tf / MFMA_F32_PEAK_TF, gbs / HBM_PEAK_GBS
        r["bound"] = "mfma" if r["frac_mfma"] >= r["frac_hbm"] else "hbm"
        r["frac"] = max(r["frac_mfma"], r["frac_hbm"])
        if gbs > HBM_COPY_CEILING_GBS:
            # a burst rate above the chip's measured copy ceiling is cache bandwidth, not an HBM rate: no HBM fraction is claimed
            r["frac_hbm"] = None
            r["frac"] = r["frac_mfma"] if r["bound"] == "mfma" else None
            r["note"] = "burst rate %.0f GB/s exceeds the %.0f GB/s HBM copy ceiling: cache-resident replay, no HBM fraction" % (gbs, HBM_COPY_CEILING_GBS)
        # the burst replays ONE launch back to back on the operands the previous replay left in the L2s / the 256 MB Infinity Cache:
        # `gbs` of a burst-timed row is algorithmic bytes over time, NOT a measured HBM rate (a value near or above the chip's
        # 6.3 TB/s copy ceiling is cache bandwidth); the rocprofv3 averages of the same kernels inside the replayed step are in
        # profiles/ (bench_b32_kernel_stats.csv)
        r["cache_hot"] = True
    return table


# ----------------------------------------------------------------------------------------------- aggregation kernel
def aggregation_probe(g, feat, iters=300):
    """tsgnn_ell_spmm_f32 / tsgnn_csr_spmm_f32 at F = feat on the batch's own neighbour table; (ms, bytes, kernel)."""
    import torch
    from two_stage_gnn_amd import message_passing as mp
    from two_stage_gnn_amd import _native as nat
    from two_stage_gnn_amd.synthetic import aggregation_bytes
    x = torch.randn(g.total_rows, feat, device="cuda")
    y = torch.empty_like(x)
    s = torch.cuda.current_stream()
    use_ell = g.val is None and mp.ell_ok(x) and g.total_rows <= mp.ELL_MAX_ROWS
    if use_ell:
        g.ell()
        fn = lambda: mp.spmm_ell(g, x, out=y)               # what aggregate() launches for cache-resident batches
    else:
        fn = lambda: mp.spmm_raw(g.rowptr, g.col, g.val, x, g.total_rows, out=y)
    fn()
    kernel = nat.last_kernel()
    ms = burst_ms(fn, iters, s)
    nbytes = aggregation_bytes(g.total_rows, g.nnz, feat, weighted=g.val is not None)
    return ms, nbytes, "%s (tsgnn_%s, F=%d)" % (kernel, "ell_spmm_f32" if use_ell else "csr_spmm_f32", feat)


def load_traffic():
    """PMC-measured HBM bytes per launch (rocprofv3 --pmc in separate passes, scripts/pmc_traffic.*): a recorded measurement,
    not taken in this run — labelled as such."""
    for rel in (os.path.join(PROFILE_DIR, "step_traffic.json"), os.path.join("profiles", "r01", "agg_traffic.json")):
        try:
            return json.load(open(os.path.join(ROOT, rel))), rel
        except (OSError, ValueError):
            continue
    return {}, None


# ----------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(hb, hidden, layers, steps, state):
    """The reference's dense formulation (adj[B,Nmax,Nmax] @ x, encoders.py:30-42,169-217) restated by the
    CPU oracle, fwd + CE + bwd + clip + Adam, on this host's cores — test infrastructure used as the checker /
    baseline only (never on the product path)."""
    import torch
    from oracle import dense_ref as R
    from two_stage_gnn_amd.synthetic import to_dense
    cores_available = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores_available, int(os.environ.get("TSGNN_CPU_THREADS", 16)))     # the GPU box's CPU share per GPU is 16
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    torch.set_num_threads(cores)
    x, adj = to_dense(hb)
    label = torch.from_numpy(hb["label"])
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in state.items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-3)

    def make_step(adj_):
        def one():
            opt.zero_grad()
            _, ypred = R.gcn_encoder(p, x, adj_, bn=True, final_dim="number_classes")
            loss = torch.nn.functional.cross_entropy(ypred, label)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(list(p.values()), 2.0)
            opt.step()
        return one

    def timed(one, steps, budget):
        one()
        t0 = time.perf_counter(); one(); t1 = time.perf_counter()
        if steps <= 0:
            steps = int(max(3, min(200, budget / max(t1 - t0, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        return (time.perf_counter() - t0) / steps, steps

    dt, steps_d = timed(make_step(adj), steps, 12.0)
    # second line: the same step with the aggregation as a sparse product over the block-diagonal batch
    # adjacency; everything else (slot BN over the padded rows, normalise, readout) unchanged
    B, N = adj.size(0), adj.size(1)
    bi, ri, ci = adj.nonzero(as_tuple=True)
    adj_sp = torch.sparse_coo_tensor(torch.stack([bi * N + ri, bi * N + ci]), adj[bi, ri, ci], (B * N, B * N)).coalesce()
    dts, steps_s = timed(make_step(adj_sp), steps, 8.0)
    return {"value": len(hb["sizes"]) / dt, "unit": "graphs/s", "cores": cores, "cores_available": cores_available, "cpu_model": cpu_model,
            "cores_note": "threads used = min(cores available to this process, TSGNN_CPU_THREADS = 16: the box's CPU share per GPU)",
            "kind": "port",
            "sample": "%d steps of the same %d-graph batch, dense adj@x formulation at Nmax=%d (%.1f ms/step)"
                      % (steps_d, len(hb["sizes"]), hb["nmax"], dt * 1e3),
            "sparse_variant": {"value": len(hb["sizes"]) / dts, "unit": "graphs/s",
                               "sample": "%d steps, aggregation as torch.sparse.mm over the block-diagonal batch adjacency, "
                                         "padded rows otherwise as the reference (%.1f ms/step)" % (steps_s, dts * 1e3)}}


# ----------------------------------------------------------------------------------------------- the step over seeds
def over_seeds(a, model, trainer, dev, seeds, steps):
    """The timed step on the batches of several seeds (at N ranks, rank r draws seed r: the slowest of them paces every
    all-reduce), each replayed from its own hipGraph on the same model / trainer.  The headline batch (seed 0: 8,151 rows =
    255 row panels, one per compute unit) is a favourable case; the mean and the worst of seeds 0-7 say what a DD batch costs."""
    import torch
    from two_stage_gnn_amd import synthetic
    from two_stage_gnn_amd.data_parallel import GraphedStep
    out = []
    snap = [t.clone() for t in (trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state)]
    for seed in seeds:
        hb = synthetic.host_batch(seed=seed, B=a.batch, shape=a.shape, nmax=a.nmax)
        g, x, label = synthetic.to_device(hb, dev)
        gs = GraphedStep(trainer, lambda: model.loss(model(x, g)[1], label), warmup=3, steps_per_replay=max(1, a.steps_per_graph))
        gs.run(12)
        gs.stream.synchronize()
        best = None
        for _ in range(3):
            ms = hip_event_ms(lambda: gs.run(steps), 1, gs.stream) / steps
            best = ms if best is None else min(best, ms)
        out.append({"seed": seed, "rows": int(g.n_rows), "row_panels": (int(g.n_rows) + 31) // 32, "edges_directed": int(g.nnz),
                    "ms_per_step": best})
        del gs, g, x, label
    for t, s_ in zip((trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state), snap):
        t.copy_(s_)
    torch.cuda.synchronize()
    ms = [r["ms_per_step"] for r in out]
    mean = sum(ms) / len(ms)
    return {"per_seed": out, "mean_ms_per_step": mean, "worst_ms_per_step": max(ms), "best_ms_per_step": min(ms),
            "worst_over_best": max(ms) / min(ms), "value_mean": a.batch / (mean * 1e-3), "value_worst": a.batch / (max(ms) * 1e-3),
            "unit": "graphs/s", "note": "HIP events around %d replays per seed, best of 3; one GPU; the same model and optimiser state" % steps}


# ----------------------------------------------------------------------------------------------- ingest on the clock
def ingest_run(a, model, trainer, dev, steps, rank, value):
    """graphs/s when every step consumes a NEW mini-batch (f1, train.py:110-119 / graph_sampler.py:102-114): random batches of a
    512-graph synthetic TU-style dataset; native worker threads collate the batches ahead (C) into pinned staging buffers, the
    first two launches of the slot's own hipGraph pull the staged batch over PCIe and expand it on the device, then the step
    (two_stage_gnn_amd/ingest.py)."""
    import numpy as np
    import torch
    from two_stage_gnn_amd import ingest
    ds = ingest.synthetic_dataset(seed=4242 + rank, n_graphs=512, shape=a.shape, nmax=a.nmax)
    rng = np.random.default_rng(77 + rank)
    warm = 20
    sched = [rng.choice(len(ds), size=a.batch, replace=False) for _ in range(steps + warm)]
    pipe = ingest.IngestPipeline(model, trainer, ds, a.batch, a.nmax, dev, sched)
    rows = [int(ds.sizes[ids].sum()) for ids in sched]
    # (1) host collate alone (one core)
    t0 = time.perf_counter()
    scratch = np.zeros(pipe.slots[0].words, dtype=np.int32)
    for ids in sched[:50]:
        ingest.host_collate_compact(ds, ids, a.batch, a.nmax, pipe.row_cap, pipe.edge_cap, scratch)
    collate_us = (time.perf_counter() - t0) / 50 * 1e6
    # (2) the capacity-padded step alone: replays of one slot, no new batches (what the padding costs)
    torch.cuda.synchronize()
    for _ in range(warm):
        pipe.steps[0].step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.steps[0].step()
    torch.cuda.synchronize()
    cap_ms = (time.perf_counter() - t0) / steps * 1e3
    # (3) a new batch every step, double-buffered
    pipe.run(sched[:warm])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(sched[warm:])
    torch.cuda.synchronize()
    ov_ms = (time.perf_counter() - t0) / steps * 1e3
    # (4) the same without overlap: collate, upload, step, wait
    t0 = time.perf_counter()
    nser = min(steps, 50)
    for ids in sched[warm:warm + nser]:
        pipe.run([ids], workers=0)
        torch.cuda.synchronize()
    ser_ms = (time.perf_counter() - t0) / nser * 1e3
    # (5) what the drawn batches cost WITHOUT ingest: the scheduled batch closest to the mean size, resident in HBM as an exact
    # packed batch (no capacity padding), replayed like the headline step — separates the batch-size effect (the drawn batches
    # are larger than the headline batch) from what pull + expand + capacity padding add
    from two_stage_gnn_amd.data_parallel import GraphedStep
    k_mean = int(np.argmin(np.abs(np.asarray(rows) - np.mean(rows))))
    g_m, x_m, y_m = ds.collate(sched[k_mean], a.nmax, ds.features("node-label"), dev)
    gs_m = GraphedStep(trainer, lambda: model.loss(model(x_m, g_m)[1], y_m), warmup=2)
    for _ in range(warm):
        gs_m.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gs_m.step()
    torch.cuda.synchronize()
    same_ms = (time.perf_counter() - t0) / steps * 1e3
    return {"value": a.batch / (ov_ms * 1e-3), "unit": "graphs/s", "ms_per_step": ov_ms, "steps": steps,
            "vs_resident_input": (a.batch / (ov_ms * 1e-3)) / value,
            "resident_mean_size_batch": {"rows": int(rows[k_mean]), "ms_per_step": same_ms, "vs": same_ms / ov_ms,
                                         "note": "the scheduled batch closest to the mean size as an exact resident batch: `vs` = its "
                                                 "step time / the ingest step time (the rest is pull + expand + capacity padding); the "
                                                 "headline batch is smaller than the drawn batches"},
            "without_overlap": {"value": a.batch / (ser_ms * 1e-3), "ms_per_step": ser_ms, "steps": nser},
            "capacity_padded_step_only_ms": cap_ms,
            "row_capacity": pipe.row_cap, "rows_mean": float(np.mean(rows)), "rows_max": int(np.max(rows)),
            "ghost_slots": pipe.slots[0].g.ghost_slots_fixed,
            "host_collate_us_per_batch_one_core": collate_us,
            "pcie_bytes_per_batch_mean": float(4 * (4 + a.batch + 2 + a.nmax + 2 * a.batch + 3 * np.mean(rows) + 2
                                                    + np.mean([int((ds.rowptr[ds.graph_ptr[ids + 1]] - ds.rowptr[ds.graph_ptr[ids]]).sum()) for ids in sched]))),
            "note": "every step draws %d new graphs from a 512-graph %s-shaped dataset; node-label (one-hot) features expanded on the "
                    "device; one hipGraph per slot (the step, with the NEXT batch's PCIe pull and expansion riding in its first "
                    "hidden-layer and head launches) replays every batch (capacity-padded rows); 2 native collate threads"
                    % (a.batch, a.shape)}


def triplet_run(a, dev, with_cpu):
    """f3 — the 2stg / 2stg+ training step (tripletnet.py:16-45, train_triplet.py:247-287): ONE triplet (anchor, positive, negative)
    per optimiser step, MarginRankingLoss on the pairwise distances of the three embeddings, clip 2.0 + Adam.  The reference runs the
    encoder three times at B = 1 on dense [1, Nmax, Nmax] adjacencies; here the three graphs are one block-diagonal batch with the
    per-graph batch-norm statistics a B = 1 forward has (two_stage_gnn_amd/triplet.py).  Three figures: the drop-in module fed from
    host dicts every step as the reference's loop feeds it (upload and dense->CSR inside the step), the same step on a resident
    packed triplet replayed from a hipGraph, and the CPU oracle's three B = 1 forwards."""
    import numpy as np
    import torch
    from two_stage_gnn_amd import dense_encoders as E, synthetic
    from two_stage_gnn_amd.triplet import tripletnet
    from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep
    hb = synthetic.host_batch(11, 3, a.shape, a.nmax)
    x, adj = synthetic.to_dense(hb)                                   # the reference's inputs: dense, padded to Nmax, on the host

    class Args:
        bias = True

    class G_:
        def __init__(self, b):
            self.graph = {"adj": adj[b].numpy(), "feats": x[b].numpy(), "num_nodes": int(hb["sizes"][b]), "assign_feats": x[b].numpy()}

    torch.manual_seed(5)
    model = E.GcnEncoderGraph(hb["fin"], a.hidden, a.hidden, 2, a.layers, bn=True, args=Args(), final_dim="output_dim").to(dev)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    net = tripletnet(model)
    trip = [G_(b) for b in range(3)]
    crit = torch.nn.MarginRankingLoss(margin=1.0)
    target = torch.full((1,), -1.0, device=dev)
    params = [q for q in model.parameters() if q.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)

    def drop_in():
        opt.zero_grad(set_to_none=True)
        dp, dn = net(*trip)[:2]
        crit(dp, dn, target).backward()
        torch.nn.utils.clip_grad_norm_(params, 2.0)
        opt.step()

    for _ in range(5):
        drop_in()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_e = 30
    for _ in range(n_e):
        drop_in()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / n_e * 1e3
    # resident: the packed triplet in HBM, fwd + loss + bwd + clip + Adam from one hipGraph
    g3, x3, _ = synthetic.to_device(hb, dev)
    trainer = FlatTrainer(model, lr=1e-3, clip=2.0)

    def loss_fn():
        dp, dn = net._embed(x3, g3, hb["sizes"], x3)[:2]               # what tripletnet.forward runs once the batch is assembled
        return crit(dp, dn, target)

    n_g = 200

    def replay_ms(fn):
        gs = GraphedStep(trainer, fn, warmup=3)
        for _ in range(10):
            gs.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_g):
            gs.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_g * 1e3

    graph_ms = replay_ms(loss_fn)
    from two_stage_gnn_amd.triplet import MarginRankingLoss
    crit_f = MarginRankingLoss(margin=1.0)                              # the drop-in criterion: one launch each way

    def loss_fn_f():
        dp, dn = net._embed(x3, g3, hb["sizes"], x3)[:2]
        return crit_f(dp, dn, target)

    graph_f_ms = replay_ms(loss_fn_f)
    out = {"unit": "triplets/s",
           "config": "%s-shaped triplet (%s nodes, Nmax %d), GcnEncoderGraph %d layers h=%d final_dim='output_dim', margin 1, clip 2.0 + Adam"
                     % (a.shape, "/".join(str(int(v)) for v in hb["sizes"]), a.nmax, a.layers, a.hidden),
           "drop_in_from_host_dicts": {"value": 1e3 / eager_ms, "ms_per_step": eager_ms, "steps": n_e,
                                       "note": "tripletnet(model)(a, p, n) on `.graph` dicts as the reference's loop passes them (a graph "
                                               "object's CSR rows and features go to the device at its first use and stay resident), eager "
                                               "launches, torch clip + Adam"},
           "resident_hipgraph": {"value": 1e3 / graph_ms, "ms_per_step": graph_ms, "steps": n_g,
                                 "note": "the packed triplet resident in HBM, forward + loss + backward + clip + Adam replayed from one "
                                         "hipGraph (per-graph batch-norm: the per-op kernels, not the fused stack)"},
           "resident_hipgraph_library_criterion": {"value": 1e3 / graph_f_ms, "ms_per_step": graph_f_ms, "steps": n_g,
                                                   "note": "the same with two_stage_gnn_amd.triplet.MarginRankingLoss (one launch each "
                                                           "way) in place of torch.nn.MarginRankingLoss (~17 element-wise launches)"},
           "cpu_baseline": None}
    if with_cpu:
        from oracle import dense_ref as R
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, int(os.environ.get("TSGNN_CPU_THREADS", 16)))
        torch.set_num_threads(cores)
        pr = {k: v.clone().requires_grad_(True) for k, v in state.items()}
        opt_c = torch.optim.Adam(list(pr.values()), lr=1e-3)
        tgt = torch.tensor([-1.0])

        def cpu_step():
            opt_c.zero_grad()
            e = [R.gcn_encoder(pr, x[b:b + 1], adj[b:b + 1], bn=True, final_dim="output_dim")[1] for b in range(3)]
            dp = torch.nn.functional.pairwise_distance(e[0], e[1], 2)
            dn = torch.nn.functional.pairwise_distance(e[0], e[2], 2)
            crit(dp, dn, tgt).backward()
            torch.nn.utils.clip_grad_norm_(list(pr.values()), 2.0)
            opt_c.step()

        cpu_step()
        t0 = time.perf_counter(); cpu_step(); t1 = time.perf_counter()
        n_c = int(max(3, min(200, 6.0 / max(t1 - t0, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n_c):
            cpu_step()
        dt = (time.perf_counter() - t0) / n_c
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "triplets/s", "cores": cores, "kind": "port",
                               "sample": "%d steps of the same triplet: three B = 1 dense forwards of the oracle at Nmax=%d, margin loss, "
                                         "backward, clip, Adam (%.1f ms/step)" % (n_c, a.nmax, dt * 1e3)}
    return out


def pyg_surface_run(a, dev, hb, ms_surface_a):
    """The SAME synthetic batch through the torch_geometric-named operator surface the north_star lists (pyg.SageNet: SAGEConv layers
    as fused launches, csrc/sageconv.hip; [gmp || gap] readouts in the layers' epilogues; lin1-3 head; nll) as a full optimiser step
    from one hipGraph — beside `value`, which times the reference's own GcnEncoderGraph.  PARITY UNPINNED (no PyG in the reference
    tree or this image): tests/test_gpu_fullsize.py::test_pyg_sage_timed_step_vs_oracle checks this very step against oracle/pyg_ref.py."""
    import torch
    from two_stage_gnn_amd import message_passing as mp, pyg, synthetic
    from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep

    class D:
        pass
    d = D()
    d.x, d.edge_index, d.batch, lab = synthetic.to_pyg(hb, dev)
    torch.manual_seed(1234)
    net = pyg.SageNet(synthetic.SHAPES[a.shape][2], a.hidden, 2, num_layers=a.layers).to(dev).train()
    tr = FlatTrainer(net, lr=1e-3, clip=2.0, defer_loss=True)
    gs = GraphedStep(tr, lambda: mp.nll_loss(net(d), lab), warmup=3)
    n = max(50, min(a.steps, 200))
    with torch.cuda.stream(gs.stream):
        gs.run(64); torch.cuda.synchronize()
        ms = min(hip_event_ms(lambda: gs.run(n), 1, gs.stream) / n for _ in range(3))
    mp.check_device_errors()
    return {"model": "pyg.SageNet: %d x SAGEConv(mean) h=%d + ReLU, sum over layers of [global_max_pool || global_mean_pool], lin1-3, "
                     "log_softmax, nll; clip 2.0 + Adam" % (a.layers, a.hidden),
            "ms_per_step": ms, "value": a.batch / (ms * 1e-3), "unit": "graphs/s", "launch": gs.describe(),
            "vs_reference_surface_step": ms / ms_surface_a, "rows": int(d.x.size(0)),
            "note": "surface (B) of SURVEY 8(b): the operators BASELINE.json's north_star names; one step per hipGraph launch; "
                    "the reference itself never calls SAGEConv (parity unpinned)"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)                               # before anything touches the GPU; does not return
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    # stdout carries exactly ONE JSON line: libraries that write banners to fd 1 (RCCL prints its version block there at
    # communicator creation) are pointed at stderr until the line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)            # one GPU per rank on the 8-GPU node; TSGNN_DIST_BACKEND=gloo lets
    torch.cuda.set_device(dev_index)                 # several ranks share one GPU for a functional rehearsal
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("TSGNN_DIST_BACKEND", "nccl")
    # TSGNN_FORCE_DIST=1 takes the N > 1 code path (process group, hipGraphs around the RCCL all-reduce) with a single
    # rank: the rehearsal of the multi-GPU path that fits a one-GPU box.
    multi = world > 1 or os.environ.get("TSGNN_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from two_stage_gnn_amd import dense_encoders as E
    from two_stage_gnn_amd import synthetic
    from two_stage_gnn_amd.data_parallel import FlatTrainer, GraphedStep

    class Args:
        bias = True
    torch.manual_seed(1234)                                   # identical initial weights on every rank (and broadcast by the trainer)
    fin = synthetic.SHAPES[a.shape][2]
    model = E.GcnEncoderGraph(fin, a.hidden, a.hidden, 2, a.layers, bn=True, args=Args(), final_dim="number_classes").to(dev)
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    hb = synthetic.host_batch(seed=a.seed_base + rank, B=a.batch, shape=a.shape, nmax=a.nmax)      # per-rank batch (weak scaling)
    g, x, label = synthetic.to_device(hb, dev)
    trainer = FlatTrainer(model, lr=1e-3, clip=2.0, defer_loss=True)     # the loss value is written by the head's backward kernel
    trainer.always_reduce = multi

    use_graph = not a.no_graph
    # N = 1: one hipGraph for the whole step; N > 1: hipGraphs around the RCCL all-reduce (data_parallel.GraphedStep)
    gstep = GraphedStep(trainer, lambda: model.loss(model(x, g)[1], label), warmup=3, use_graph=use_graph,
                        steps_per_replay=1 if multi else max(1, a.steps_per_graph))
    stream = gstep.stream
    settle_steps = 0
    ms_unsettled = None
    with torch.cuda.stream(stream):
        if use_graph and a.settle_steps > 0:     # part of the set-up (capture + settle), not of the W warm-up steps below
            # the same W + K steps right after capture, BEFORE the settling replays: what the driver's short run measured in the
            # rounds without them (reported as `ms_per_step_unsettled`; every rank runs the same count)
            gstep.run(a.warmup)
            torch.cuda.synchronize()
            tu = time.perf_counter()
            gstep.run(a.steps)
            torch.cuda.synchronize()
            ms_unsettled = (time.perf_counter() - tu) / a.steps * 1e3
            settle_steps = int(a.settle_steps)   # (a fixed count: every rank issues the same collectives)
            gstep.run(settle_steps)
            torch.cuda.synchronize()
        gstep.run(a.warmup)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gstep.run(a.steps)                       # exactly a.steps optimiser steps (graph launches of steps_per_replay steps + the rest)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        per_rank = None
        if multi:
            mine = torch.tensor([elapsed, float(g.n_rows), float(g.nnz)], device=dev, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [{"rank": r, "seed": r, "rows": int(v[1].item()), "edges_directed": int(v[2].item()),
                         "ms_per_step": float(v[0].item()) / a.steps * 1e3} for r, v in enumerate(allr)]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        allreduce = None
        if multi:
            # the gradient all-reduce BY ITSELF (the bucket of the timed step, same communicator and stream), 50 back to back, slowest
            # rank: with TSGNN_AR_BUCKETS = 1 (default) the collective sits between the backward and the optimiser of every step, so
            # this is also what the step exposes of it; every rank issues the same count (collective-safe)
            keep = trainer._grad_store.clone()    # (55 sums of the same bucket: put the step's own gradients back afterwards)
            for _ in range(5):
                trainer.all_reduce()
            torch.cuda.synchronize()
            dist.barrier()
            ar_ms = torch.tensor([hip_event_ms(trainer.all_reduce, 50, stream)], device=dev, dtype=torch.float64)
            dist.all_reduce(ar_ms, op=dist.ReduceOp.MAX)
            allreduce = {"us_alone": float(ar_ms.item()) * 1e3, "bytes": int(trainer._grad_store.numel()) * 4,
                         "buckets": int(trainer.buckets), "inside_the_hipgraph": bool(getattr(gstep, "one_graph", False)),
                         "backend": backend,
                         "note": "50 consecutive all-reduces of the flat gradient bucket (+ the 16-byte error-word slot) on the step's stream, "
                                 "max over ranks; one per optimiser step, between backward and optimiser (exposed unless TSGNN_AR_BUCKETS=2 "
                                 "starts the head's share early)"}
            trainer._grad_store.copy_(keep)
            trainer._err_slot.zero_()
            torch.cuda.synchronize()
    from two_stage_gnn_amd import message_passing as mp_
    mp_.check_device_errors()                     # the host has synchronised: did any step report invalid results?

    out = None
    if rank == 0:
        ms_step = elapsed / a.steps * 1e3
        out = {
            "metric": "graphs/sec fwd+bwd, DD batch=32 SAGE-3L h=128",
            "value": world * a.batch * a.steps / elapsed, "unit": "graphs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-shaped synthetic graphs (avg %d nodes / %d edges, F_in=%d), batch=%d per GPU, "
                                   "GcnEncoderGraph (GraphSage 'base') %d layers h=%d, Nmax=%d, slot-BN, CE loss, "
                                   "clip 2.0 + Adam; %d step(s) per hipGraph launch, seed = rank, %d untimed settle steps before the warm-up"
                                   % (a.shape, *synthetic.SHAPES[a.shape], a.batch, a.layers, a.hidden, a.nmax,
                                      1 if multi else max(1, a.steps_per_graph), settle_steps),
                       "global_batch": world * a.batch, "parallelism": "dp%d" % world,
                       "launch": ("hipGraph replay (%s)" % gstep.describe()) if use_graph else "eager",
                       "settle": {"untimed_steps": settle_steps,
                                  "note": "set-up, before the warm-up steps: replays of the freshly captured step until its time has "
                                          "settled (the first ~80 replays after capture run 3-4 % slower; --settle-steps 0 disables)"},
                       "rows": int(g.n_rows), "edges_directed": int(g.nnz)},
        }
        # what `value` is and is not, at the TOP level of the line (VERDICT r3 #4 / ADVICE r3): the batch of seed = rank (seed 0 is the
        # most favourable DD batch: 255 row panels on 256 CUs), timed after `settle_untimed_steps` replays of the captured step
        out["steps_per_graph_launch"] = 1 if multi else max(1, a.steps_per_graph)
        out["settle_untimed_steps"] = settle_steps
        out["ms_per_step_unsettled"] = ms_unsettled
        if out["steps_per_graph_launch"] == 1:
            out["ms_per_step_one_step_per_graph"] = ms_step
        if per_rank is not None:
            out["per_rank"] = per_rank            # every rank's own batch and its own clock around the same K steps
        if allreduce is not None:
            out["allreduce"] = allreduce
        roofline = {}
        traffic, traffic_src = load_traffic()
        with torch.cuda.stream(stream):
            if not a.no_kernels:
                rows = step_kernel_table(gstep, trainer, stream)
                table = summarise_kernels(rows, int(g.nnz))
                top = max((r for r in table if r["frac"] is not None), key=lambda r: r["us_per_step"])
                for r_ in table:                                # PMC traffic (recorded measurement) next to the algorithmic bytes
                    t_ = traffic.get(r_["kernel"].split(" (")[0].replace(" ", ""))
                    if isinstance(t_, dict):
                        r_["traffic"] = t_.get("traffic_bytes_per_launch")
                tr = traffic.get(top["kernel"].split(" (")[0].replace(" ", ""), {})
                tr = tr if isinstance(tr, dict) else {}
                roofline = {"bound": top["bound"], "kernel": "%s (%s: %s)" % (top["kernel"], top["entry"], top["what"]),
                            "achieved": top["tflops"] if top["bound"] == "mfma" else top["gbs"],
                            "peak": MFMA_F32_PEAK_TF if top["bound"] == "mfma" else HBM_PEAK_GBS,
                            "unit": "TFLOP/s" if top["bound"] == "mfma" else "GB/s", "frac": top["frac"],
                            "traffic": tr.get("traffic_bytes_per_launch"),
                            "traffic_source": (traffic_src + " (rocprofv3 --pmc, separate run; not measured in this process)")
                            if tr else None,
                            "flops_per_launch": top["flops"], "bytes_per_launch": top["bytes"], "us_per_launch": top["us_per_launch"],
                            "launches_per_step": top["launches_per_step"],
                            "mfma": {"achieved": top["tflops"], "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": top["frac_mfma"]},
                            "hbm": {"achieved": top["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": top["frac_hbm"]},
                            "selection": "the kernel with the largest launches x duration in the timed step",
                            "timing": "every row: HIP events around a hipGraph burst of that very launch (its own arguments and buffers); "
                                      "operands are cache-resident from the previous replay (`cache_hot`), as they largely are inside the "
                                      "replayed step, where each kernel's inputs were written by the launch before it",
                            "step_kernels": table,
                            "step_launches": sum(r["launches_per_step"] for r in table),
                            "step_kernel_us_sum": sum(r["us_per_step"] for r in table),
                            "step_flops": sum(r["flops"] * r["launches_per_step"] for r in table),
                            "step_bytes": sum(r["bytes"] * r["launches_per_step"] for r in table)}
                roofline["step"] = {"mfma_frac": roofline["step_flops"] / (ms_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                                    "hbm_frac": roofline["step_bytes"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
                # the aggregation-only bytes of one pass over the real rows / the fused kernel that contains it
                n = int(g.n_rows)
                fused = [r for r in table if r["entry"] in ("tsgnn_sage_layer_fwd_f32", "tsgnn_sage_layer_fwd_ro_f32", "tsgnn_gather_rowgemm_f32",
                                                            "tsgnn_sage_layer_fwd_bn_f32", "tsgnn_gather_rowgemm_st_f32")]
                agg_in = []
                for r in fused:
                    K = a.hidden if r["entry"].startswith("tsgnn_sage_layer_fwd") else x.size(1)
                    nb = synthetic.aggregation_bytes(n, int(g.nnz), K)
                    agg_in.append({"kernel": r["kernel"], "F": K, "aggregation_bytes": nb, "us_per_launch": r["us_per_launch"],
                                   "achieved": nb / r["us_per_launch"] / 1e3, "unit": "GB/s",
                                   "frac": nb / r["us_per_launch"] / 1e3 / HBM_PEAK_GBS})
                roofline["aggregation_in_fused"] = agg_in
            # the stand-alone aggregation kernel on the step's batch and its batch-size sweep
            agg_ms, agg_bytes, agg_kernel = aggregation_probe(g, a.hidden)
            tr = {}
            if a.shape == "DD" and a.batch == 32 and a.hidden == 128 and int(g.total_rows) == 9151:       # the shape the PMC run measured
                tr = traffic.get(agg_kernel.split(" ")[0].replace(" ", ""), traffic.get("dd_b32_rows9151_f128", {}))
                tr = tr if isinstance(tr, dict) else {}
            roofline["aggregation_standalone"] = {
                "bound": "hbm", "kernel": agg_kernel, "achieved": agg_bytes / agg_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": agg_bytes / agg_ms / 1e6 / HBM_PEAK_GBS, "traffic": tr.get("traffic_bytes_per_launch"),
                "traffic_source": (traffic_src + " (rocprofv3 --pmc, separate run)") if tr else None,
                "bytes_per_launch": agg_bytes, "us_per_launch": agg_ms * 1e3, "cache_hot": True,
                "timing": "HIP events around a hipGraph burst of the launch: launch-to-launch time of back-to-back replays on "
                          "cache-resident operands; on a 4-us kernel consecutive launches overlap ramp and drain, so this is the "
                          "OPTIMISTIC figure — `rocprof` below is the kernel's own begin-to-end time and the one to quote",
                "note": "not launched by the timed step (the step aggregates inside the fused kernels); what non-fused callers run"}
            try:                                             # recorded measurement (rocprofv3 --kernel-trace --stats of this command)
                rec = json.load(open(os.path.join(ROOT, PROFILE_DIR, "agg_rocprof.json")))
                k0 = agg_kernel.split(" ")[0].replace(" ", "")
                if k0 in rec and a.shape == "DD" and a.batch == 32 and a.hidden == 128:
                    us = float(rec[k0]["us"])
                    roofline["aggregation_standalone"]["rocprof"] = {
                        "us_per_launch": us, "achieved": agg_bytes / us / 1e3, "unit": "GB/s", "frac": agg_bytes / us / 1e3 / HBM_PEAK_GBS,
                        "source": PROFILE_DIR + "/agg_rocprof.json (kernel begin-to-end, rocprofv3 --kernel-trace --stats of bench.py; "
                                  "a recorded measurement, not taken in this process)"}
            except (OSError, ValueError, KeyError):
                pass
            if not roofline.get("kernel"):
                roofline.update({k: roofline["aggregation_standalone"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")})
            if not a.no_sweep:
                sweep = []
                for B in [int(v) for v in a.sweep_sizes.split(",") if v]:
                    hs = synthetic.tiled_batch(seed=100 + B, B=B, shape=a.shape, nmax=a.nmax)
                    gsw = synthetic.structure_to_device(hs, dev)
                    ms, nb, kern = aggregation_probe(gsw, a.hidden, iters=20 if B > 2048 else (100 if B > 256 else 300))
                    sweep.append({"graphs": B, "rows": int(gsw.total_rows), "edges_directed": int(gsw.nnz), "bytes_per_launch": nb,
                                  "us_per_launch": ms * 1e3, "achieved": nb / ms / 1e6, "unit": "GB/s",
                                  "frac": nb / ms / 1e6 / HBM_PEAK_GBS, "kernel": kern})
                    print("sweep B=%d rows=%d: %.2f us, %.0f GB/s (%.1f%% of 8 TB/s)  %s"
                          % (B, gsw.n_rows, ms * 1e3, nb / ms / 1e6, nb / ms / 1e6 / HBM_PEAK_GBS * 100, kern), file=sys.stderr)
                    del gsw, hs
                    torch.cuda.empty_cache()
                roofline["sweep"] = sweep
                roofline["sweep_note"] = ("stand-alone aggregation, F=%d, DD-shaped graphs; batches above 256 graphs repeat 256 generated "
                                          "graphs (every copy owns its rows)" % a.hidden)
        out["roofline"] = roofline
        if world == 1 and use_graph:
            # the same resident batch with k = 4 (or 1) consecutive optimiser steps per graph launch: between two graph launches the
            # device idles for the launch's own latency, once per replay whatever the graph holds — a loop that refills its input
            # buffers between steps cannot use k > 1, which is why it is NOT `value`
            other_k = 4 if out["steps_per_graph_launch"] == 1 else 1
            snap_ = [t_.clone() for t_ in (trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state)]
            gk = GraphedStep(trainer, lambda: model.loss(model(x, g)[1], label), warmup=3, steps_per_replay=other_k)
            with torch.cuda.stream(gk.stream):
                gk.run(64); torch.cuda.synchronize()
                nk = max(40, min(a.steps, 200)) // 4 * 4
                msk = min(hip_event_ms(lambda: gk.run(nk), 1, gk.stream) / nk for _ in range(3))
            for t_, s_ in zip((trainer.flat_param, trainer.exp_avg, trainer.exp_avg_sq, trainer.state), snap_):
                t_.copy_(s_)
            torch.cuda.synchronize()
            out["ms_per_step_four_steps_per_graph" if other_k == 4 else "ms_per_step_one_step_per_graph"] = msk
            del gk
        if not a.no_seeds and world == 1 and use_graph:
            vs = over_seeds(a, model, trainer, dev, [int(v) for v in a.seeds.split(",") if v != ""], max(50, min(a.steps, 200)))
            out["value_over_seeds"] = vs
            out["value_mean_over_seeds"] = vs["value_mean"]          # what "a DD batch" costs: the mean of seeds 0-7, not the best seed
            out["value_worst_seed"] = vs["value_worst"]
            # what an N-rank step is paced by: rank r draws seed r, every rank waits for the slowest at the all-reduce
            roofline["straggler"] = {"ms_per_step": vs["worst_ms_per_step"], "seed": max(vs["per_seed"], key=lambda r: r["ms_per_step"])["seed"],
                                     "rows": max(vs["per_seed"], key=lambda r: r["ms_per_step"])["rows"],
                                     "vs_headline_batch": vs["worst_ms_per_step"] / ms_step,
                                     "note": "max over seeds 0-7 of the one-GPU step = the pace of an 8-rank step before the all-reduce"}
    if rank == 0 and world == 1 and use_graph and not a.no_pyg:
        out["pyg_surface"] = pyg_surface_run(a, dev, hb, out["ms_per_step"])
    if a.ingest and world == 1:
        out["ingest"] = ingest_run(a, model, trainer, dev, max(a.steps, 100), rank, out["value"])
    if a.triplet and world == 1:
        out["triplet"] = triplet_run(a, dev, not a.no_cpu_baseline)
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(hb, a.hidden, a.layers, a.cpu_steps, init_state)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
