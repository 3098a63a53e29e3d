#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): graphs/sec, forward+backward(+all-reduce+clip+Adam) of the
3-layer h=128 GraphSage-style encoder (GcnEncoderGraph, `--method=base`) on DD-shaped synthetic batches
of 32 graphs per GPU, on 1/2/4/8 MI355X, plus the HBM-roofline fraction of the dominant aggregation
kernel and the reference's dense formulation timed on the host CPU.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One JSON line on stdout (rank 0).  A "step" = one optimiser step on one batch per rank: forward, CE loss,
backward, gradient bucket (+RCCL all-reduce when N>1), clip_grad_norm(2.0), Adam — exactly the body of the
reference's loop (train.py:110-131) minus the host->device copies (inputs are resident in HBM).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="0 = auto-size the sample to ~15 s")
    ap.add_argument("--no-overlap", action="store_true", help="accepted for compatibility: the stack always runs on one stream")
    ap.add_argument("--sweep", action="store_true", help="also print the aggregation-kernel batch-size sweep (stderr)")
    return ap.parse_args()


def hip_event_ms(fn, iters, stream):
    """average ms per call of fn() measured with HIP events recorded on `stream`."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def aggregation_probe(g, feat, iters=300):
    """The dominant HBM kernel of the step (tsgnn_csr_spmm_f32 at F = hidden) launched back to back on the
    step's own CSR / buffers; duration from HIP events on the launching stream."""
    from two_stage_gnn_amd import message_passing as mp
    from two_stage_gnn_amd.synthetic import aggregation_bytes
    x = torch.randn(g.total_rows, feat, device="cuda")
    y = torch.empty_like(x)
    s = torch.cuda.current_stream()
    use_ell = g.val is None and mp.ell_ok(x) and g.total_rows <= mp.ELL_MAX_ROWS
    if use_ell:
        g.ell()
        fn = lambda: mp.spmm_ell(g, x, out=y)               # what the step's aggregate() launches
    else:
        fn = lambda: mp.spmm_raw(g.rowptr, g.col, g.val, x, g.total_rows, out=y)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    # the burst is replayed from a hipGraph so the host's ~8 us per ctypes launch does not pace it
    burst = torch.cuda.CUDAGraph()
    with torch.cuda.graph(burst, stream=s):
        for _ in range(iters):
            fn()
    burst.replay()
    torch.cuda.synchronize()
    ms = min(hip_event_ms(burst.replay, 1, s) for _ in range(5)) / iters
    nbytes = aggregation_bytes(g.total_rows, g.nnz, feat, weighted=g.val is not None)
    aggregation_probe.kernel = ("spmm_ell_vec4<32,16> (tsgnn_ell_spmm_f32" if use_ell else
                                ("spmm_vec4_rb<32,4,false> (tsgnn_csr_spmm_f32" if g.total_rows >= 262144 else
                                 "spmm_vec4<32,false,false> (tsgnn_csr_spmm_f32")) + ", F=%d)" % feat
    return ms, nbytes


MFMA_F32_PEAK_TF = 157.3       # dense fp32 MFMA, /opt/skills/guides/MI355X_MICROARCH.md:42


def fused_layer_probe(g, feat, iters=300):
    """The dominant kernel of the step when the aggregation is fused into the transform (tsgnn_gather_rowgemm_f32 at
    K = N = hidden, as layers 1.. of the forward launch it: neighbour gather + .W + bias + L2 normalise, z written for the
    weight gradient), launched back to back on the step's own neighbour table; HIP events on the launching stream.
    Returns (ms, flops, algorithmic bytes) per launch."""
    from two_stage_gnn_amd import _native as nat
    ell, ell_w, tail = g.ell()
    R = g.total_rows
    x = torch.randn(R, feat, device="cuda")
    w = torch.randn(feat, feat, device="cuda") * 0.1
    b = torch.randn(feat, device="cuda")
    v = torch.empty_like(x); z = torch.empty_like(x); rinv = torch.empty(R, device="cuda")
    gs = min(g.nmax, int(g.sizes.max()) + 1) if g.n_ghost else 0
    s = torch.cuda.current_stream()
    tp, tc = tail if tail is not None else (None, None)
    fn = lambda: nat.call("gather_rowgemm_f32", ell, ell_w, tp, tc, x, feat, w, feat, 0, b, v, feat, rinv, z, feat, g.n_rows, feat, feat, 1, gs)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    burst = torch.cuda.CUDAGraph()
    with torch.cuda.graph(burst, stream=s):
        for _ in range(iters):
            fn()
    burst.replay()
    torch.cuda.synchronize()
    ms = min(hip_event_ms(burst.replay, 1, s) for _ in range(5)) / iters
    n = int(g.n_rows)
    flops = 2.0 * n * feat * feat
    # SURVEY 8(d) aggregation bytes on the real rows (X read once, indices, row pointers) + z and v written + W + rinv
    nbytes = 4 * n * feat + 4 * int(g.nnz) + 4 * (n + 1) + 2 * 4 * n * feat + 4 * feat * feat + 4 * (n + gs) + 4 * gs * feat
    return ms, flops, nbytes


def cpu_baseline(hb, hidden, layers, steps, state):
    """The reference's dense formulation (adj[B,Nmax,Nmax] @ x, encoders.py:30-42,169-217) restated by the
    CPU oracle, fwd + CE + bwd + clip + Adam, on this host's cores — test infrastructure used as the checker /
    baseline only (never on the product path)."""
    from oracle import dense_ref as R
    from two_stage_gnn_amd.synthetic import to_dense
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("TSGNN_CPU_THREADS", 16)))     # the GPU box's CPU share per GPU is 16
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
    return {"value": len(hb["sizes"]) / dt, "unit": "graphs/s", "cores": cores, "kind": "port",
            "sample": "%d steps of the same %d-graph batch, dense adj@x formulation at Nmax=%d (%.1f ms/step)"
                      % (steps_d, len(hb["sizes"]), hb["nmax"], dt * 1e3),
            "sparse_variant": {"value": len(hb["sizes"]) / dts, "unit": "graphs/s",
                               "sample": "%d steps, aggregation as torch.sparse.mm over the block-diagonal batch adjacency, "
                                         "padded rows otherwise as the reference (%.1f ms/step)" % (steps_s, dts * 1e3)}}


def main():
    a = parse()
    # stdout carries exactly ONE JSON line: libraries that write banners to fd 1 (RCCL prints its version block there at
    # communicator creation) are pointed at stderr until the line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)            # one GPU per rank on the 8-GPU node; TSGNN_DIST_BACKEND=gloo lets
    torch.cuda.set_device(dev_index)                 # several ranks share one GPU for a functional rehearsal
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("TSGNN_DIST_BACKEND", "nccl")
    # TSGNN_FORCE_DIST=1 takes the N > 1 code path (process group, two hipGraphs around the RCCL all-reduce) with a single
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
    from two_stage_gnn_amd import sage_stack

    class Args:
        bias = True
    torch.manual_seed(1234)                                   # identical initial weights on every rank
    fin = synthetic.SHAPES[a.shape][2]
    model = E.GcnEncoderGraph(fin, a.hidden, a.hidden, 2, a.layers, bn=True, args=Args(), final_dim="number_classes").to(dev)
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    hb = synthetic.host_batch(seed=rank, B=a.batch, shape=a.shape, nmax=a.nmax)      # per-rank batch (weak scaling)
    g, x, label = synthetic.to_device(hb, dev)
    trainer = FlatTrainer(model, lr=1e-3, clip=2.0, defer_loss=True)     # the loss value is written by the head's backward kernel
    trainer.always_reduce = multi

    use_graph = not a.no_graph
    # N > 1: fwd+bwd+bucket and clip+Adam are two hipGraphs with the RCCL all-reduce issued between them on the same stream;
    # N = 1: one hipGraph for the whole step (data_parallel.GraphedStep)
    gstep = GraphedStep(trainer, lambda: model.loss(model(x, g)[1], label), warmup=3, use_graph=use_graph)
    stream = gstep.stream
    step = gstep.step
    with torch.cuda.stream(stream):
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

        out = None
        if rank == 0:
            ms_step = elapsed / a.steps * 1e3
            agg_ms, agg_bytes = aggregation_probe(g, a.hidden)
            achieved = agg_bytes / (agg_ms * 1e-3) / 1e9
            tr = {}                 # PMC-measured HBM bytes per launch of the same kernels/shape (scripts/pmc_traffic.*)
            try:
                if a.shape == "DD" and a.batch == 32 and a.hidden == 128 and int(g.total_rows) == 9151:
                    tr = json.load(open(os.path.join(ROOT, "profiles", "r01", "agg_traffic.json")))
            except (OSError, ValueError):
                pass
            standalone = {"bound": "hbm", "kernel": aggregation_probe.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS, "traffic": tr.get("dd_b32_rows9151_f128", {}).get("traffic_bytes_per_launch"),
                          "bytes_per_launch": agg_bytes, "us_per_launch": agg_ms * 1e3}
            roofline = standalone
            if sage_stack._gather_ok(g, x.new_empty(1, a.hidden)) and g.n_ghost > 0 and a.hidden <= 128:
                # the step aggregates inside the transform: that fused kernel is the dominant one; it sits at the ridge of the
                # two rooflines (20 flop per algorithmic byte vs 19.7 for the chip), so both fractions are given and the larger
                # one names the bound
                f_ms, f_flops, f_bytes = fused_layer_probe(g, a.hidden)
                tf = f_flops / (f_ms * 1e-3) / 1e12
                gbs = f_bytes / (f_ms * 1e-3) / 1e9
                mf, hf = tf / MFMA_F32_PEAK_TF, gbs / HBM_PEAK_GBS
                roofline = {"bound": "mfma" if mf >= hf else "hbm",
                            "kernel": "rowgemm_gather_ks2_kernel<4,false> (tsgnn_gather_rowgemm_f32: aggregation + .W + bias + L2 normalise, K=N=%d)" % a.hidden,
                            "achieved": tf if mf >= hf else gbs, "peak": MFMA_F32_PEAK_TF if mf >= hf else HBM_PEAK_GBS,
                            "unit": "TFLOP/s" if mf >= hf else "GB/s", "frac": max(mf, hf),
                            "traffic": tr.get("dd_b32_gather_rowgemm_k128_n128", {}).get("traffic_bytes_per_launch"),
                            "flops_per_launch": f_flops, "bytes_per_launch": f_bytes, "us_per_launch": f_ms * 1e3,
                            "mfma": {"achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": mf},
                            "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hf},
                            "aggregation_standalone": standalone}
            out = {
                "metric": "graphs/sec fwd+bwd, DD batch=32 SAGE-3L h=128",
                "value": world * a.batch * a.steps / elapsed, "unit": "graphs/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%s-shaped synthetic graphs (avg %d nodes / %d edges, F_in=%d), batch=%d per GPU, "
                                       "GcnEncoderGraph (GraphSage 'base') %d layers h=%d, Nmax=%d, slot-BN, CE loss, "
                                       "clip 2.0 + Adam" % (a.shape, *synthetic.SHAPES[a.shape], a.batch, a.layers, a.hidden, a.nmax),
                           "global_batch": world * a.batch, "parallelism": "dp%d" % world,
                           "launch": "hipGraph replay" if use_graph else "eager",
                           "rows": int(g.n_rows), "edges_directed": int(g.nnz)},
                "roofline": roofline,
            }
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(hb, a.hidden, a.layers, a.cpu_steps, init_state)
        else:
            out["cpu_baseline"] = None
        if a.sweep:
            out["roofline"]["sweep"] = []
            with torch.cuda.stream(stream):
                for B in (32, 256, 2048, 16384):
                    hs = synthetic.host_batch(seed=100 + B, B=B, shape=a.shape, nmax=a.nmax)
                    gs, _, _ = synthetic.to_device(hs, dev)
                    ms, nb = aggregation_probe(gs, a.hidden, iters=50 if B > 2048 else 200)
                    out["roofline"]["sweep"].append({"graphs": B, "rows": int(gs.total_rows), "us_per_launch": ms * 1e3,
                                                     "achieved": nb / ms / 1e6, "frac": nb / ms / 1e6 / HBM_PEAK_GBS,
                                                     "kernel": aggregation_probe.kernel})
                    print("sweep B=%d rows=%d: %.2f us, %.0f GB/s (%.1f%% of 8 TB/s)" % (B, gs.n_rows, ms * 1e3, nb / ms / 1e6,
                                                                                       nb / ms / 1e6 / HBM_PEAK_GBS * 100), file=sys.stderr)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
