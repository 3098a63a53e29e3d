#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (same rule as gen_golden.py): wall time of the TU reader on a DD-sized synthetic TU
directory — two_stage_gnn_amd.tu_data.read_tu against the reference's own load_data.read_graphfile (load_data.py:12-126; networkx)
when /root/reference is present.  Host code on both sides: run it where the reference is (this container), not on the GPU box.

    python3 oracle/time_tu_reader.py [n_graphs=1178]
"""
import contextlib
import io
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF_DIR = "/root/reference/Code/sage+gat+diffpool"


def synth(seed, n_graphs, nbar=284, ebar=715, n_labels=89):
    """DD's shape (1,178 graphs, 284 nodes and 715 edges on average, 89 node labels): both directions of every edge as
    lines of _A.txt, as the real files have them"""
    rng = np.random.default_rng(seed)
    sizes = np.clip(rng.gamma(2.2, nbar / 2.2, n_graphs).astype(np.int64), 30, 5748)
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]]) + 1
    indic = np.repeat(np.arange(1, n_graphs + 1), sizes)
    edges = []
    for s, n in zip(starts, sizes):
        m = int(ebar * n / nbar)
        u = rng.integers(0, n, m)
        v = (u + 1 + rng.integers(0, min(n - 1, 12), m)) % n          # local structure, no self loops
        e = np.stack([u + s, v + s], 1)
        edges.append(np.concatenate([e, e[:, ::-1]]))
    edges = np.concatenate(edges)
    nlab = rng.integers(0, n_labels, int(sizes.sum()))
    glab = rng.integers(1, 3, n_graphs)
    return indic, edges, nlab, glab


def main():
    from gen_golden import write_tu_files
    from two_stage_gnn_amd import tu_data
    n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 1178
    indic, edges, nlab, glab = synth(7, n_graphs)
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        write_tu_files(tmp, "SYN", indic, edges, nlab, glab, None)
        mb = sum(os.path.getsize(os.path.join(tmp, "SYN", f)) for f in os.listdir(os.path.join(tmp, "SYN"))) / 1e6
        print("synthetic TU directory: %d graphs, %d nodes, %d edge lines, %.1f MB of text (written in %.1f s)"
              % (n_graphs, indic.size, edges.shape[0], mb, time.perf_counter() - t0))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ds = tu_data.read_tu(tmp, "SYN")
            best = min(best, time.perf_counter() - t0)
        print("two_stage_gnn_amd.tu_data.read_tu: %.2f s (best of 3) = %.0f graphs/s, %.1f MB/s of text; %d graphs, %d rows, %d CSR entries"
              % (best, n_graphs / best, mb / best, len(ds), int(ds.sizes.sum()), int(ds.rowptr[-1])))
        if os.path.isdir(REF_DIR):
            import networkx as nx
            sys.path.insert(0, REF_DIR)
            nx.__version__ = "3.4"                                   # load_data.py:112 parses it with float(); see gen_golden.gen_tu
            import load_data
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                graphs = load_data.read_graphfile(tmp, "SYN", max_nodes=None)
            tr = time.perf_counter() - t0
            print("reference load_data.read_graphfile (networkx %s): %.2f s = %.0f graphs/s -> %.1fx; same graph count: %s, same node total: %s"
                  % ("3.4.2", tr, n_graphs / tr, tr / best, len(graphs) == len(ds),
                     sum(G.number_of_nodes() for G in graphs) == int(ds.sizes.sum())))
        else:
            print("reference reader: /root/reference not present")


if __name__ == "__main__":
    main()
