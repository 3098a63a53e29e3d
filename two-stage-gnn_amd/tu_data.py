"""TU-format dataset reader and CSR-native batch collate (SURVEY §8 "next" rows f2 / f1).

Replaces, without networkx and without ever building a dense [Nmax,Nmax] matrix:
  * ``load_data.read_graphfile`` (Code/sage+gat+diffpool/load_data.py:12-126): ``<name>_A.txt``,
    ``_graph_indicator.txt``, ``_graph_labels.txt``, optional ``_node_labels.txt`` / ``_node_attributes.txt``;
  * ``GraphSampler.__getitem__`` + default collate + the per-step ``.cuda()`` of ``adj[B,Nmax,Nmax]``
    (graph_sampler.py:102-114, train.py:114-119).

Bug-compatible details that change results downstream (node order = BatchNorm slot):
  * a graph is built from its EDGES only (``nx.from_edgelist``, load_data.py:90): nodes without an edge disappear;
  * node order = order of first appearance in the edge list, edge endpoints visited (e0, e1) per line (:112-121);
  * an edge belongs to the graph of its FIRST endpoint (:81); duplicates collapse, self loops stay (diagonal 1);
  * graph labels are renumbered in order of first appearance (:56-67); node labels are 1-based -> one-hot of
    ``max+1`` classes (:30-34, :98-102); graphs with more than ``max_nodes`` nodes are dropped (:91-92).
Host-side integer work (numpy); the device only sees the finished CSR.
"""
import os

import numpy as np
import torch

from .graph import GraphBatch


def _read_ints(path):
    """every integer of a text file (separators: commas and white space) — numpy's C tokenizer, not a Python loop per token
    (DD's _A.txt has 3.4 M of them)"""
    with open(path) as f:
        text = f.read().replace(",", " ")
    if not text.strip():
        return np.zeros(0, dtype=np.int64)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                   # numpy's text mode only WARNS when it stops at a token it cannot parse
        try:
            out = np.fromstring(text, dtype=np.int64, sep=" ")
        except (DeprecationWarning, ValueError) as e:
            raise ValueError("%s: not a list of integers (%s)" % (path, e))
    # ... and returns the prefix it did parse: a float-formatted or malformed file must fail loudly, as the reference's int() does
    # (load_data.py:24-60), not yield fewer nodes / edges / labels
    n_tok = len(text.split())
    if out.size != n_tok:
        raise ValueError("%s: %d of %d tokens are integers" % (path, out.size, n_tok))
    return out


class TUDataset:
    """All kept graphs as one CSR over re-labelled nodes + per-graph pointers."""

    def __init__(self, graph_ptr, rowptr, col, graph_label, node_label, node_attr, num_node_labels):
        self.graph_ptr = graph_ptr          # int64[G+1] node offsets
        self.rowptr = rowptr                # int64[N+1]
        self.col = col                      # int64[nnz] global (dataset) node ids
        self.graph_label = graph_label      # int64[G]
        self.node_label = node_label        # int64[N] or None
        self.node_attr = node_attr          # float32[N,d] or None
        self.num_node_labels = num_node_labels

    def __len__(self):
        return len(self.graph_ptr) - 1

    @property
    def sizes(self):
        return np.diff(self.graph_ptr)

    def max_num_nodes(self):
        return int(self.sizes.max())

    def features(self, kind="node-label", input_dim=10):
        """per-node feature rows as train.py:214-236 attaches them."""
        if kind == "node-label" and self.node_label is not None:
            f = np.zeros((len(self.node_label), self.num_node_labels), dtype=np.float32)
            f[np.arange(len(self.node_label)), self.node_label] = 1.0
            return f
        if kind == "node-feat" and self.node_attr is not None:
            return self.node_attr
        return np.ones((int(self.graph_ptr[-1]), input_dim), dtype=np.float32)       # constant features (train.py:233-236)

    def dense(self, i, nmax):
        """reference layout of ONE graph (GraphSampler.__getitem__): adj[Nmax,Nmax], used by tests only."""
        a, b = int(self.graph_ptr[i]), int(self.graph_ptr[i + 1])
        adj = np.zeros((nmax, nmax), dtype=np.float32)
        for r in range(a, b):
            adj[r - a, self.col[self.rowptr[r]:self.rowptr[r + 1]] - a] = 1.0
        return adj

    # ------------------------------------------------------------------ f1: collate straight to CSR
    def collate(self, idx, nmax, feats, device, labels=True):
        """-> (GraphBatch in the packed layout, feature rows [N+nmax, ld] on the device, labels).  ``idx``: graph ids of
        the mini-batch; ``feats``: per-node feature matrix (self.features(...))."""
        idx = np.asarray(idx, dtype=np.int64)
        a, b = self.graph_ptr[idx], self.graph_ptr[idx + 1]
        sizes = (b - a)
        if sizes.max() > nmax:
            raise ValueError("a graph has more nodes than nmax")
        N = int(sizes.sum())
        node_src = np.concatenate([np.arange(x, y) for x, y in zip(a, b)]) if N else np.zeros(0, np.int64)
        new_off = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(sizes, out=new_off[1:])
        deg = self.rowptr[node_src + 1] - self.rowptr[node_src]
        rowptr = np.zeros(N + nmax + 1, dtype=np.int32)
        np.cumsum(deg, out=rowptr[1:N + 1])
        rowptr[N + 1:] = rowptr[N]
        # columns: dataset id -> batch row id (same graph, so a constant shift per graph)
        shift = np.repeat(new_off[:-1] - a, sizes)
        nnz = int(rowptr[N])
        starts = self.rowptr[node_src]
        flat = (np.repeat(starts - np.concatenate([[0], np.cumsum(deg)[:-1]]), deg) + np.arange(nnz)) if nnz else np.zeros(0, np.int64)
        col = (self.col[flat] + np.repeat(shift, deg)).astype(np.int32) if nnz else np.zeros(0, np.int32)
        g = GraphBatch.from_csr(torch.from_numpy(rowptr).to(device), torch.from_numpy(col if nnz else np.zeros(1, np.int32)).to(device),
                                None, sizes, nmax, assume_symmetric=True)
        g.nnz = nnz
        F = feats.shape[1]
        ld = (F + 3) // 4 * 4
        x = torch.zeros(g.total_rows, ld, dtype=torch.float32, device=device)
        if N:
            x[:N, :F] = torch.from_numpy(np.ascontiguousarray(feats[node_src])).to(device)
        y = torch.from_numpy(self.graph_label[idx]).to(device) if labels else None
        return g, x, y


def read_tu(datadir, name, max_nodes=None):
    prefix = os.path.join(datadir, name, name)
    indic = _read_ints(prefix + "_graph_indicator.txt")                    # 1-based graph id of node i (1-based)
    raw_labels = _read_ints(prefix + "_graph_labels.txt")
    n_graphs = len(raw_labels)
    # graph labels renumbered by first appearance
    _, first = np.unique(raw_labels, return_index=True)
    order_vals = raw_labels[np.sort(first)]
    remap = {int(v): i for i, v in enumerate(order_vals)}
    graph_label_all = np.array([remap[int(v)] for v in raw_labels], dtype=np.int64)
    node_label_all = None
    num_node_labels = 0
    if os.path.exists(prefix + "_node_labels.txt"):
        node_label_all = _read_ints(prefix + "_node_labels.txt") - 1
        num_node_labels = int(node_label_all.max()) + 1
    node_attr_all = None
    if os.path.exists(prefix + "_node_attributes.txt"):
        node_attr_all = np.loadtxt(prefix + "_node_attributes.txt", delimiter=",", dtype=np.float32, ndmin=2)
    edges = _read_ints(prefix + "_A.txt").reshape(-1, 2)
    e_graph = indic[edges[:, 0] - 1]                                      # graph of the FIRST endpoint
    order = np.argsort(e_graph, kind="stable")
    edges, e_graph = edges[order], e_graph[order]
    bounds = np.searchsorted(e_graph, np.arange(1, n_graphs + 2))
    graph_ptr = [0]
    rowptrs, cols, keep_label, nl, na = [], [], [], [], []
    node_total = 0
    nnz_total = 0
    for gi in range(n_graphs):
        e = edges[bounds[gi]:bounds[gi + 1]]
        if len(e) == 0:
            # nx.from_edgelist([]) is an empty graph: 0 nodes (kept by the reference unless filtered)
            nodes = np.zeros(0, np.int64)
        else:
            flat = e.reshape(-1)                                           # e0, e1, e0, e1, ... = nx insertion order
            _, first_pos, inverse = np.unique(flat, return_index=True, return_inverse=True)
            by_first = np.argsort(first_pos, kind="stable")                # sorted-unique index of the k-th node to appear
            nodes = flat[first_pos[by_first]]
        n = len(nodes)
        if max_nodes is not None and n > max_nodes:
            continue
        if n:
            rank = np.empty(n, dtype=np.int64)                             # sorted-unique index -> order of first appearance
            rank[by_first] = np.arange(n)
            local = rank[inverse.reshape(-1)].reshape(-1, 2)               # both endpoints of every edge line, relabelled
            u, v = local[:, 0], local[:, 1]
            uu = np.concatenate([u, v]); vv = np.concatenate([v, u])       # undirected
            code = np.unique(uu * n + vv)                                  # dedup (self loops appear once)
            r, c = code // n, code % n
            deg = np.bincount(r, minlength=n)
            rp = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(deg, out=rp[1:])
            rowptrs.append(rp[1:] + nnz_total)
            cols.append(c + node_total)
            nnz_total += len(c)
            if node_label_all is not None:
                nl.append(node_label_all[nodes - 1])
            if node_attr_all is not None:
                na.append(node_attr_all[nodes - 1])
        node_total += n
        graph_ptr.append(node_total)
        keep_label.append(graph_label_all[gi])
    rowptr = np.concatenate([[0]] + rowptrs).astype(np.int64) if rowptrs else np.zeros(1, np.int64)
    col = np.concatenate(cols).astype(np.int64) if cols else np.zeros(0, np.int64)
    return TUDataset(np.asarray(graph_ptr, dtype=np.int64), rowptr, col, np.asarray(keep_label, dtype=np.int64),
                     np.concatenate(nl) if nl else None, np.concatenate(na).astype(np.float32) if na else None, num_node_labels)
