"""TU reader + CSR-native collate.  CPU: the vectorised reader against fixtures produced by the
REFERENCE's load_data.read_graphfile (load_data.py:12-126; oracle/gen_golden.py `tu` imports it in the build container and
stores its graphs as arrays in tests/golden/tu_*.npz), and against a networkx restatement of the same rules on further random
inputs.  GPU: collate -> encoder == dense path."""
import os

import numpy as np
import pytest
import torch


def write_tu(tmp, name, seed=0, n_graphs=7):
    rng = np.random.default_rng(seed)
    d = os.path.join(tmp, name)
    os.makedirs(d, exist_ok=True)
    indic, edges, nlab, glab = [], [], [], []
    nid = 1
    for g in range(1, n_graphs + 1):
        n = int(rng.integers(3, 12))
        ids = list(range(nid, nid + n))
        nid += n
        indic += [g] * n
        nlab += [int(rng.integers(1, 5)) for _ in ids]
        glab.append([7, -1, 3][g % 3])                       # non-consecutive labels, renumbered by first appearance
        m = int(rng.integers(1, 2 * n))
        for _ in range(m):
            u, v = rng.choice(ids, 2)
            edges.append((int(u), int(v)))                  # may contain duplicates and self loops; some nodes stay isolated
            if rng.random() < 0.5:
                edges.append((int(v), int(u)))
    pre = os.path.join(d, name)
    open(pre + "_graph_indicator.txt", "w").write("\n".join(map(str, indic)) + "\n")
    open(pre + "_graph_labels.txt", "w").write("\n".join(map(str, glab)) + "\n")
    open(pre + "_node_labels.txt", "w").write("\n".join(map(str, nlab)) + "\n")
    open(pre + "_A.txt", "w").write("\n".join("%d, %d" % e for e in edges) + "\n")
    return indic, edges, nlab, glab


def nx_reference(indic, edges, nlab, glab, max_nodes):
    """the reference's construction rules, restated with networkx (load_data.py:72-121)."""
    import networkx as nx
    label_vals = []
    for v in glab:
        if v not in label_vals:
            label_vals.append(v)
    adj_list = {i: [] for i in range(1, len(glab) + 1)}
    for e0, e1 in edges:
        adj_list[indic[e0 - 1]].append((e0, e1))
    out = []
    k = max(nlab)
    for i in range(1, len(glab) + 1):
        G = nx.from_edgelist(adj_list[i])
        if max_nodes is not None and G.number_of_nodes() > max_nodes:
            continue
        nodes = list(G.nodes)
        A = np.asarray(nx.to_numpy_array(G, nodelist=nodes)) if nodes else np.zeros((0, 0))
        onehot = np.zeros((len(nodes), k), dtype=np.float32)
        for r, u in enumerate(nodes):
            onehot[r, nlab[u - 1] - 1] = 1
        out.append((A, onehot, label_vals.index(glab[i - 1])))
    return out


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", ["labels_attrs", "plain"])
@pytest.mark.parametrize("mtag,max_nodes", [("all", None), ("max8", 8)])
def test_reader_matches_reference_fixture(tmp_path, tag, mtag, max_nodes):
    """read_tu == the reference's read_graphfile on the same files: kept graphs, node order (= BatchNorm slot), adjacency
    (duplicates collapsed, self loops on the diagonal, nodes without an edge dropped, a graph without edges kept with 0 nodes),
    one-hot node labels, node attributes, graph labels renumbered by first appearance."""
    from oracle.gen_golden import write_tu_files
    from two_stage_gnn_amd.tu_data import read_tu
    d = np.load(os.path.join(GOLDEN, "tu_%s.npz" % tag))
    write_tu_files(str(tmp_path), "TOY", d["indic"], d["edges"], d["nlab"] if "nlab" in d.files else None, d["glab"],
                   d["attrs"] if "attrs" in d.files else None)
    ds = read_tu(str(tmp_path), "TOY", max_nodes=max_nodes)
    sizes = d[mtag + ".sizes"]
    np.testing.assert_array_equal(ds.sizes, sizes)
    np.testing.assert_array_equal(ds.graph_label, d[mtag + ".labels"])
    flat, off = d[mtag + ".adj_flat"], 0
    for i, n in enumerate(sizes):
        n = int(n)
        np.testing.assert_array_equal(ds.dense(i, max(n, 1))[:n, :n].reshape(-1), (flat[off:off + n * n] > 0).astype(np.float32))
        off += n * n
    assert off == flat.size
    if "nlab" in d.files:
        np.testing.assert_array_equal(ds.features("node-label"), d[mtag + ".onehot"])
    else:
        assert ds.node_label is None
        np.testing.assert_array_equal(ds.features("node-label", input_dim=5), np.ones((int(sizes.sum()), 5), np.float32))
    if "attrs" in d.files:
        np.testing.assert_allclose(ds.features("node-feat"), d[mtag + ".feat"], rtol=0, atol=1e-6)
    else:
        assert ds.node_attr is None


@pytest.mark.parametrize("max_nodes", [None, 8])
def test_reader_matches_networkx_rules(tmp_path, max_nodes):
    from two_stage_gnn_amd.tu_data import read_tu
    raw = write_tu(str(tmp_path), "TOY", seed=3)
    ds = read_tu(str(tmp_path), "TOY", max_nodes=max_nodes)
    ref = nx_reference(*raw, max_nodes=max_nodes)
    assert len(ds) == len(ref)
    feats = ds.features("node-label")
    for i, (A, onehot, lab) in enumerate(ref):
        n = A.shape[0]
        assert ds.sizes[i] == n
        np.testing.assert_array_equal(ds.dense(i, max(n, 1))[:n, :n], (A > 0).astype(np.float32))
        np.testing.assert_array_equal(feats[ds.graph_ptr[i]:ds.graph_ptr[i + 1]], onehot)
        assert ds.graph_label[i] == lab


@pytest.mark.gpu
def test_collate_equals_dense_path(tmp_path):
    from oracle import dense_ref as R
    from two_stage_gnn_amd import dense_encoders as E
    from two_stage_gnn_amd.tu_data import read_tu
    write_tu(str(tmp_path), "TOY", seed=5, n_graphs=9)
    ds = read_tu(str(tmp_path), "TOY")
    nmax = ds.max_num_nodes()
    feats = ds.features("node-label")
    idx = [0, 3, 4, 8]
    g, x, y = ds.collate(idx, nmax, feats, torch.device("cuda"))
    # the reference's dense tensors of the same mini-batch
    B, F = len(idx), feats.shape[1]
    adj = torch.zeros(B, nmax, nmax); xd = torch.zeros(B, nmax, F)
    for k, i in enumerate(idx):
        n = int(ds.sizes[i])
        adj[k] = torch.from_numpy(ds.dense(i, nmax))
        xd[k, :n] = torch.from_numpy(feats[ds.graph_ptr[i]:ds.graph_ptr[i + 1]])

    class A:
        bias = True
    torch.manual_seed(0)
    m = E.GcnEncoderGraph(F, 8, 8, 3, 3, bn=True, args=A(), final_dim="number_classes").cuda()
    a, b = m(x, g)
    p = {k2: v.detach().cpu() for k2, v in m.state_dict().items()}
    a_ref, b_ref = R.gcn_encoder(p, xd, adj, bn=True, final_dim="number_classes")
    torch.testing.assert_close(a.cpu(), a_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.cpu(), b_ref, rtol=1e-4, atol=1e-4)
    assert y.tolist() == ds.graph_label[idx].tolist()


def test_sag_plan_sizes_follow_pyg_topk():
    """host-known structure of the sync-free SAGPool levels: k_b = ceil(ratio * n_b) in float32 as PyG's topk computes it,
    graph pointers and row -> graph maps per level (device-independent: built here on the CPU)"""
    import numpy as np
    import torch
    from oracle import pyg_ref as P
    from two_stage_gnn_amd.sag_stack import SagPlan
    sizes = [20, 1, 7, 136, 3, 64]
    for ratio in (0.5, 0.8, 0.25):
        plan = SagPlan.get(sizes, ratio, torch.device("cpu"), depth=3)
        cur = np.asarray(sizes)
        for lvl in plan.levels:
            assert lvl.N == int(cur.sum()) and lvl.B == len(sizes) and lvl.max_seg == int(cur.max())
            np.testing.assert_array_equal(lvl.gp.numpy(), np.concatenate([[0], np.cumsum(cur)]))
            np.testing.assert_array_equal(lvl.row_graph.numpy(), np.repeat(np.arange(len(sizes)), cur))
            batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.from_numpy(cur))
            perm = P.topk(torch.arange(int(cur.sum()), dtype=torch.float32), ratio, batch)     # the oracle's per-graph k
            cur = np.bincount(batch[perm].numpy(), minlength=len(sizes))
    assert SagPlan.get(sizes, 0.5, torch.device("cpu"), depth=3) is SagPlan.get(list(sizes), 0.5, torch.device("cpu"), depth=3)
