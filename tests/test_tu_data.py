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


def test_host_collate_device_layout():
    """tsgnn_host_collate_tu (host C code, no GPU): the capacity-padded batch in device layout — graph pointers (+ the dummy
    graph of the padding rows), slot counts, row maps, the fixed-width neighbour table with its tail, labels — against numpy on
    a dataset with rows of more than 16 neighbours, an empty graph, and the error paths"""
    from two_stage_gnn_amd import ingest
    from two_stage_gnn_amd.tu_data import TUDataset
    rng = np.random.default_rng(3)
    sizes = np.array([5, 0, 40, 12, 23, 1, 30])
    gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rows, cols = [], []
    for gi, n in enumerate(sizes):
        a = (rng.random((n, n)) < (0.7 if gi == 2 else 0.2))
        a = np.triu(a, 1); a = a | a.T
        for r in range(n):
            rows.append(np.flatnonzero(a[r]) + gp[gi])
    deg = np.array([len(r) for r in rows])
    assert deg.max() > 16
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate(rows).astype(np.int64)
    ds = TUDataset(gp, rowptr, col, np.array([1, 0, 1, 1, 0, 0, 1], dtype=np.int64), rng.integers(0, 9, size=int(gp[-1])).astype(np.int64), None, 9)
    B, nmax, cap, tcap = 4, 48, 96, 512
    ids = np.array([2, 1, 4, 0])
    off = ingest.layout(B, nmax, cap, 16, tcap)
    st = np.full(off[9], 12345, dtype=np.int32)
    n, nnz, ntail, largest = ingest.host_collate(ds, ids, B, nmax, cap, st, 16, tcap)
    bs = sizes[ids]
    assert n == bs.sum() and largest == 40 and nnz == sum(deg[gp[i]:gp[i + 1]].sum() for i in ids)
    g_ptr = st[off[0]:off[0] + B + 2]
    np.testing.assert_array_equal(g_ptr, np.concatenate([[0], np.cumsum(bs), [cap]]))
    np.testing.assert_array_equal(st[off[1]:off[1] + nmax], [(bs > s).sum() for s in range(nmax)])
    rg, rs = st[off[2]:off[2] + cap], st[off[3]:off[3] + cap]
    np.testing.assert_array_equal(rg[:n], np.repeat(np.arange(B), bs))
    np.testing.assert_array_equal(rs[:n], np.concatenate([np.arange(k) for k in bs]))
    assert (rg[n:] == B).all()                                              # padding rows: the dummy graph
    ell = st[off[4]:off[4] + (cap + nmax) * 16].reshape(cap + nmax, 16)
    tp, tc = st[off[5]:off[5] + cap + nmax + 1], st[off[6]:off[6] + tcap]
    row = 0
    for b, i in enumerate(ids):
        for r in range(gp[i], gp[i + 1]):
            want = col[rowptr[r]:rowptr[r + 1]] - gp[i] + g_ptr[b]
            got = np.concatenate([ell[row][ell[row] >= 0], tc[tp[row]:tp[row + 1]]])
            np.testing.assert_array_equal(got, want)
            assert (ell[row][:min(len(want), 16)] >= 0).all() and (ell[row][len(want):] == -1).all()
            row += 1
    assert row == n and ntail == tp[-1] == sum(max(0, d - 16) for i in ids for d in deg[gp[i]:gp[i + 1]])
    assert (ell[n:] == -1).all() and (tp[n:] == ntail).all()
    np.testing.assert_array_equal(st[off[7]:off[7] + n], np.concatenate([ds.node_label[gp[i]:gp[i + 1]] for i in ids]))
    np.testing.assert_array_equal(st[off[8]:off[8] + 2 * B].view(np.int64), ds.graph_label[ids])
    # capacity / size errors
    with pytest.raises(RuntimeError, match="not supported"):
        ingest.host_collate(ds, ids, B, nmax, 64, st, 16, tcap)               # 75 rows do not fit 64
    with pytest.raises(RuntimeError, match="not supported"):
        ingest.host_collate(ds, ids, B, 32, cap, st, 16, tcap)                # a 40-node graph with nmax 32
    with pytest.raises(RuntimeError, match="not supported"):
        ingest.host_collate(ds, ids, B, nmax, cap, st, 16, 2)                 # tail capacity


@pytest.mark.gpu
def test_capacity_padded_step_equals_exact_batch():
    """a mini-batch through the ingest slot (capacity-padded rows, dummy graph, fixed ghost-slot bound, one-hot features from the
    uploaded labels) trains exactly like the same graphs as an exact packed batch: loss, every gradient, two optimiser steps"""
    from two_stage_gnn_amd import dense_encoders as E, ingest
    from two_stage_gnn_amd.data_parallel import FlatTrainer
    from two_stage_gnn_amd.graph import GraphBatch
    dev = torch.device("cuda")
    ds = ingest.synthetic_dataset(seed=9, n_graphs=24, shape="DD", nmax=600)
    ids = np.array([3, 17, 5, 11, 20, 8])
    B, nmax, fin = len(ids), 600, ds.num_node_labels
    n = int(ds.sizes[ids].sum())
    nnz = int(sum(ds.rowptr[ds.graph_ptr[i + 1]] - ds.rowptr[ds.graph_ptr[i]] for i in ids))
    slot = ingest.CapacityBatch(B, nmax, (n + 200 + 31) // 32 * 32, nnz + 500, fin, dev, ghost_slots=int(ds.sizes.max()) + 1)
    slot.collate(ds, ids)
    slot.pull()
    torch.cuda.synchronize()
    # the expanded arrays against the exact batch built by the numpy collate
    assert (slot.rows, slot.edges) == (n, nnz)
    # the same graphs as an exact batch
    feats = ds.features("node-label")
    g, x, y = ds.collate(ids, nmax, feats, dev)
    torch.testing.assert_close(slot.x[:n], x[:n], rtol=0, atol=0)
    assert float(slot.x[n:].abs().sum()) == 0.0 and torch.equal(slot.label, y)
    ell_ref, W, tail_ref = g.ell()
    ell_slot = slot.g._ell[0].view(-1, 16)
    assert W <= 16 and tail_ref is None
    torch.testing.assert_close(ell_slot[:n, :W], ell_ref.view(-1, W)[:n], rtol=0, atol=0)
    assert bool((ell_slot[:n, W:] == -1).all()) and bool((ell_slot[n:] == -1).all())
    assert torch.equal(slot.g.graph_ptr[:B + 1], g.graph_ptr) and int(slot.g.graph_ptr[B + 1]) == slot.row_cap
    assert torch.equal(slot.g.slot_count[:nmax], g.slot_count)
    assert torch.equal(slot.g.row_graph[:n], g.row_graph[:n]) and torch.equal(slot.g.row_slot[:n], g.row_slot[:n])
    assert bool((slot.g.row_graph[n:] == B).all()) and int(slot.g._ell[2][0].abs().sum()) == 0
    # the slot-annotated copy of the table (operand of the fused slot batch-norm path) = what the exact batch builds on the host side
    es_ref = g.ell_slots()[0].view(-1, W)
    es_slot = slot.g.ell_slots()[0].view(-1, 16)
    torch.testing.assert_close(es_slot[:n, :W], es_ref[:n], rtol=0, atol=0)
    assert bool((es_slot[:n, W:] == -1).all()) and bool((es_slot[n:] == -1).all()) and bool((slot.g.row_slot[n:] == -1).all())

    class A:
        bias = True
    res = []
    for use_slot in (False, True):
        torch.manual_seed(2)
        m = E.GcnEncoderGraph(fin, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith("bias") and "conv" in k:
                    p.copy_(torch.randn_like(p) * 0.2)
        tr = FlatTrainer(m, lr=1e-2, clip=2.0)
        losses, grads = [], None
        for it in range(2):
            tr.zero_grad()
            if use_slot:
                slot.pull()                                    # (idempotent: the same staged batch again)
            loss = m.loss(m(slot.x, slot.g)[1], slot.label) if use_slot else m.loss(m(x, g)[1], y)
            tr.backward(loss)
            tr.gather_grads()
            if grads is None:
                grads = tr.flat_grad.clone()
            tr.apply()
            losses.append(float(loss.detach()))
        res.append((losses, grads, tr.flat_param.clone()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-6)
    scale = float(res[0][1].abs().max())
    torch.testing.assert_close(res[1][1], res[0][1], rtol=0, atol=2e-5 * scale)
    # parameters after two Adam steps: the two runs take different kernels for the slot batch-norm (the exact batch forms it inside
    # the products, the slot batch keeps the launches), so gradients agree to rounding — and Adam turns an entry whose gradient
    # is itself at rounding level into +-lr whatever its size.  Entries with a well-conditioned first-step gradient must agree
    # tightly; the others (a handful) may differ by what two steps can move them.
    well = res[0][1].abs() > 1e-4 * scale
    torch.testing.assert_close(res[1][2][well], res[0][2][well], rtol=1e-4, atol=2e-6)
    assert float((res[1][2] - res[0][2]).abs().max()) <= 2 * 2 * 1e-2
    assert int(((res[1][2] - res[0][2]).abs() > 1e-4 * res[0][2].abs() + 2e-6).sum()) <= 64


def test_host_collate_compact_layout():
    """tsgnn_host_collate_compact (host C code): the compact CSR batch the step's own graph pulls — header, graph pointers with
    the dummy graph, slot counts, shifted row pointers / columns, tail of the rows beyond 16 neighbours, labels; error paths"""
    from two_stage_gnn_amd import ingest
    from two_stage_gnn_amd.tu_data import TUDataset
    rng = np.random.default_rng(4)
    sizes = np.array([6, 0, 35, 14, 20])
    gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rows = []
    for gi, n in enumerate(sizes):
        a = (rng.random((n, n)) < (0.8 if gi == 2 else 0.25))
        a = np.triu(a, 1); a = a | a.T
        rows += [np.flatnonzero(a[r]) + gp[gi] for r in range(n)]
    deg = np.array([len(r) for r in rows])
    assert deg.max() > 16
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate(rows).astype(np.int64)
    ds = TUDataset(gp, rowptr, col, np.arange(5, dtype=np.int64) % 2, rng.integers(0, 9, size=int(gp[-1])).astype(np.int64), None, 9)
    B, nmax, cap, ecap, tcap = 3, 40, 96, 2048, 512
    ids = np.array([2, 4, 0])
    off = ingest.compact_layout(B, nmax, cap, ecap, tcap)
    st = np.full(off[9], 777, dtype=np.int32)
    n, nnz, ntail, largest = ingest.host_collate_compact(ds, ids, B, nmax, cap, ecap, st, 16, tcap)
    bs = sizes[ids]
    np.testing.assert_array_equal(st[off[0]:off[0] + 4], [n, nnz, ntail, largest])
    assert n == bs.sum() and largest == 35
    g_ptr = st[off[1]:off[1] + B + 2]
    np.testing.assert_array_equal(g_ptr, np.concatenate([[0], np.cumsum(bs), [cap]]))
    np.testing.assert_array_equal(st[off[2]:off[2] + nmax], [(bs > s).sum() for s in range(nmax)])
    np.testing.assert_array_equal(st[off[3]:off[3] + 2 * B].view(np.int64), ds.graph_label[ids])
    rp, nl, tp = st[off[4]:off[4] + n + 1], st[off[5]:off[5] + n], st[off[6]:off[6] + n + 1]
    cc, tc = st[off[7]:off[7] + nnz], st[off[8]:off[8] + ntail]
    row = 0
    for b, i in enumerate(ids):
        for r in range(gp[i], gp[i + 1]):
            want = col[rowptr[r]:rowptr[r + 1]] - gp[i] + g_ptr[b]
            np.testing.assert_array_equal(cc[rp[row]:rp[row + 1]], want)
            np.testing.assert_array_equal(tc[tp[row]:tp[row + 1]], want[16:])
            assert nl[row] == ds.node_label[r]
            row += 1
    assert row == n and rp[n] == nnz and tp[n] == ntail
    for kw in (dict(row_cap=32), dict(edge_cap=64), dict(tail_cap=1), dict(nmax=20)):
        args = dict(nmax=nmax, row_cap=cap, edge_cap=ecap, tail_cap=tcap); args.update(kw)
        with pytest.raises(RuntimeError, match="not supported"):
            ingest.host_collate_compact(ds, ids, B, args["nmax"], args["row_cap"], args["edge_cap"], st, 16, args["tail_cap"])


@pytest.mark.gpu
def test_ingest_acknowledges_pulled_batches():
    """the hand-shake that replaced the per-step HIP event: whoever collates a batch stamps it with a sequence number, the
    expand launch echoes the number of the batch it pulled into pinned host memory; inline collates and the worker pool wait for
    the echo of the batch the staging buffer holds before they overwrite it"""
    from two_stage_gnn_amd import ingest
    dev = torch.device("cuda")
    ds = ingest.synthetic_dataset(seed=3, n_graphs=12, shape="DD", nmax=400)
    ids = np.array([1, 5, 7])
    slot = ingest.CapacityBatch(3, 400, 1216, 8192, ds.num_node_labels, dev, ghost_slots=401)
    assert slot.seq == 0 and int(slot.ack[0]) == 0
    slot.collate(ds, ids)                                          # nothing staged before: no wait
    assert slot.seq == 1
    slot.pull(); slot.pull()                                       # the same staged batch twice: the same echo
    torch.cuda.synchronize()
    assert int(slot.ack[0]) == 1
    rows1 = slot.rows
    slot.collate(ds, ids[::-1].copy())                             # waits for the echo of batch 1 (already there)
    assert slot.seq == 2 and int(slot.ack[0]) == 1
    slot.pull()
    torch.cuda.synchronize()
    assert int(slot.ack[0]) == 2 and slot.rows == rows1
    # through the native workers: the job waits for the echo of batch 2, then stamps batch 3
    pool = ingest.CollatePool(1)
    slot.collate_async(pool, ds, np.array([0, 2, 4]))
    slot.collate_wait()
    assert slot.seq == 3 and int(slot.host[slot._seq_word]) == 3
    slot.pull()
    torch.cuda.synchronize()
    assert int(slot.ack[0]) == 3 and slot.rows == int(ds.sizes[[0, 2, 4]].sum())
    # a batch that was never enqueued for pulling may be overwritten at once
    slot.collate_async(pool, ds, ids)
    slot.collate_wait()
    assert slot.seq == 4 and int(slot.ack[0]) == 3
    pool.close()


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [0, 2])
def test_ingest_pipeline_riding_pull_equals_pull_at_the_head(workers):
    """the pipeline whose steps pull the NEXT batch as passengers of their first hidden-layer product (csrc/ingest_rider.h) trains
    exactly like the pipeline whose steps pull their own batch with their first launch: same parameters after the schedule, two
    runs in a row (the first batch of a run is pulled by a launch of its own), and the rider really is taken by a layer launch"""
    from two_stage_gnn_amd import dense_encoders as E, ingest, _native as nat
    from two_stage_gnn_amd.data_parallel import FlatTrainer
    dev = torch.device("cuda")
    ds = ingest.synthetic_dataset(seed=21, n_graphs=40, shape="DD", nmax=600)
    rng = np.random.default_rng(5)
    sched = [rng.choice(len(ds), size=6, replace=False) for _ in range(11)]

    class A:
        bias = True
    out = []
    for ride in (2, 1, 0):                                         # 2: pull and expansion ride; 1: the pull only; 0: neither
        torch.manual_seed(4)
        m = E.GcnEncoderGraph(ds.num_node_labels, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
        tr = FlatTrainer(m, lr=1e-2, clip=2.0, defer_loss=True)
        pipe = ingest.IngestPipeline(m, tr, ds, 6, 600, dev, sched, ride=ride)
        assert pipe.ride == bool(ride) and pipe.ride_expand == (ride == 2)
        if ride:                                                   # an eager step of position 0: who launches what?
            nat.trace = []
            try:
                with torch.cuda.stream(pipe.compute):
                    pipe.steps[0].loss_fn()
                names = [t[0] for t in nat.trace]
                flush = [t for t in nat.trace if t[0] == "ingest_flush_pull_rider"]
            finally:
                nat.trace = None
            torch.cuda.synchronize()
            assert not any(n in names for n in ("ingest_pull_f32", "ingest_pull_expand_ack_f32"))
            assert ("ingest_expand_ack_f32" in names) == (ride == 1) and (ride == 2 or names[0] == "ingest_expand_ack_f32")
            # (the carriers: the hidden layers' product launches and the head's forward — with or without the fused slot batch-norm)
            assert (names.count("sage_layer_fwd_f32") + names.count("sage_layer_fwd_ro_f32") + names.count("sage_layer_fwd_bn_f32") == 2
                    and ("packed_head_fwd_f32" in names or "packed_head_fwd_z_f32" in names))
            assert len(flush) == 1 and not flush[0][2].startswith("ingest_")      # nothing was left for launches of their own
        pipe.run(sched[:4], workers=workers)                       # (no synchronisation in between: the second run drains the
        pipe.run(sched[4:], workers=workers)                       # last replay's passengers itself)
        torch.cuda.synchronize()
        out.append(tr.flat_param.clone())
    assert torch.isfinite(out[0]).all()
    assert torch.equal(out[0], out[2]) and torch.equal(out[1], out[2])


@pytest.mark.gpu
@pytest.mark.parametrize("parts,skip", [(1, 0), (2, 0), (3, 0), (4, 0), (1, 2), (2, 2)])
def test_ingest_pull_rider_placements_train_alike(parts, skip, monkeypatch):
    """the next batch's PCIe pull dealt over `parts` carrier launches after `skip` carriers that go without (csrc/ingest_rider.h,
    tsgnn_ingest_arm_pull_rider_parts; the default of a 3-layer model is parts 2 / skip 1): whatever the placement — including shares
    that no carrier is left for and that the closing flush launches alone — the pipeline trains exactly like the one whose steps pull
    their own batch with their first launch"""
    from two_stage_gnn_amd import dense_encoders as E, ingest
    from two_stage_gnn_amd.data_parallel import FlatTrainer
    dev = torch.device("cuda")
    ds = ingest.synthetic_dataset(seed=22, n_graphs=30, shape="DD", nmax=600)
    rng = np.random.default_rng(6)
    sched = [rng.choice(len(ds), size=6, replace=False) for _ in range(7)]

    class A:
        bias = True
    out = []
    for ride in (2, 0):
        monkeypatch.setenv("TSGNN_INGEST_PULL_PARTS", str(parts))
        monkeypatch.setenv("TSGNN_INGEST_PULL_SKIP", str(skip))
        torch.manual_seed(4)
        m = E.GcnEncoderGraph(ds.num_node_labels, 128, 128, 2, 3, bn=True, args=A(), final_dim="number_classes").to(dev)
        tr = FlatTrainer(m, lr=1e-2, clip=2.0, defer_loss=True)
        pipe = ingest.IngestPipeline(m, tr, ds, 6, 600, dev, sched, ride=ride)
        if ride:
            assert (pipe.pull_parts, pipe.pull_skip) == (parts, skip)
        pipe.run(sched, workers=2)
        torch.cuda.synchronize()
        out.append(tr.flat_param.clone())
    assert torch.isfinite(out[0]).all() and torch.equal(out[0], out[1])


def test_malformed_integer_file_raises(tmp_path):
    """a token that is not an integer (float-formatted label, stray word) must raise like the reference's int() (load_data.py:24-60),
    not silently truncate the file"""
    from two_stage_gnn_amd import tu_data
    f = tmp_path / "X_graph_indicator.txt"
    f.write_text("1\n1\n2.0\n2\n")
    with pytest.raises(ValueError):
        tu_data._read_ints(str(f))
    f.write_text("1\n1\nabc\n2\n")
    with pytest.raises(ValueError):
        tu_data._read_ints(str(f))
    f.write_text("1, 2\n3,4\n")
    assert tu_data._read_ints(str(f)).tolist() == [1, 2, 3, 4]
