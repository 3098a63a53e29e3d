"""Seeded synthetic graph batches of the TU-dataset shapes the reference trains on (README.md:39-44),
built directly as CSR (no dense [B,Nmax,Nmax] tensor is ever materialised on the product path).

Generator: per graph  n_i = clamp(round(nbar*(1+0.35*N(0,1))), 8, Nmax);
e_i = ebar*n_i/nbar undirected edges drawn uniformly, symmetrised, no self loops, de-duplicated;
features N(0,1) fp32; labels uniform {0,1}.
"""
import numpy as np
import torch

from .graph import GraphBatch

SHAPES = {            # name: (avg nodes, avg undirected edges, input feature width)
    "DD": (269, 676, 89),
    "PROTEINS": (39, 73, 3),
    "MUTAG": (18, 20, 7),
    "IMDB-BINARY": (20, 97, 1),
}


def graph_sizes(rng, B, nbar, nmax):
    n = np.rint(nbar * (1.0 + 0.35 * rng.standard_normal(B))).astype(np.int64)
    return np.clip(n, min(8, nmax), nmax)


def random_edges(rng, n, e_target):
    """undirected simple graph on n nodes with ~e_target edges -> directed (src,dst) both ways, sorted by dst."""
    if n < 2 or e_target <= 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    m = int(min(e_target, n * (n - 1) // 2))
    u = rng.integers(0, n, size=int(m * 1.3) + 8)
    v = rng.integers(0, n, size=u.size)
    keep = u != v
    lo, hi = np.minimum(u[keep], v[keep]), np.maximum(u[keep], v[keep])
    code = np.unique(lo * n + hi)
    if code.size > m:
        code = rng.permutation(code)[:m]
    lo, hi = code // n, code % n
    src = np.concatenate([lo, hi])
    dst = np.concatenate([hi, lo])
    order = np.lexsort((src, dst))
    return src[order], dst[order]


def host_batch(seed, B, shape="DD", nmax=1000, nbar=None, ebar=None, fin=None):
    """-> dict of numpy arrays: sizes[B], rowptr[N+nmax+1] (ghost rows empty), col[E], x[N,fin], label[B]."""
    d_nbar, d_ebar, d_fin = SHAPES[shape]
    nbar = d_nbar if nbar is None else nbar
    ebar = d_ebar if ebar is None else ebar
    fin = d_fin if fin is None else fin
    rng = np.random.default_rng(seed)
    sizes = graph_sizes(rng, B, nbar, nmax)
    N = int(sizes.sum())
    deg = np.zeros(N + nmax, dtype=np.int64)
    cols = []
    off = 0
    for n in sizes:
        src, dst = random_edges(rng, int(n), int(round(ebar * n / nbar)))
        np.add.at(deg, off + dst, 1)
        cols.append(off + src)
        off += int(n)
    rowptr = np.zeros(N + nmax + 1, dtype=np.int32)
    np.cumsum(deg, out=rowptr[1:])
    col = (np.concatenate(cols) if cols else np.zeros(0, np.int64)).astype(np.int32)
    x = rng.standard_normal((N, fin)).astype(np.float32)
    label = rng.integers(0, 2, size=B).astype(np.int64)
    return {"sizes": sizes, "rowptr": rowptr, "col": col, "x": x, "label": label, "nmax": nmax, "fin": fin}


def tiled_batch(seed, B, shape="DD", nmax=1000, unique=256):
    """A B-graph batch made of ``unique`` generated graphs repeated B/unique times (structure only matters to the caller: the
    aggregation sweep of bench.py).  Every copy owns its rows, so nothing is shared between copies; generating 16,384 graphs
    one by one costs ~40 s of host time, tiling 256 of them well under a second."""
    if B <= unique:
        return host_batch(seed, B, shape, nmax)
    if B % unique:
        raise ValueError("B must be a multiple of %d" % unique)
    hb = host_batch(seed, unique, shape, nmax)
    reps = B // unique
    n = int(hb["sizes"].sum())
    rp = hb["rowptr"][: n + 1].astype(np.int64)
    e = int(rp[-1])
    sizes = np.tile(hb["sizes"], reps)
    col = (np.tile(hb["col"][:e].astype(np.int64), reps) + np.repeat(np.arange(reps, dtype=np.int64) * n, e)).astype(np.int32)
    rowptr = np.empty(reps * n + nmax + 1, dtype=np.int64)
    rowptr[: reps * n] = (np.tile(rp[:-1], reps) + np.repeat(np.arange(reps, dtype=np.int64) * e, n))
    rowptr[reps * n:] = reps * e
    if reps * e >= 2 ** 31 or reps * n >= 2 ** 31:
        raise ValueError("batch too large for int32 indices")
    return {"sizes": sizes, "rowptr": rowptr.astype(np.int32), "col": col, "x": None, "label": np.tile(hb["label"], reps),
            "nmax": nmax, "fin": hb["fin"]}


def structure_to_device(hb, device):
    """host batch -> GraphBatch only (no features)"""
    return GraphBatch.from_csr(torch.from_numpy(hb["rowptr"]).to(device), torch.from_numpy(hb["col"]).to(device), None,
                               hb["sizes"], hb["nmax"], assume_symmetric=True)


def to_device(hb, device):
    """host batch -> (GraphBatch, packed feature rows [N+nmax, ld] with 16-byte rows, labels)."""
    g = GraphBatch.from_csr(torch.from_numpy(hb["rowptr"]).to(device), torch.from_numpy(hb["col"]).to(device), None,
                            hb["sizes"], hb["nmax"], assume_symmetric=True)
    fin = hb["fin"]
    ld = (fin + 3) // 4 * 4
    x = torch.zeros(g.total_rows, ld, dtype=torch.float32, device=device)
    x[: g.n_rows, :fin] = torch.from_numpy(hb["x"]).to(device)
    return g, x, torch.from_numpy(hb["label"]).to(device)


def to_dense(hb):
    """host batch -> the reference's dense padded tensors (x[B,Nmax,F], adj[B,Nmax,Nmax]) on the CPU;
    used by tests and by bench.py's cpu_baseline leg only."""
    sizes, nmax, fin = hb["sizes"], hb["nmax"], hb["fin"]
    B = len(sizes)
    x = torch.zeros(B, nmax, fin)
    adj = torch.zeros(B, nmax, nmax)
    rp, col = hb["rowptr"], hb["col"]
    off = 0
    for b, n in enumerate(sizes):
        n = int(n)
        x[b, :n] = torch.from_numpy(hb["x"][off:off + n])
        for r in range(n):
            c = col[rp[off + r]:rp[off + r + 1]] - off
            adj[b, r, c] = 1.0
        off += n
    return x, adj


def aggregation_bytes(n_rows, nnz, feat, weighted=False):
    """Algorithmic HBM bytes of one aggregation pass:
    4NF (read X) + 4NF (write Y) + 4E (col) + 4(N+1) (rowptr) [+4E weights]."""
    return 8 * n_rows * feat + 4 * nnz + 4 * (n_rows + 1) + (4 * nnz if weighted else 0)


def to_pyg(hb, device, pad_features=True):
    """host batch -> what a torch_geometric DataLoader hands a model: (x [N, F], edge_index int64 [2, E] (row 0 = source, row 1 = target),
    batch int64 [N], label int64 [B]) on `device`.  pad_features: feature rows padded with zero columns to 16 bytes (what a CSR-native
    collate emits, like to_device() for the dense path); False: exactly [N, fin]."""
    sizes = hb["sizes"]
    n = int(sizes.sum())
    rp, col = hb["rowptr"][: n + 1], hb["col"]
    dst = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    ei = torch.from_numpy(np.stack([col[: int(rp[-1])].astype(np.int64), dst]))
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.from_numpy(np.asarray(sizes, dtype=np.int64)))
    fin = hb["fin"]
    ld = (fin + 3) // 4 * 4 if pad_features else fin
    x = torch.zeros(n, ld, dtype=torch.float32)
    x[:, :fin] = torch.from_numpy(hb["x"])
    return x.to(device), ei.to(device), batch.to(device), torch.from_numpy(hb["label"]).to(device)
