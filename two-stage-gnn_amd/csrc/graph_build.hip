// Graph-structure ingest for the aggregation path:
//   * dense padded adjacency  adj[B,Nmax,Nmax] (graph_sampler.py:102-114)  ->  CSR over graph rows
//   * COO edge_index[2,E] int64 (PyG layout, Code/sag)                      ->  CSR grouped by target
//   * CSR transpose (for dX = A^T dY when A is not symmetric)
//   * row -> (graph, slot) maps, exclusive scan
// Integer/byte work, HBM-bound.  One wave (64 lanes) per adjacency row; ballot + popcount ranks
// give a deterministic ascending column order.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

// ---------------------------------------------------------------- exclusive scan (int32)
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;                     // items per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ int block_exclusive_scan(int v, int* total, int* lds /*>= 4+1 ints*/) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
    int s = lds[w];
    if (w < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// phase 1: per-tile sums
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_sums(const int* __restrict__ in, int64_t n,
                                                            int* __restrict__ tile_sum) {
  __shared__ int lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    int64_t i = base + (int64_t)threadIdx.x * SCAN_ITEMS + k;
    if (i < n) s += in[i];
  }
  int tot;
  block_exclusive_scan(s, &tot, lds);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}
// phase 2: one block scans the tile sums in place (exclusive) ; writes grand total to out[n]
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_offsets(int* __restrict__ tile_sum, int ntiles,
                                                               int* __restrict__ out_total) {
  __shared__ int lds[8];
  int carry = 0;
  for (int start = 0; start < ntiles; start += SCAN_BLOCK) {
    int i = start + threadIdx.x;
    int v = i < ntiles ? tile_sum[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, &tot, lds);
    if (i < ntiles) tile_sum[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *out_total = carry;
}
// phase 3: per-tile exclusive scan + tile offset
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply(const int* __restrict__ in, int64_t n,
                                                        const int* __restrict__ tile_off,
                                                        int* __restrict__ out) {
  __shared__ int lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  int tot;
  int ex = block_exclusive_scan(s, &tot, lds) + tile_off[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
}

// ---------------------------------------------------------------- row -> (graph, slot)
__global__ void row_maps_kernel(const int* __restrict__ graph_ptr, int B, int64_t n_rows,
                                int* __restrict__ row_graph, int* __restrict__ row_slot) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  int lo = 0, hi = B;                       // last b with graph_ptr[b] <= r
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (graph_ptr[mid] <= r) lo = mid; else hi = mid;
  }
  if (row_graph) row_graph[r] = lo;
  if (row_slot) row_slot[r] = (int)(r - graph_ptr[lo]);
}

// ---------------------------------------------------------------- dense adjacency -> CSR
// One wave per output row r = graph_ptr[b] + n.  Scans adj[b, n, 0:sz_b], sz_b = rows of graph b
// in the output layout (n_b when packed, Nmax when padded).  Non-zero test is `!= 0` so weighted
// (normalised) adjacencies keep their values.
template <bool FILL>
__global__ __launch_bounds__(256) void dense_adj_rows(const float* __restrict__ adj, int nmax,
                                                      const int* __restrict__ graph_ptr,
                                                      const int* __restrict__ row_graph,
                                                      int64_t n_rows, int* __restrict__ row_cnt,
                                                      const int* __restrict__ rowptr,
                                                      int* __restrict__ col, float* __restrict__ val) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int b = row_graph[r];
  const int g0 = graph_ptr[b];
  const int sz = graph_ptr[b + 1] - g0;
  const int n = (int)(r - g0);
  const float* __restrict__ arow = adj + ((int64_t)b * nmax + n) * nmax;
  int count = 0;
  int out = FILL ? rowptr[r] : 0;
  for (int j0 = 0; j0 < sz; j0 += 64) {
    const int j = j0 + lane;
    const float a = (j < sz) ? arow[j] : 0.f;
    const bool nz = (a != 0.f);
    const unsigned long long m = __ballot(nz);
    if (FILL) {
      if (nz) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        col[out + rank] = g0 + j;
        val[out + rank] = a;
      }
      out += __popcll(m);
    } else {
      count += __popcll(m);
    }
  }
  if (!FILL && lane == 0) row_cnt[r] = count;
}

// ---------------------------------------------------------------- COO -> CSR (grouped by key row)
__global__ void coo_hist(const int64_t* __restrict__ key, int64_t E, int64_t n_rows, int* __restrict__ cnt,
                         int* __restrict__ bad) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t k = key[e];
  if (k < 0 || k >= n_rows) { atomicOr(bad, 1); return; }
  atomicAdd(&cnt[k], 1);
}
// stable placement: edge e goes to rowptr[key] + (#edges e' < e with the same key).  Rank is
// obtained without atomics-order dependence by a per-row insertion afterwards: we place with an
// atomic cursor, then sort each row's slice by original edge id (rows are short).
__global__ void coo_place(const int64_t* __restrict__ key, const int64_t* __restrict__ other, int64_t E,
                          int64_t n_rows, const int* __restrict__ rowptr, int* __restrict__ cursor,
                          int* __restrict__ col, int* __restrict__ eid) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t k = key[e];
  if (k < 0 || k >= n_rows) return;
  int p = rowptr[k] + atomicAdd(&cursor[k], 1);
  col[p] = (int)other[e];
  eid[p] = (int)e;
}
__global__ void csr_sort_rows_by_eid(const int* __restrict__ rowptr, int64_t n_rows, int* __restrict__ col,
                                     int* __restrict__ eid) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int s = rowptr[r], t = rowptr[r + 1];
  for (int i = s + 1; i < t; ++i) {         // insertion sort (degree is small; deterministic result)
    int ke = eid[i], kc = col[i];
    int j = i - 1;
    while (j >= s && eid[j] > ke) { eid[j + 1] = eid[j]; col[j + 1] = col[j]; --j; }
    eid[j + 1] = ke; col[j + 1] = kc;
  }
}

// ---------------------------------------------------------------- CSR transpose
__global__ void csr_col_hist(const int* __restrict__ col, int64_t nnz, int* __restrict__ cnt) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz) atomicAdd(&cnt[col[e]], 1);
}
// one wave per source row; entries of one source row go to distinct target rows, so the only
// ordering freedom is between source rows -> fixed afterwards by sorting each target row by col.
__global__ void csr_transpose_place(const int* __restrict__ rowptr, const int* __restrict__ col,
                                    const float* __restrict__ val, int64_t n_rows,
                                    const int* __restrict__ rowptr_t, int* __restrict__ cursor,
                                    int* __restrict__ col_t, int* __restrict__ src_e) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
    int c = col[e];
    int p = rowptr_t[c] + atomicAdd(&cursor[c], 1);
    col_t[p] = (int)r;
    src_e[p] = e;
  }
}
__global__ void csr_sort_rows_by_col(const int* __restrict__ rowptr, int64_t n_rows, int* __restrict__ col,
                                     int* __restrict__ aux) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int s = rowptr[r], t = rowptr[r + 1];
  for (int i = s + 1; i < t; ++i) {
    int kc = col[i], ka = aux[i];
    int j = i - 1;
    while (j >= s && (col[j] > kc || (col[j] == kc && aux[j] > ka))) { col[j + 1] = col[j]; aux[j + 1] = aux[j]; --j; }
    col[j + 1] = kc; aux[j + 1] = ka;
  }
}
__global__ void gather_f32(const float* __restrict__ src, const int* __restrict__ idx, int64_t n,
                           float* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

}  // namespace

extern "C" {

int tsgnn_scan_workspace_ints(int64_t n, int64_t* ws_ints) {
  if (n < 0 || !ws_ints) return TSGNN_EINVAL;
  *ws_ints = ceil_div64(n > 0 ? n : 1, SCAN_TILE) + 1;
  return TSGNN_OK;
}

int tsgnn_exclusive_scan_i32(const int* in, int64_t n, int* out /*n+1*/, int* ws, hipStream_t stream) {
  if (n < 0 || (n > 0 && (!in || !out)) || !out || !ws) return TSGNN_EINVAL;
  if (n == 0) { (void)hipMemsetAsync(out, 0, sizeof(int), stream); TSGNN_CHECK_LAUNCH(); return TSGNN_OK; }
  const int ntiles = (int)ceil_div64(n, SCAN_TILE);
  scan_tile_sums<<<ntiles, SCAN_BLOCK, 0, stream>>>(in, n, ws);
  scan_tile_offsets<<<1, SCAN_BLOCK, 0, stream>>>(ws, ntiles, out + n);
  scan_apply<<<ntiles, SCAN_BLOCK, 0, stream>>>(in, n, ws, out);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_row_maps(const int* graph_ptr, int B, int64_t n_rows, int* row_graph, int* row_slot,
                   hipStream_t stream) {
  if (!graph_ptr || B <= 0 || n_rows < 0) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  row_maps_kernel<<<(unsigned)ceil_div64(n_rows, 256), 256, 0, stream>>>(graph_ptr, B, n_rows, row_graph, row_slot);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_dense_adj_count(const float* adj, int B, int nmax, const int* graph_ptr, const int* row_graph,
                          int64_t n_rows, int* row_cnt, hipStream_t stream) {
  if (!adj || !graph_ptr || !row_graph || !row_cnt || B <= 0 || nmax <= 0 || n_rows < 0) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  dense_adj_rows<false><<<(unsigned)ceil_div64(n_rows, 4), 256, 0, stream>>>(
      adj, nmax, graph_ptr, row_graph, n_rows, row_cnt, nullptr, nullptr, nullptr);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_dense_adj_fill(const float* adj, int B, int nmax, const int* graph_ptr, const int* row_graph,
                         int64_t n_rows, const int* rowptr, int* col, float* val, hipStream_t stream) {
  if (!adj || !graph_ptr || !row_graph || !rowptr || !col || !val || B <= 0 || nmax <= 0 || n_rows < 0)
    return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  dense_adj_rows<true><<<(unsigned)ceil_div64(n_rows, 4), 256, 0, stream>>>(
      adj, nmax, graph_ptr, row_graph, n_rows, nullptr, rowptr, col, val);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_coo_count(const int64_t* key, int64_t E, int64_t n_rows, int* cnt /*n_rows, zeroed here*/,
                    int* bad_flag /*1 int, zeroed here*/, hipStream_t stream) {
  if (E < 0 || n_rows < 0 || !cnt || !bad_flag || (E > 0 && !key)) return TSGNN_EINVAL;
  (void)hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)(n_rows > 0 ? n_rows : 1), stream);
  (void)hipMemsetAsync(bad_flag, 0, sizeof(int), stream);
  if (E > 0) coo_hist<<<(unsigned)ceil_div64(E, 256), 256, 0, stream>>>(key, E, n_rows, cnt, bad_flag);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_coo_fill(const int64_t* key, const int64_t* other, int64_t E, int64_t n_rows, const int* rowptr,
                   int* cursor /*n_rows scratch*/, int* col, int* eid, hipStream_t stream) {
  if (E < 0 || n_rows < 0 || !rowptr || !cursor || (E > 0 && (!key || !other || !col || !eid))) return TSGNN_EINVAL;
  if (E == 0 || n_rows == 0) return TSGNN_OK;
  (void)hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)n_rows, stream);
  coo_place<<<(unsigned)ceil_div64(E, 256), 256, 0, stream>>>(key, other, E, n_rows, rowptr, cursor, col, eid);
  csr_sort_rows_by_eid<<<(unsigned)ceil_div64(n_rows, 256), 256, 0, stream>>>(rowptr, n_rows, col, eid);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_csr_transpose(const int* rowptr, const int* col, const float* val, int64_t n_rows, int64_t n_cols,
                        int64_t nnz, int* rowptr_t /*n_cols+1*/, int* col_t, float* val_t /*nullable iff val null*/,
                        int* src_e /*nnz: source entry of each transposed entry*/, int* cnt_ws /*n_cols*/,
                        int* scan_ws, hipStream_t stream) {
  if (n_rows < 0 || n_cols < 0 || nnz < 0 || !rowptr || !rowptr_t || !cnt_ws || !scan_ws) return TSGNN_EINVAL;
  if (nnz > 0 && (!col || !col_t || !src_e)) return TSGNN_EINVAL;
  if ((val == nullptr) != (val_t == nullptr)) return TSGNN_EINVAL;
  (void)hipMemsetAsync(cnt_ws, 0, sizeof(int) * (size_t)(n_cols > 0 ? n_cols : 1), stream);
  if (nnz > 0) csr_col_hist<<<(unsigned)ceil_div64(nnz, 256), 256, 0, stream>>>(col, nnz, cnt_ws);
  int rc = tsgnn_exclusive_scan_i32(cnt_ws, n_cols, rowptr_t, scan_ws, stream);
  if (rc) return rc;
  if (nnz > 0 && n_rows > 0) {
    (void)hipMemsetAsync(cnt_ws, 0, sizeof(int) * (size_t)n_cols, stream);
    csr_transpose_place<<<(unsigned)ceil_div64(n_rows, 256), 256, 0, stream>>>(rowptr, col, val, n_rows, rowptr_t,
                                                                               cnt_ws, col_t, src_e);
    csr_sort_rows_by_col<<<(unsigned)ceil_div64(n_cols, 256), 256, 0, stream>>>(rowptr_t, n_cols, col_t, src_e);
    if (val) gather_f32<<<(unsigned)ceil_div64(nnz, 256), 256, 0, stream>>>(val, src_e, nnz, val_t);
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
