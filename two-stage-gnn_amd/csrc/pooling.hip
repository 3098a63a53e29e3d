// SAGPool path: per-graph top-k selection, gated gather, edge filtering with
// compaction, element-wise ReLU.  The reference calls torch_geometric's topk / filter_adj
// (Code/sag/layers.py:20-24); their arithmetic is restated from the documented PyG formulas
// (parity unpinned by the reference).
//
// top-k: one workgroup per graph sorts 64-bit keys (order-preserving score bits << 32 | ~index) in LDS
// with a bitonic network: descending score, ties -> smaller node index (deterministic, unlike
// torch.sort's unstable order).  Integer / byte work; nothing here is GEMM-shaped.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

constexpr int TOPK_MAX_SEG = 16384;     // 128 KiB of LDS keys (of 160 KiB per CU)
constexpr int TOPK_THREADS = 1024;

__global__ __launch_bounds__(TOPK_THREADS) void topk_segments_kernel(const float* __restrict__ score,
                                                                    const int* __restrict__ graph_ptr,
                                                                    const int* __restrict__ k_ptr, int* __restrict__ perm,
                                                                    int* __restrict__ new_id) {
  extern __shared__ unsigned long long keys[];
  const int b = blockIdx.x;
  const int g0 = graph_ptr[b];
  const int n = graph_ptr[b + 1] - g0;
  const int k0 = k_ptr[b], k = k_ptr[b + 1] - k0;
  if (n <= 0) return;
  int np = 1;
  while (np < n) np <<= 1;
  for (int i = threadIdx.x; i < np; i += TOPK_THREADS) {
    unsigned long long key = 0ull;                               // padding sorts last
    if (i < n) key = ((unsigned long long)f32_ordered(score[g0 + i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    keys[i] = key;
  }
  __syncthreads();
  for (int size = 2; size <= np; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (np >> 1); t += TOPK_THREADS) {
        const int lo = 2 * t - (t & (stride - 1));               // index with bit `stride` cleared
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);                    // descending blocks first -> overall descending
        const unsigned long long a = keys[lo], c = keys[hi];
        if ((a < c) == desc) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += TOPK_THREADS) {
    const int node = g0 + (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
    if (i < k) perm[k0 + i] = node;
    if (new_id != nullptr) new_id[node] = (i < k) ? k0 + i : -1;      // filter_adj's relabelling map, -1 = dropped
  }
}

// out[p,:] = x[perm[p],:] * gate(score[perm[p]]),  gate = tanh (Code/sag/layers.py:21)
__global__ __launch_bounds__(256) void gather_gate_fwd(const float* __restrict__ x, int64_t ldx, const float* __restrict__ score,
                                                       const int* __restrict__ perm, int64_t K, int F, int use_tanh,
                                                       float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= K) return;
  const int64_t r = perm[p];
  const float gt = use_tanh ? tanhf(score[r]) : score[r];
  for (int f = lane; f < F; f += 64) out[p * ldo + f] = x[r * ldx + f] * gt;
}
// dx[perm[p],:] = dout[p,:] * gate ; dscore[perm[p]] = gate' * <dout[p], x[perm[p]]>   (dx, dscore zero-initialised)
__global__ __launch_bounds__(256) void gather_gate_bwd(const float* __restrict__ x, int64_t ldx, const float* __restrict__ score,
                                                       const int* __restrict__ perm, int64_t K, int F, int use_tanh,
                                                       const float* __restrict__ dout, int64_t ldo, float* __restrict__ dx,
                                                       int64_t lddx, float* __restrict__ dscore) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= K) return;
  const int64_t r = perm[p];
  const float gt = use_tanh ? tanhf(score[r]) : score[r];
  float dot = 0.f;
  for (int f = lane; f < F; f += 64) {
    const float d = dout[p * ldo + f];
    dx[r * lddx + f] = d * gt;
    dot = fmaf(d, x[r * ldx + f], dot);
  }
  dot = wave_sum(dot);
  if (lane == 0) dscore[r] = dot * (use_tanh ? (1.f - gt * gt) : 1.f);
}

// filter_adj: new_id[perm[p]] = p; keep edges whose two ends are kept, relabelled, original order
__global__ void fill_i32(int* __restrict__ p, int64_t n, int v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void mark_kept_nodes(const int* __restrict__ perm, int64_t K, int* __restrict__ new_id) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K) new_id[perm[i]] = (int)i;
}
__global__ void edge_keep_flags(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E,
                                const int* __restrict__ new_id, int* __restrict__ flag) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < E) flag[e] = (new_id[src[e]] >= 0 && new_id[dst[e]] >= 0) ? 1 : 0;
}
__global__ void edge_compact(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E,
                             const int* __restrict__ new_id, const int* __restrict__ flag, const int* __restrict__ pos,
                             int64_t* __restrict__ out_src, int64_t* __restrict__ out_dst, int64_t* __restrict__ kept_eid) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E || !flag[e]) return;
  const int p = pos[e];
  out_src[p] = new_id[src[e]];
  out_dst[p] = new_id[dst[e]];
  if (kept_eid) kept_eid[p] = e;
}

__global__ void relu_fwd_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, int64_t n, float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// row softmax (DiffPool assignment, nn.Softmax(dim=-1), encoders.py:369) with optional row mask; one wave per row
// rows >= zero_from are written as zeros (the assignment rows of ghost nodes: "* embedding_mask", encoders.py:371)
__global__ __launch_bounds__(256) void row_softmax_fwd(const float* __restrict__ x, int64_t ldx, int64_t rows, int C,
                                                       float* __restrict__ y, int64_t ldy, int64_t zero_from) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  if (r >= zero_from) {
    for (int c = lane; c < C; c += 64) y[r * ldy + c] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[r * ldx + c]);
  m = wave_max(m);
  float d = 0.f;
  for (int c = lane; c < C; c += 64) d += expf(x[r * ldx + c] - m);
  d = wave_sum(d);
  for (int c = lane; c < C; c += 64) y[r * ldy + c] = expf(x[r * ldx + c] - m) / d;
}
__global__ __launch_bounds__(256) void row_softmax_bwd(const float* __restrict__ y, int64_t ldy, const float* __restrict__ dy,
                                                       int64_t lddy, int64_t rows, int C, float* __restrict__ dx, int64_t lddx,
                                                       int64_t zero_from) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  if (r >= zero_from) {
    for (int c = lane; c < C; c += 64) dx[r * lddx + c] = 0.f;
    return;
  }
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot = fmaf(y[r * ldy + c], dy[r * lddy + c], dot);
  dot = wave_sum(dot);
  for (int c = lane; c < C; c += 64) dx[r * lddx + c] = y[r * ldy + c] * (dy[r * lddy + c] - dot);
}

// PyG dense_diff_pool's assignment: s = softmax(logits) * m (row mask, nullable) AND the row's entropy term h = -sum_k s_k log(s_k + eps)
// in the same pass; hpart[block] = the block's four rows' sum (fixed order), summed by the host-side caller's one small reduction
__global__ __launch_bounds__(256) void row_softmax_ent_fwd(const float* __restrict__ x, int64_t ldx, int64_t rows, int C,
                                                           const float* __restrict__ mask, float eps, float* __restrict__ y, int64_t ldy,
                                                           float* __restrict__ hpart) {
  __shared__ float hs[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wid;
  float h = 0.f;
  if (r < rows) {
    const float mk = mask ? mask[r] : 1.f;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[r * ldx + c]);
    m = wave_max(m);
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d += expf(x[r * ldx + c] - m);
    d = wave_sum(d);
    for (int c = lane; c < C; c += 64) {
      const float sv = expf(x[r * ldx + c] - m) / d * mk;
      y[r * ldy + c] = sv;
      h -= sv * logf(sv + eps);
    }
    h = wave_sum(h);
  }
  if (lane == 0) hs[wid] = h;
  __syncthreads();
  if (threadIdx.x == 0) hpart[blockIdx.x] = (hs[0] + hs[1]) + (hs[2] + hs[3]);
}
// dlogits of the same: the gradient arriving at s (ds, nullable) plus g_ent[0] * d h / d s = -g (log(s + eps) + s / (s + eps)), through the
// mask and the softmax.  y = the MASKED s the forward wrote; the softmax itself is y / m on unmasked rows (masked rows: zero gradient).
__global__ __launch_bounds__(256) void row_softmax_ent_bwd(const float* __restrict__ y, int64_t ldy, const float* __restrict__ ds, int64_t ldds,
                                                           const float* __restrict__ mask, const float* __restrict__ g_ent, float g_scale,
                                                           float eps, int64_t rows, int C, float* __restrict__ dx, int64_t lddx) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float mk = mask ? mask[r] : 1.f;
  if (mk == 0.f) {
    for (int c = lane; c < C; c += 64) dx[r * lddx + c] = 0.f;
    return;
  }
  const float g = g_ent ? g_ent[0] * g_scale : 0.f;
  const float inv = 1.f / mk;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float sv = y[r * ldy + c];
    const float dv = (ds ? ds[r * ldds + c] : 0.f) - g * (logf(sv + eps) + sv / (sv + eps));
    dot = fmaf(sv * inv, dv * mk, dot);              // softmax value . gradient arriving at the softmax
  }
  dot = wave_sum(dot);
  for (int c = lane; c < C; c += 64) {
    const float sv = y[r * ldy + c];
    const float dv = (ds ? ds[r * ldds + c] : 0.f) - g * (logf(sv + eps) + sv / (sv + eps));
    dx[r * lddx + c] = sv * inv * (dv * mk - dot);
  }
}

}  // namespace

extern "C" {

/* PyG dense_diff_pool (north_star operator; no call site in the reference): s = softmax(x, -1) * mask (mask nullable,
 * one float per row) and hpart[ceil(rows / 4)] = partial sums of the rows' entropy terms -sum_k s log(s + eps) (their total / rows is
 * the operator's entropy loss) — one pass. */
int tsgnn_row_softmax_ent_fwd_f32(const float* x, int64_t ldx, int64_t rows, int C, const float* mask, float eps, float* y, int64_t ldy,
                                  float* hpart, tsgnn_stream_t stream) {
  if (!x || !y || !hpart || rows < 0 || C <= 0 || ldx < C || ldy < C) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  row_softmax_ent_fwd<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(x, ldx, rows, C, mask, eps, y, ldy, hpart);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}
/* its backward: dx = d logits from ds (gradient arriving at the masked s, nullable) and the entropy term's own gradient
 * g_ent[0] * g_scale * d(sum of the rows' entropy terms) / ds (g_ent: DEVICE scalar, nullable).  y = the forward's output. */
int tsgnn_row_softmax_ent_bwd_f32(const float* y, int64_t ldy, const float* ds, int64_t ldds, const float* mask, const float* g_ent,
                                  float g_scale, float eps, int64_t rows, int C, float* dx, int64_t lddx, tsgnn_stream_t stream) {
  if (!y || !dx || rows < 0 || C <= 0 || ldy < C || lddx < C || (ds && ldds < C)) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  row_softmax_ent_bwd<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(y, ldy, ds, ldds, mask, g_ent, g_scale, eps, rows, C, dx, lddx);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_topk_max_segment(void) { return TOPK_MAX_SEG; }

int tsgnn_topk_segments_f32(const float* score, const int* graph_ptr, const int* k_ptr, int B, int max_seg, int* perm,
                            int* new_id, tsgnn_stream_t stream) {
  if (!score || !graph_ptr || !k_ptr || !perm || B <= 0 || max_seg < 0) return TSGNN_EINVAL;
  if (max_seg > TOPK_MAX_SEG) return TSGNN_EUNSUPPORTED;
  if (max_seg == 0) return TSGNN_OK;
  int np = 1;
  while (np < max_seg) np <<= 1;
  const size_t lds = sizeof(unsigned long long) * (size_t)np;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_segments_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  topk_segments_kernel<<<B, TOPK_THREADS, lds, stream>>>(score, graph_ptr, k_ptr, perm, new_id);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gather_gate_fwd_f32(const float* x, int64_t ldx, const float* score, const int* perm, int64_t K, int F, int use_tanh,
                              float* out, int64_t ldo, tsgnn_stream_t stream) {
  if (!x || !score || !perm || !out || K < 0 || F <= 0 || ldx < F || ldo < F) return TSGNN_EINVAL;
  if (K == 0) return TSGNN_OK;
  gather_gate_fwd<<<(unsigned)ceil_div64(K, 4), 256, 0, stream>>>(x, ldx, score, perm, K, F, use_tanh, out, ldo);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gather_gate_bwd_f32(const float* x, int64_t ldx, const float* score, const int* perm, int64_t K, int F, int use_tanh,
                              const float* dout, int64_t ldo, float* dx, int64_t lddx, float* dscore, tsgnn_stream_t stream) {
  if (!x || !score || !perm || !dout || !dx || !dscore || K < 0 || F <= 0) return TSGNN_EINVAL;
  if (K == 0) return TSGNN_OK;
  gather_gate_bwd<<<(unsigned)ceil_div64(K, 4), 256, 0, stream>>>(x, ldx, score, perm, K, F, use_tanh, dout, ldo, dx, lddx, dscore);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* pass 1: new_id (N ints) + keep flags (E ints); caller scans flags -> pos; pass 2 compacts */
int tsgnn_filter_edges_mark(const int* perm, int64_t K, int64_t N, const int64_t* src, const int64_t* dst, int64_t E,
                            int* new_id, int* flag, tsgnn_stream_t stream) {
  if (!perm || !new_id || K < 0 || N < 0 || E < 0 || (E > 0 && (!src || !dst || !flag))) return TSGNN_EINVAL;
  if (N > 0) fill_i32<<<(unsigned)ceil_div64(N, 256), 256, 0, stream>>>(new_id, N, -1);
  if (K > 0) mark_kept_nodes<<<(unsigned)ceil_div64(K, 256), 256, 0, stream>>>(perm, K, new_id);
  if (E > 0) edge_keep_flags<<<(unsigned)ceil_div64(E, 256), 256, 0, stream>>>(src, dst, E, new_id, flag);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}
int tsgnn_filter_edges_compact(const int64_t* src, const int64_t* dst, int64_t E, const int* new_id, const int* flag,
                               const int* pos, int64_t* out_src, int64_t* out_dst, int64_t* kept_eid, tsgnn_stream_t stream) {
  if (E < 0 || (E > 0 && (!src || !dst || !new_id || !flag || !pos || !out_src || !out_dst))) return TSGNN_EINVAL;
  if (E == 0) return TSGNN_OK;
  edge_compact<<<(unsigned)ceil_div64(E, 256), 256, 0, stream>>>(src, dst, E, new_id, flag, pos, out_src, out_dst, kept_eid);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_relu_fwd_f32(const float* x, int64_t n, float* y, tsgnn_stream_t stream) {
  if (n < 0 || (n > 0 && (!x || !y))) return TSGNN_EINVAL;
  if (n == 0) return TSGNN_OK;
  relu_fwd_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(x, n, y);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}
int tsgnn_relu_bwd_f32(const float* y, const float* dy, int64_t n, float* dx, tsgnn_stream_t stream) {
  if (n < 0 || (n > 0 && (!y || !dy || !dx))) return TSGNN_EINVAL;
  if (n == 0) return TSGNN_OK;
  relu_bwd_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(y, dy, n, dx);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_row_softmax_fwd_f32(const float* x, int64_t ldx, int64_t rows, int C, float* y, int64_t ldy, tsgnn_stream_t stream) {
  return tsgnn_row_softmax_masked_fwd_f32(x, ldx, rows, C, y, ldy, rows, stream);
}
int tsgnn_row_softmax_masked_fwd_f32(const float* x, int64_t ldx, int64_t rows, int C, float* y, int64_t ldy, int64_t zero_from,
                                     tsgnn_stream_t stream) {
  if (!x || !y || rows < 0 || C <= 0 || ldx < C || ldy < C || zero_from < 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  row_softmax_fwd<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(x, ldx, rows, C, y, ldy, zero_from);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}
int tsgnn_row_softmax_bwd_f32(const float* y, int64_t ldy, const float* dy, int64_t lddy, int64_t rows, int C, float* dx,
                              int64_t lddx, tsgnn_stream_t stream) {
  return tsgnn_row_softmax_masked_bwd_f32(y, ldy, dy, lddy, rows, C, dx, lddx, rows, stream);
}
int tsgnn_row_softmax_masked_bwd_f32(const float* y, int64_t ldy, const float* dy, int64_t lddy, int64_t rows, int C, float* dx,
                                     int64_t lddx, int64_t zero_from, tsgnn_stream_t stream) {
  if (!y || !dy || !dx || rows < 0 || C <= 0 || zero_from < 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  row_softmax_bwd<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(y, ldy, dy, lddy, rows, C, dx, lddx, zero_from);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
