// SAGPool level, sync-free: GCN propagation with the symmetric normalisation generated from
// per-row coefficients, per-graph top-k selection output consumed on the device, CSR -> CSR edge filtering, gated
// gather, max||mean readout, and the explicit backward of all of it.
//
// Replaces, per level of Code/sag/network.py:33-44,
//     x = relu(conv(x, edge_index)); x, edge_index, _, batch, _ = pool(x, edge_index, None, batch)   (layers.py:14-25)
//     x_l = cat([gmp(x, batch), gap(x, batch)])
// where PyG's topk / filter_adj return tensors whose SIZES depend on the data (k is host-computable from the graph
// sizes; E' is not).  Here the filtered adjacency stays a CSR whose row count is host-known and whose entry count
// lives in rowptr'[K] on the device, so a level needs no host round trip and the step is hipGraph-capturable.
//
// GCN normalisation (PyG gcn_norm, unit edge weights): A^ = D^-1/2 (A + I) D^-1/2 is never materialised per entry:
//     (A^ x)_i = dinv_i * sum_{j in N(i)} dinv_j x_j + self_w_i x_i,   self_w_i = dinv_i^2 (0 if i has a self loop)
// HBM/L2-bound gathers: one lane group of G lanes per row, float4 per lane, neighbour indices and their dinv fetched
// with one coalesced load and broadcast with wave shuffles.  No atomics on data (bitwise reproducible).
#include "common.h"
#include <cstdlib>
#include "../../include/tsgnn.h"
#include "du_reduce_body.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ void fma4(float4& a, float w, float4 v) {
  a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

// ---------------------------------------------------------------- dinv / self_w of gcn_norm (unit weights)
__global__ void gcn_coef_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int64_t n,
                                float* __restrict__ dinv, float* __restrict__ self_w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  bool has_self = false;                                   // add_remaining_self_loops keeps an existing self loop
  for (int e = e0; e < e1; ++e) has_self |= (col[e] == (int)i);
  const float d = (float)(e1 - e0) + (has_self ? 0.f : 1.f);
  const float di = 1.0f / sqrtf(d);
  dinv[i] = di;
  self_w[i] = has_self ? 0.f : di * di;
}

struct PropArgs {
  const int* rowptr;
  const int* rowend;       // nullable: row r = entries [rowptr[r], rowend[r]) (graph segments with slack between them, written by
                           // the per-graph pooling kernel); NULL: the standard [rowptr[r], rowptr[r+1])
  const int* col;
  const float* dinv;
  const float* self_w;
  const float* x;
  int64_t ldx;
  const float* bias;       // nullable, added after the aggregation
  const float* w_dot;      // nullable: t[row] = <out[row], w_dot> (+ *dot_bias)
  const float* dot_bias;   // nullable (1 float)
  float* y;                // nullable: skip the store (dot-only mode)
  int64_t ldy;
  float* t;
  int64_t n_rows;
  int feat;
  int relu_in;             // gather relu(x) instead of x
  // narrow inputs only (feat <= 8, wave-per-row kernel): the GCNConv transform that follows the aggregation in the same launch,
  // lin_y[row, :lin_n] = agg[row, :feat] . lin_w[feat, lin_n] + lin_b  (network.py:34 at num_features = 1: an outer product)
  const float* lin_w; int64_t ldlw; const float* lin_b; float* lin_y; int64_t ldly; int lin_n;
  // mean aggregation with the coefficients taken from the row lengths (no coefficient arrays; vec4 / generic kernels only):
  //   1: out_i = (1 / max(len_i, 1)) sum_j x_j            2 (its transpose on a symmetric list): out_i = sum_j x_j / max(len_j, 1)
  // and, with either, the self term read from a SECOND buffer with weight 1: out_i += xself[i]  (NULL: no self term)
  int scale_mode;
  const float* xself; int64_t ldxs;
};
__device__ __forceinline__ float prop_col(const PropArgs& a, int j) {
  if (a.scale_mode == 0) return a.dinv[j];
  if (a.scale_mode == 1) return 1.f;
  const int len = (a.rowend ? a.rowend[j] : a.rowptr[j + 1]) - a.rowptr[j];
  return 1.0f / (float)max(len, 1);
}
__device__ __forceinline__ float prop_row(const PropArgs& a, int64_t row, int e0, int e1) {
  if (a.scale_mode == 0) return a.dinv[row];
  return a.scale_mode == 1 ? 1.0f / (float)max(e1 - e0, 1) : 1.f;
}

template <int G>
__global__ __launch_bounds__(256) void gcn_propagate_vec4(PropArgs a, unsigned nblk) {
  constexpr int RPB = 256 / G;
  const unsigned lb = xcd_remap(blockIdx.x, nblk);
  const int lig = threadIdx.x & (G - 1);
  const int64_t row = (int64_t)lb * RPB + threadIdx.x / G;
  if (row >= a.n_rows) return;                              // group-uniform
  const int nvec = a.feat >> 2;
  const bool live = lig < nvec;
  const int64_t co = live ? 4 * lig : 0;
  const int e0 = a.rowptr[row], e1 = a.rowend ? a.rowend[row] : a.rowptr[row + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int eb = e0; eb < e1; eb += G) {
    const int me = eb + lig;
    const int cj = (me < e1) ? a.col[me] : 0;
    const float dj = (me < e1) ? prop_col(a, cj) : 0.f;
    const int cnt = min(G, e1 - eb);
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {
      const int j0 = __shfl(cj, k, G), j1 = __shfl(cj, k + 1, G), j2 = __shfl(cj, k + 2, G), j3 = __shfl(cj, k + 3, G);
      float4 v0 = ld4(a.x + (int64_t)j0 * a.ldx + co);
      float4 v1 = ld4(a.x + (int64_t)j1 * a.ldx + co);
      float4 v2 = ld4(a.x + (int64_t)j2 * a.ldx + co);
      float4 v3 = ld4(a.x + (int64_t)j3 * a.ldx + co);
      if (a.relu_in) { v0 = relu4(v0); v1 = relu4(v1); v2 = relu4(v2); v3 = relu4(v3); }
      fma4(acc, __shfl(dj, k, G), v0); fma4(acc, __shfl(dj, k + 1, G), v1);
      fma4(acc, __shfl(dj, k + 2, G), v2); fma4(acc, __shfl(dj, k + 3, G), v3);
    }
    for (; k < cnt; ++k) {
      const int j = __shfl(cj, k, G);
      float4 v = ld4(a.x + (int64_t)j * a.ldx + co);
      if (a.relu_in) v = relu4(v);
      fma4(acc, __shfl(dj, k, G), v);
    }
  }
  const float di = prop_row(a, row, e0, e1);
  float sw;
  float4 xs;
  if (a.scale_mode == 0) {
    sw = a.self_w[row];
    xs = ld4(a.x + row * a.ldx + co);
    if (a.relu_in) xs = relu4(xs);
  } else {
    sw = a.xself ? 1.f : 0.f;
    xs = a.xself ? ld4(a.xself + row * a.ldxs + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 o = make_float4(fmaf(di, acc.x, sw * xs.x), fmaf(di, acc.y, sw * xs.y), fmaf(di, acc.z, sw * xs.z), fmaf(di, acc.w, sw * xs.w));
  if (a.bias != nullptr) {
    const float4 b = ld4(a.bias + co);
    o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
  }
  if (a.y != nullptr && live) *reinterpret_cast<float4*>(a.y + row * a.ldy + co) = o;
  if (a.w_dot != nullptr) {
    float d = live ? dot4(o, ld4(a.w_dot + co)) : 0.f;
    d = group_sum<G>(d);
    if (lig == 0) a.t[row] = d + (a.dot_bias ? a.dot_bias[0] : 0.f);
  }
}

// Row-batched variant for batches that no longer fit the caches (>= 262,144 rows): a lane group owns RB consecutive rows,
// ONE rowptr fetch and ONE coalesced index (+ dinv) fetch cover all their neighbours and the feature rows are gathered eight
// at a time — ~RB x fewer dependent hops per row (the one-row-per-group kernel is latency- rather than bandwidth-bound there:
// 0.33 of the HBM spec at 2,048 DD graphs).  Per-row summation order is unchanged (bitwise equal to gcn_propagate_vec4).
template <int G, int RB>
__global__ __launch_bounds__(256) void gcn_propagate_vec4_rb(PropArgs a, unsigned nblk) {
  constexpr int GROUPS = 256 / G;
  const unsigned lb = xcd_remap(blockIdx.x, nblk);
  const int lig = threadIdx.x & (G - 1);
  const int64_t row0 = ((int64_t)lb * GROUPS + threadIdx.x / G) * RB;
  if (row0 >= a.n_rows) return;
  const int nvec = a.feat >> 2;
  const bool live = lig < nvec;
  const int64_t co = live ? 4 * lig : 0;
  const int64_t rr = min(row0 + lig, a.n_rows);
  const int rp = (lig <= RB) ? a.rowptr[rr] : 0;
  int eb[RB + 1];
#pragma unroll
  for (int k = 0; k <= RB; ++k) eb[k] = __shfl(rp, k, G);
  float4 acc[RB];
#pragma unroll
  for (int k = 0; k < RB; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = eb[0]; base < eb[RB]; base += G) {
    const int me = base + lig;
    const int cj = (me < eb[RB]) ? a.col[me] : 0;
    const float dj = (me < eb[RB]) ? prop_col(a, cj) : 0.f;
    const int cnt = min(G, eb[RB] - base);
    for (int k = 0; k < cnt; k += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = __shfl(cj, (k + u) & (G - 1), G);
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + u < cnt) v[u] = ld4(a.x + (int64_t)j * a.ldx + co);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k + u < cnt) {
          const int e = base + k + u;
          const float w = __shfl(dj, (k + u) & (G - 1), G);
          const float4 vv = a.relu_in ? relu4(v[u]) : v[u];
#pragma unroll
          for (int r = 0; r < RB; ++r)
            if (e >= eb[r] && e < eb[r + 1]) fma4(acc[r], w, vv);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int64_t row = row0 + r;
    if (row >= a.n_rows) break;                          // group-uniform
    const float di = a.dinv[row], sw = a.self_w[row];
    float4 xs = ld4(a.x + row * a.ldx + co);
    if (a.relu_in) xs = relu4(xs);
    float4 o = make_float4(fmaf(di, acc[r].x, sw * xs.x), fmaf(di, acc[r].y, sw * xs.y), fmaf(di, acc[r].z, sw * xs.z),
                           fmaf(di, acc[r].w, sw * xs.w));
    if (a.bias != nullptr) {
      const float4 b = ld4(a.bias + co);
      o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    }
    if (a.y != nullptr && live) *reinterpret_cast<float4*>(a.y + row * a.ldy + co) = o;
    if (a.w_dot != nullptr) {
      float d = live ? dot4(o, ld4(a.w_dot + co)) : 0.f;
      d = group_sum<G>(d);
      if (lig == 0) a.t[row] = d + (a.dot_bias ? a.dot_bias[0] : 0.f);
    }
  }
}

// any feature width / leading dimension (F = 1 input of IMDB-B, the raw 89-wide DD labels): one wave per row
__global__ __launch_bounds__(256) void gcn_propagate_generic(PropArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n_rows) return;
  const int e0 = a.rowptr[row], e1 = a.rowend ? a.rowend[row] : a.rowptr[row + 1];
  const float di = prop_row(a, row, e0, e1);
  const float sw = a.scale_mode == 0 ? a.self_w[row] : (a.xself ? 1.f : 0.f);
  const float* xself = a.scale_mode == 0 ? a.x : (a.xself ? a.xself : a.x);      // self rows (weight 0 without xself in the mean modes)
  const int64_t ldxs = (a.scale_mode != 0 && a.xself) ? a.ldxs : a.ldx;
  const bool self_relu = a.relu_in && a.scale_mode == 0;
  float dot = 0.f;
  if (a.feat <= 8) {
    // narrow inputs (the one-column degree / constant feature): lanes over the ENTRIES — with lanes over features one lane
    // walked the neighbours alone, a dependent round trip each
    float ov[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      ov[f] = 0.f;
      if (f < a.feat) {
        float acc = 0.f;
        for (int e = e0 + lane; e < e1; e += 64) {
          const int j = a.col[e];
          float v = a.x[(int64_t)j * a.ldx + f];
          if (a.relu_in) v = fmaxf(v, 0.f);
          acc = fmaf(prop_col(a, j), v, acc);
        }
        acc = wave_sum(acc);
        float xs = xself[row * ldxs + f];
        if (self_relu) xs = fmaxf(xs, 0.f);
        float o = fmaf(di, acc, sw * xs);
        if (a.bias != nullptr) o += a.bias[f];
        if (a.y != nullptr && lane == 0) a.y[row * a.ldy + f] = o;
        if (a.w_dot != nullptr) dot = fmaf(o, a.w_dot[f], dot);
        ov[f] = o;
      }
    }
    if (a.w_dot != nullptr && lane == 0) a.t[row] = dot + (a.dot_bias ? a.dot_bias[0] : 0.f);
    if (a.lin_y != nullptr) {                                // every lane holds the row's aggregate: lanes over the output columns
      for (int c = lane; c < a.lin_n; c += 64) {
        float v = a.lin_b ? a.lin_b[c] : 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f)
          if (f < a.feat) v = fmaf(ov[f], a.lin_w[(int64_t)f * a.ldlw + c], v);
        a.lin_y[row * a.ldly + c] = v;
      }
    }
    return;
  }
  for (int fb = 0; fb < a.feat; fb += 64) {                // wave-uniform trip count
    const int f = fb + lane;
    const bool live = f < a.feat;
    const int fo = live ? f : 0;
    float acc = 0.f;
    for (int eb = e0; eb < e1; eb += 64) {
      const int me = eb + lane;
      const int cj = (me < e1) ? a.col[me] : 0;
      const float dj = (me < e1) ? prop_col(a, cj) : 0.f;
      const int cnt = min(64, e1 - eb);
      for (int k = 0; k < cnt; ++k) {
        const int j = __shfl(cj, k, 64);
        float v = a.x[(int64_t)j * a.ldx + fo];
        if (a.relu_in) v = fmaxf(v, 0.f);
        acc = fmaf(__shfl(dj, k, 64), v, acc);
      }
    }
    float xs = xself[row * ldxs + fo];
    if (self_relu) xs = fmaxf(xs, 0.f);
    float o = fmaf(di, acc, sw * xs);
    if (a.bias != nullptr) o += a.bias[fo];
    if (a.y != nullptr && live) a.y[row * a.ldy + f] = o;
    if (a.w_dot != nullptr && live) dot = fmaf(o, a.w_dot[f], dot);
  }
  if (a.w_dot != nullptr) {
    dot = wave_sum(dot);
    if (lane == 0) a.t[row] = dot + (a.dot_bias ? a.dot_bias[0] : 0.f);
  }
}

// narrow inputs of a large batch (the one-column degree / constant feature of the IMDB sets, <= 8 columns): one THREAD per
// row — a wave per row leaves 63 of 64 lanes idle there (8,192 IMDB-B graphs: 90 us for a 160k x 1 propagate)
constexpr int PROP_NARROW_MAX = 8;
constexpr int64_t PROP_NARROW_MIN_ROWS = 32768;
__global__ __launch_bounds__(256) void gcn_propagate_narrow(PropArgs a) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= a.n_rows) return;
  const int e0 = a.rowptr[row], e1 = a.rowend ? a.rowend[row] : a.rowptr[row + 1];
  float acc[PROP_NARROW_MAX];
#pragma unroll
  for (int f = 0; f < PROP_NARROW_MAX; ++f) acc[f] = 0.f;
#pragma unroll 4
  for (int e = e0; e < e1; ++e) {
    const int j = a.col[e];
    const float dj = prop_col(a, j);
    const float* xr = a.x + (int64_t)j * a.ldx;
#pragma unroll
    for (int f = 0; f < PROP_NARROW_MAX; ++f) {
      if (f < a.feat) {
        float v = xr[f];
        if (a.relu_in) v = fmaxf(v, 0.f);
        acc[f] = fmaf(dj, v, acc[f]);
      }
    }
  }
  const float di = a.dinv[row], sw = a.self_w[row];
  float dot = 0.f;
#pragma unroll
  for (int f = 0; f < PROP_NARROW_MAX; ++f) {
    if (f < a.feat) {
      float xs = a.x[row * a.ldx + f];
      if (a.relu_in) xs = fmaxf(xs, 0.f);
      float o = fmaf(di, acc[f], sw * xs);
      if (a.bias != nullptr) o += a.bias[f];
      if (a.y != nullptr) a.y[row * a.ldy + f] = o;
      if (a.w_dot != nullptr) dot = fmaf(o, a.w_dot[f], dot);
    }
  }
  if (a.w_dot != nullptr) a.t[row] = dot + (a.dot_bias ? a.dot_bias[0] : 0.f);
}

// ---------------------------------------------------------------- kept rows: gated gather + kept-neighbour count
// xp[p,:] = relu?(y[perm[p],:]) * tanh(score[perm[p]])  (layers.py:21) ; cnt[p] = #neighbours of perm[p] that are kept
template <int G>
__global__ __launch_bounds__(256) void sag_pool_gather(const float* __restrict__ y, int64_t ldy, const float* __restrict__ score,
                                                       const int* __restrict__ perm, const int* __restrict__ new_id,
                                                       const int* __restrict__ rowptr, const int* __restrict__ col, int64_t K,
                                                       int F, int relu_in, float* __restrict__ xp, int64_t ldo,
                                                       int* __restrict__ cnt) {
  constexpr int RPB = 256 / G;
  const int lig = threadIdx.x & (G - 1);
  const int64_t p = (int64_t)blockIdx.x * RPB + threadIdx.x / G;
  if (p >= K) return;
  const int r = perm[p];
  const float gate = tanhf(score[r]);
  const int nvec = F >> 2;
  if (y != nullptr && lig < nvec) {                         // y == nullptr: count only (the transposed adjacency)
    float4 v = ld4(y + (int64_t)r * ldy + 4 * lig);
    if (relu_in) v = relu4(v);
    *reinterpret_cast<float4*>(xp + p * ldo + 4 * lig) = make_float4(v.x * gate, v.y * gate, v.z * gate, v.w * gate);
  }
  float c = 0.f;                                            // exact: degrees are far below 2^24
  for (int e = rowptr[r] + lig; e < rowptr[r + 1]; e += G) c += (new_id[col[e]] >= 0) ? 1.f : 0.f;
  c = group_sum<G>(c);
  if (lig == 0) cnt[p] = (int)c;
}

// out[b, f] (+)= max_p xp[p, f] ; out[b, F + f] (+)= mean_p xp[p, f]  over the rows of graph b (network.py:36,40,44);
// arg[b, f] = the row that holds the max (ties -> smallest row).  grid (B, ceil(F / 64)), 64 columns x 4 row lanes.
__global__ __launch_bounds__(256) void sag_readout_kernel(const float* __restrict__ xp, int64_t ld, const int* __restrict__ gp, int F,
                                                          int accumulate, float* __restrict__ out, int64_t ldo,
                                                          int* __restrict__ arg) {
  __shared__ float s_m[4][64];
  __shared__ float s_s[4][64];
  __shared__ int s_a[4][64];
  const int b = blockIdx.x, c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = blockIdx.y * 64 + c;
  const int p0 = gp[b], p1 = gp[b + 1];
  float m = -INFINITY, s = 0.f;
  int am = p0;
  if (f < F) {
    for (int p = p0 + rl; p < p1; p += 4) {
      const float v = xp[(int64_t)p * ld + f];
      if (v > m) { m = v; am = p; }
      s += v;
    }
  }
  s_m[rl][c] = m; s_s[rl][c] = s; s_a[rl][c] = am;
  __syncthreads();
  if (rl != 0 || f >= F) return;
#pragma unroll
  for (int q = 1; q < 4; ++q) {
    const float v = s_m[q][c];
    const int a = s_a[q][c];
    if (v > m || (v == m && a < am)) { m = v; am = a; }
    s += s_s[q][c];
  }
  const float mean = s / (float)max(p1 - p0, 1);
  float* o = out + (int64_t)b * ldo;
  if (accumulate) { o[f] += m; o[F + f] += mean; }
  else { o[f] = m; o[F + f] = mean; }
  arg[(int64_t)b * F + f] = am;
}

// one row of the normalised propagation, computed by a lane group of G (float4 per lane): dinv_r sum_j dinv_j x_j + self_w_r x_r
// over the row's CSR entries [rowptr[r], rowend[r]) — the arithmetic (and its order) of gcn_propagate_vec4, for use inside the
// per-graph kernels (the next level's aggregation / the gradient arriving from the next level touch one graph's rows only)
template <int G>
__device__ __forceinline__ float4 prop_row(const int* __restrict__ rowptr, const int* __restrict__ rowend, const int* __restrict__ col,
                                           const float* __restrict__ dinv, const float* __restrict__ self_w,
                                           const float* __restrict__ x, int64_t ldx, int64_t row, int lig, int64_t co) {
  const int e0 = rowptr[row], e1 = rowend[row];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int eb = e0; eb < e1; eb += G) {
    const int me = eb + lig;
    const int cj = (me < e1) ? col[me] : 0;
    const float dj = (me < e1) ? dinv[cj] : 0.f;
    const int cnt = min(G, e1 - eb);
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {
      const int j0 = __shfl(cj, k, G), j1 = __shfl(cj, k + 1, G), j2 = __shfl(cj, k + 2, G), j3 = __shfl(cj, k + 3, G);
      const float4 v0 = ld4(x + (int64_t)j0 * ldx + co);
      const float4 v1 = ld4(x + (int64_t)j1 * ldx + co);
      const float4 v2 = ld4(x + (int64_t)j2 * ldx + co);
      const float4 v3 = ld4(x + (int64_t)j3 * ldx + co);
      fma4(acc, __shfl(dj, k, G), v0); fma4(acc, __shfl(dj, k + 1, G), v1);
      fma4(acc, __shfl(dj, k + 2, G), v2); fma4(acc, __shfl(dj, k + 3, G), v3);
    }
    for (; k < cnt; ++k) {
      const int j = __shfl(cj, k, G);
      fma4(acc, __shfl(dj, k, G), ld4(x + (int64_t)j * ldx + co));
    }
  }
  const float di = dinv[row], sw = self_w[row];
  const float4 xs = ld4(x + row * ldx + co);
  return make_float4(fmaf(di, acc.x, sw * xs.x), fmaf(di, acc.y, sw * xs.y), fmaf(di, acc.z, sw * xs.z), fmaf(di, acc.w, sw * xs.w));
}

// ---------------------------------------------------------------- one workgroup per graph: score -> top-k -> gather -> readout
// Everything between the conv output y and the pooled level touches ONE graph's rows only (the score layer's neighbours,
// the top-k segment, the kept rows, the readout), so for graphs of up to 4,096 nodes a single 1,024-thread workgroup does
//   t_j = relu(y_j) . w_s                      (lane group per row, DPP reduction)                         -> LDS
//   s_i = dinv_i sum_j dinv_j t_j + self_w_i t_i + b_s   (the GCNConv(C -> 1) score layer, layers.py:18)   -> score, sort keys
//   bitonic sort of (score, ~index) keys in LDS (descending, ties -> smaller node id)                       -> perm, new_id
//   xp[p] = relu(y[perm[p]]) * tanh(s)  (layers.py:21),  max || mean readout of the kept rows (network.py:36)
//   cnt[p] = kept neighbours of perm[p]  (input of the CSR filter's scan)
// in one launch instead of four (score propagate, top-k, gather, readout), with the relabelling map and the scores staying
// in LDS between the phases.
constexpr int PG_THREADS = 1024;
constexpr int PG_MAX_NODES = 4096;
constexpr int PG_SMALL_NODES = 256;       // batches whose graphs all fit this MAY run the per-graph kernels with 256-thread workgroups ...
// ... and do when there are more graphs than the chip hosts 1,024-thread workgroups at once (two per CU): four times the resident graphs
// per CU (8,192 IMDB-B graphs per launch).  Below that every graph has a CU to itself anyway and the wider workgroup finishes its phases
// sooner (more lane groups per phase): IMDB-B b128, the level-0 kernels 13.8 -> 10.2 us forward, 11.2 -> 7.2 us backward, the step
// 132.3 -> 123.3 us (measured with the threshold forced either way).
inline bool pg_small_block(int max_seg, int B) {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return max_seg <= PG_SMALL_NODES && B > 2 * ncu;
}
#ifndef TSGNN_PG_RGROUPS
#define TSGNN_PG_RGROUPS 8
#endif
constexpr int PG_RGROUPS = TSGNN_PG_RGROUPS;            // lane groups that run the gather / readout phase
constexpr int PG_RANK_MAX = 1024;        // up to this many (padded) nodes the top-k order comes from a rank count instead of a bitonic sort

struct PoolGraphArgs {
  const float* y; int64_t ldy;
  const int* rowptr; const int* rowend; const int* col; const float* dinv; const float* self_w;
  const float* w_s; const float* b_s;
  const int* gp; const int* gp_new;
  float* score; int* perm; int* new_id;
  float* xp; int64_t ldo; int* cnt;
  float* out; int64_t ldout; int* arg; int accumulate;
  int F;
  // CSR filter inside the kernel (all nullable together): the pooled level's adjacency goes to col_new at the graph's OLD
  // segment base rowptr[gp[b]] (kept entries never outnumber the old ones), rows [rowptr_new[p], rowend_new[p])
  int* rowptr_new; int* rowend_new; int* col_new; float* dinv_new; float* self_w_new;
  // with the filter: the NEXT level's aggregation A^' xp (nullable) — one launch less per level (gcn_propagate on the pooled rows)
  float* agg_next; int64_t ldagg;
};

// BT threads per workgroup: 1,024 for graphs of up to 4,096 nodes, 256 when no graph of the batch exceeds 256 nodes (TU graphs:
// four times the resident graphs per CU; at 8,192 IMDB-B graphs per launch 266 -> see profiles)
template <int G, int BT>
__global__ __launch_bounds__(BT) void sag_pool_graph_kernel(PoolGraphArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long pg_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int g0 = a.gp[b], n = a.gp[b + 1] - g0;
  const int k0 = a.gp_new[b], k = a.gp_new[b + 1] - k0;
  if (n <= 0) return;
  TR(0);
  int np = 1;
  while (np < n) np <<= 1;
  unsigned long long* keys = pg_smem;                                   // [np]
  float* t = reinterpret_cast<float*>(keys + np);                       // [np]
  int* nid = reinterpret_cast<int*>(t + np);                            // [np]
  float* rmax = reinterpret_cast<float*>(nid + np);                     // [PG_RGROUPS][F]
  float* rsum = rmax + PG_RGROUPS * a.F;                                // [PG_RGROUPS][F]
  int* rarg = reinterpret_cast<int*>(rsum + PG_RGROUPS * a.F);          // [PG_RGROUPS][F]
  constexpr int NG = BT / G;
  constexpr int RG = NG < PG_RGROUPS ? NG : PG_RGROUPS;                 // lane groups of the gather / readout phase (LDS laid out for PG_RGROUPS)
  // the pooled rows' CSR bounds and coefficients, kept for the next level's aggregation at the end of the kernel
  int* rpn = reinterpret_cast<int*>(np <= PG_RANK_MAX ? reinterpret_cast<float*>(rarg + PG_RGROUPS * a.F) + 2 * np
                                                       : reinterpret_cast<float*>(rarg + PG_RGROUPS * a.F));   // [np] row begin
  int* ren = rpn + np;                                                  // [np] row end
  float* dvn = reinterpret_cast<float*>(ren + np);                      // [np] dinv'
  float* swn = dvn + np;                                                // [np] self_w'
  const int lig = tid & (G - 1), grp = tid / G;
  const int nvec = a.F >> 2;
  const bool live = lig < nvec;
  const int co = live ? 4 * lig : 0;
  const float4 wv = live ? ld4(a.w_s + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  // (1) t_j = relu(y_j) . w_s   (four rows per lane group in flight: one round trip for up to 4 * NG rows)
  for (int j0 = grp; j0 < n; j0 += 4 * NG) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * NG;
      v[u] = (live && j < n) ? ld4(a.y + (int64_t)(g0 + j) * a.ldy + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * NG;
      const float d = group_sum<G>(dot4(relu4(v[u]), wv));
      if (lig == 0 && j < n) t[j] = d;
    }
  }
  __syncthreads();
  TR(1);
  // (2) scores and sort keys
  // The walks over the adjacency (here, the kept-neighbour counts and the filter below) give a row to EG = 8 lanes: a TU graph has
  // tens of nodes and a handful of neighbours per node, so one pass of the block covers the graph and a row costs three dependent
  // round trips (row pointer -> entries -> coefficients) instead of two per neighbour of a serial walk (traced: 2.5 -> 1.5 us).
  constexpr int EG = 8;
  const int sub = tid & (EG - 1), jr = tid / EG;
  const float bs = a.b_s ? a.b_s[0] : 0.f;
  for (int j = jr; j < np; j += BT / EG) {                  // uniform per lane group
    unsigned long long key = 0ull;                                      // padding sorts last
    if (j < n) {
      const int r = g0 + j;
      float acc = 0.f;
      const int e1 = a.rowend ? a.rowend[r] : a.rowptr[r + 1];
      for (int e = a.rowptr[r] + sub; e < e1; e += EG) {
        const int c = a.col[e];
        if ((unsigned)(c - g0) < (unsigned)n) acc = fmaf(a.dinv[c], t[c - g0], acc);   // graphs of a batch are disjoint (PyG collate)
      }
      acc = group_sum<EG>(acc);
      const float sc = fmaf(a.dinv[r], acc, a.self_w[r] * t[j]) + bs;
      if (sub == 0) a.score[r] = sc;
      key = ((unsigned long long)f32_ordered(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)j);
    }
    if (sub == 0) keys[j] = key;
  }
  __syncthreads();
  TR(2);
  // (3) sort, descending.  Small graphs (the common case: <= 1,024 nodes) by rank: thread i counts the keys above key i
  // (LDS broadcast reads, no barrier per step — a bitonic network on 256 keys is 36 barriers of a 16-wave block) and drops
  // its key at that position of a second array; keys are unique (they carry the node index).
  if (np <= PG_RANK_MAX) {
    unsigned long long* sorted = reinterpret_cast<unsigned long long*>(rarg + PG_RGROUPS * a.F);     // [np]
    for (int i = tid; i < np; i += BT) {
      const unsigned long long mine = keys[i];
      int rank = 0;
      if (i < n) {
#pragma unroll 8
        for (int j = 0; j < n; ++j) rank += keys[j] > mine ? 1 : 0;
        sorted[rank] = mine;
      } else {
        sorted[i] = 0ull;                                                // padding keeps its place behind the n real keys
      }
    }
    __syncthreads();
    keys = sorted;
  } else {
    for (int size = 2; size <= np; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int q = tid; q < (np >> 1); q += BT) {
          const int lo = 2 * q - (q & (stride - 1));
          const int hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long x = keys[lo], z = keys[hi];
          if ((x < z) == desc) { keys[lo] = z; keys[hi] = x; }
        }
        __syncthreads();
      }
    }
  }
  TR(3);
  // (4) perm and the relabelling map
  for (int i = tid; i < n; i += BT) {
    const int j = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
    const int id = i < k ? k0 + i : -1;
    if (i < k) a.perm[k0 + i] = g0 + j;
    nid[j] = id;
    a.new_id[g0 + j] = id;
  }
  __syncthreads();
  TR(4);
  // (5) gated gather of the kept rows + their max || mean readout  (PG_RGROUPS lane groups; ties of the max -> smallest row)
  if (grp < RG) {
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), s = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 am = make_int4(k0, k0, k0, k0);
    for (int p = grp; p < k; p += RG) {
      const unsigned long long key = keys[p];
      const int j = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
      const float gate = tanhf(ordered_f32((unsigned)(key >> 32)));
      if (live) {
        float4 v = relu4(ld4(a.y + (int64_t)(g0 + j) * a.ldy + co));
        v = make_float4(v.x * gate, v.y * gate, v.z * gate, v.w * gate);
        *reinterpret_cast<float4*>(a.xp + (int64_t)(k0 + p) * a.ldo + co) = v;
        if (v.x > m.x) { m.x = v.x; am.x = k0 + p; }
        if (v.y > m.y) { m.y = v.y; am.y = k0 + p; }
        if (v.z > m.z) { m.z = v.z; am.z = k0 + p; }
        if (v.w > m.w) { m.w = v.w; am.w = k0 + p; }
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    if (live) {
      *reinterpret_cast<float4*>(rmax + grp * a.F + co) = m;
      *reinterpret_cast<float4*>(rsum + grp * a.F + co) = s;
      *reinterpret_cast<int4*>(rarg + grp * a.F + co) = am;
    }
  }
  TR(5);
  // (6) kept neighbours of every kept row: the CSR filter's counts — and, when asked, the filter itself (layers.py:23-24):
  // block scan of the counts, rows laid out from the graph's old segment base, entries relabelled in their original order,
  // and the next level's gcn_norm coefficients
  int* cl = reinterpret_cast<int*>(t);                                  // [np] counts, then exclusive offsets (t is dead by now)
  for (int p = jr; p < np; p += BT / EG) {
    int c = 0;
    if (p < k) {
      const int j = (int)(0xFFFFFFFFu - (unsigned)(keys[p] & 0xFFFFFFFFull));
      const int r = g0 + j;
      const int e1 = a.rowend ? a.rowend[r] : a.rowptr[r + 1];
      float cf = 0.f;                                                   // counts <= 4,096: exact in fp32
      for (int e = a.rowptr[r] + sub; e < e1; e += EG) {
        const int cj = a.col[e] - g0;
        cf += ((unsigned)cj < (unsigned)n && nid[cj] >= 0) ? 1.f : 0.f;
      }
      c = (int)group_sum<EG>(cf);
      if (sub == 0) a.cnt[k0 + p] = c;
    }
    if (sub == 0) cl[p] = c;
  }
  __syncthreads();
  TR(6);
  if (a.col_new != nullptr) {
    // exclusive scan of cl[0..np): four consecutive counts per thread, wave scan, wave totals through LDS (np <= 4,096)
    __shared__ int wsum[BT / 64];
    const int lane = tid & 63, wid = tid >> 6;
    int v[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = (4 * tid + q < np) ? cl[4 * tid + q] : 0; sum += v[q]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int wbase = 0;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) wbase += (w < wid) ? wsum[w] : 0;
    int ex = wbase + inc - sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (4 * tid + q < np) cl[4 * tid + q] = ex;
      ex += v[q];
    }
    __syncthreads();
    const int base = a.rowptr[g0];
    const int gsh = (tid & 63) & ~(EG - 1);                             // this lane group's bits of a wave ballot
    for (int p = jr; p < k; p += BT / EG) {                             // uniform per lane group: its lanes vote together
      const int j = (int)(0xFFFFFFFFu - (unsigned)(keys[p] & 0xFFFFFFFFull));
      const int r = g0 + j;
      const int e0 = a.rowptr[r], e1 = a.rowend ? a.rowend[r] : a.rowptr[r + 1];
      const int o0 = base + cl[p];
      int o = o0;
      bool has_self = false;
      for (int eb = e0; eb < e1; eb += EG) {                            // EG entries per step, kept ones compacted in order
        const int e = eb + sub;
        int id = -1;
        if (e < e1) {
          const int cj = a.col[e] - g0;
          if ((unsigned)cj < (unsigned)n) id = nid[cj];
        }
        const unsigned gm = (unsigned)(__ballot(id >= 0) >> gsh) & ((1u << EG) - 1u);
        if (id >= 0) {
          a.col_new[o + __popc(gm & ((1u << sub) - 1u))] = id;
          has_self |= (id == k0 + p);
        }
        o += __popc(gm);
      }
      has_self = (((unsigned)(__ballot(has_self) >> gsh)) & ((1u << EG) - 1u)) != 0u;
      if (sub == 0) {
        a.rowptr_new[k0 + p] = o0;
        a.rowend_new[k0 + p] = o;
        const float d = (float)(o - o0) + (has_self ? 0.f : 1.f);
        const float di = 1.0f / sqrtf(d);
        a.dinv_new[k0 + p] = di;
        a.self_w_new[k0 + p] = has_self ? 0.f : di * di;
        rpn[p] = o0; ren[p] = o; dvn[p] = di; swn[p] = has_self ? 0.f : di * di;
      }
    }
  }
  __syncthreads();
  TR(7);
  if (a.agg_next != nullptr) {                              // pooled rows, their CSR and coefficients were written above by this block
    // prop_row's arithmetic with the row bounds and coefficients from LDS: entries -> pooled rows are the only global hops
    for (int p = grp; p < k; p += NG) {
      const int e0 = rpn[p], e1 = ren[p];
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int eb = e0; eb < e1; eb += G) {
        const int me = eb + lig;
        const int cj = (me < e1) ? a.col_new[me] : k0;
        const float dj = (me < e1) ? dvn[cj - k0] : 0.f;
        const int cnt = min(G, e1 - eb);
        int q = 0;
        for (; q + 4 <= cnt; q += 4) {
          const int j0 = __shfl(cj, q, G), j1 = __shfl(cj, q + 1, G), j2 = __shfl(cj, q + 2, G), j3 = __shfl(cj, q + 3, G);
          const float4 v0 = ld4(a.xp + (int64_t)j0 * a.ldo + co);
          const float4 v1 = ld4(a.xp + (int64_t)j1 * a.ldo + co);
          const float4 v2 = ld4(a.xp + (int64_t)j2 * a.ldo + co);
          const float4 v3 = ld4(a.xp + (int64_t)j3 * a.ldo + co);
          fma4(acc, __shfl(dj, q, G), v0); fma4(acc, __shfl(dj, q + 1, G), v1);
          fma4(acc, __shfl(dj, q + 2, G), v2); fma4(acc, __shfl(dj, q + 3, G), v3);
        }
        for (; q < cnt; ++q) {
          const int j = __shfl(cj, q, G);
          fma4(acc, __shfl(dj, q, G), ld4(a.xp + (int64_t)j * a.ldo + co));
        }
      }
      const float di = dvn[p], sw = swn[p];
      const float4 xs = ld4(a.xp + (int64_t)(k0 + p) * a.ldo + co);
      if (live)
        *reinterpret_cast<float4*>(a.agg_next + (int64_t)(k0 + p) * a.ldagg + co) =
            make_float4(fmaf(di, acc.x, sw * xs.x), fmaf(di, acc.y, sw * xs.y), fmaf(di, acc.z, sw * xs.z), fmaf(di, acc.w, sw * xs.w));
    }
  }
  TR(8);
  for (int f = tid; f < a.F; f += BT) {
    float m = rmax[f], s = rsum[f];
    int am = rarg[f];
#pragma unroll
    for (int q = 1; q < RG; ++q) {
      const float v = rmax[q * a.F + f];
      const int z = rarg[q * a.F + f];
      if (v > m || (v == m && z < am)) { m = v; am = z; }
      s += rsum[q * a.F + f];
    }
    const float mean = s / (float)max(k, 1);
    float* o = a.out + (int64_t)b * a.ldout;
    if (a.accumulate) { o[f] += m; o[a.F + f] += mean; }
    else { o[f] = m; o[a.F + f] = mean; }
    a.arg[(int64_t)b * a.F + f] = am;
  }
  TR(9);
  TR_END();
}

// ---------------------------------------------------------------- filter_adj on CSR (layers.py:23-24)
// new row p = old row perm[p]; its entries = the kept neighbours, relabelled, original order.  One wave per new row,
// ballot + popcount ranks.  Also emits the next level's gcn_norm coefficients (the new degree is known here).
__global__ __launch_bounds__(256) void csr_filter_fill_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                              const int* __restrict__ perm, const int* __restrict__ new_id, int64_t K,
                                                              const int* __restrict__ rowptr_new, int* __restrict__ col_new,
                                                              float* __restrict__ dinv_new, float* __restrict__ self_w_new) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= K) return;
  const int r = perm[p];
  const int e0 = rowptr[r], e1 = rowptr[r + 1];
  int out = rowptr_new[p];
  bool has_self = false;
  for (int base = e0; base < e1; base += 64) {
    const int e = base + lane;
    const int nid = (e < e1) ? new_id[col[e]] : -1;
    const bool keep = nid >= 0;
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      col_new[out + __popcll(mask & ((1ull << lane) - 1ull))] = nid;
      has_self |= (nid == (int)p);
    }
    out += __popcll(mask);
  }
  const bool any_self = __ballot(has_self) != 0ull;
  if (lane == 0 && dinv_new != nullptr) {
    const float d = (float)(out - rowptr_new[p]) + (any_self ? 0.f : 1.f);
    const float di = 1.0f / sqrtf(d);
    dinv_new[p] = di;
    self_w_new[p] = any_self ? 0.f : di * di;
  }
}

// ---------------------------------------------------------------- backward of gather + readout, per OLD row
// kept row r (p = new_id[r] >= 0, graph b):  dtot = dxp[p] + dread[b, F:2F] / k_b + [arg[b,:] == p] dread[b, :F]
//   dyb[r] = dtot * gate      (gradient w.r.t. relu(y)[r] through the gated gather)
//   dscore[r] = (1 - gate^2) <dtot, relu(y)[r]>
// dropped row: dyb[r] = 0, dscore[r] = 0.
template <int G>
__global__ __launch_bounds__(256) void sag_pool_bwd(const float* __restrict__ y, int64_t ldy, const float* __restrict__ score,
                                                    const int* __restrict__ new_id, const int* __restrict__ row_graph_new,
                                                    const int* __restrict__ gp_new, const int* __restrict__ arg,
                                                    const float* __restrict__ dxp, int64_t lddxp, const float* __restrict__ dread,
                                                    int64_t lddr, int64_t N, int F, int relu_in, float* __restrict__ dyb,
                                                    int64_t lddy, float* __restrict__ dscore) {
  constexpr int RPB = 256 / G;
  const int lig = threadIdx.x & (G - 1);
  const int64_t r = (int64_t)blockIdx.x * RPB + threadIdx.x / G;
  if (r >= N) return;
  const int nvec = F >> 2;
  const bool live = lig < nvec;
  const int co = live ? 4 * lig : 0;
  const int p = new_id[r];
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  float ds = 0.f;
  if (p >= 0) {                                             // group-uniform
    const int b = row_graph_new[p];
    const float inv_k = 1.0f / (float)max(gp_new[b + 1] - gp_new[b], 1);
    float4 d = dxp ? ld4(dxp + (int64_t)p * lddxp + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 dm = ld4(dread + (int64_t)b * lddr + F + co);
    const float4 dx = ld4(dread + (int64_t)b * lddr + co);
    const int4 am = *reinterpret_cast<const int4*>(arg + (int64_t)b * F + co);
    d.x += dm.x * inv_k + (am.x == p ? dx.x : 0.f);
    d.y += dm.y * inv_k + (am.y == p ? dx.y : 0.f);
    d.z += dm.z * inv_k + (am.z == p ? dx.z : 0.f);
    d.w += dm.w * inv_k + (am.w == p ? dx.w : 0.f);
    const float gate = tanhf(score[r]);
    float4 v = ld4(y + r * ldy + co);
    if (relu_in) v = relu4(v);
    float dot = live ? dot4(d, v) : 0.f;
    dot = group_sum<G>(dot);
    ds = dot * (1.f - gate * gate);
    g = make_float4(d.x * gate, d.y * gate, d.z * gate, d.w * gate);
  }
  if (live) *reinterpret_cast<float4*>(dyb + r * lddy + co) = g;
  if (lig == 0) dscore[r] = ds;
}

// du[r] = (dyb[r] + dt[r] * w_s) * [y_pre[r] > 0],  dt = A^ dscore (the score layer's propagate, transposed = itself)
// + per-block partial sums of  dw_s = sum_r dt[r] * relu(y_pre[r])  and  db_s = sum_r dscore[r]  (block b -> part[b, 0:F] and
// part[b, F]); sag_du_reduce adds them up in a fixed order.  (A last-block-done reduction inside this kernel was measured
// at 22 us per launch: the device-scope release fence of 256 blocks costs more than a 3 us launch.)
template <int G>
__global__ __launch_bounds__(256) void sag_du_kernel(const int* __restrict__ rowptr, const int* __restrict__ rowend,
                                                     const int* __restrict__ col,
                                                     const float* __restrict__ dinv, const float* __restrict__ self_w,
                                                     const float* __restrict__ dscore, const float* __restrict__ y, int64_t ldy,
                                                     const float* __restrict__ w_s, float* __restrict__ dyb, int64_t lddy, int64_t N,
                                                     int F, float* __restrict__ part) {
  constexpr int RPB = 256 / G;
  __shared__ float4 s_part[256];
  __shared__ float s_ds[256];
  const int lig = threadIdx.x & (G - 1), grp = threadIdx.x / G;
  const int nvec = F >> 2;
  const bool live = lig < nvec;
  const int co = live ? 4 * lig : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float dsum = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * RPB + grp; r < N; r += (int64_t)gridDim.x * RPB) {   // group-uniform
    float t = 0.f;
    const int e1 = rowend ? rowend[r] : rowptr[r + 1];
    for (int e = rowptr[r] + lig; e < e1; e += G) {
      const int j = col[e];
      t = fmaf(dinv[j], dscore[j], t);
    }
    t = group_sum<G>(t);
    const float dsr = dscore[r];
    const float dt = fmaf(dinv[r], t, self_w[r] * dsr);
    const float4 v = ld4(y + r * ldy + co);
    const float4 w = ld4(w_s + co);
    float4 d = ld4(dyb + r * lddy + co);
    d.x = v.x > 0.f ? fmaf(dt, w.x, d.x) : 0.f;
    d.y = v.y > 0.f ? fmaf(dt, w.y, d.y) : 0.f;
    d.z = v.z > 0.f ? fmaf(dt, w.z, d.z) : 0.f;
    d.w = v.w > 0.f ? fmaf(dt, w.w, d.w) : 0.f;
    if (live) *reinterpret_cast<float4*>(dyb + r * lddy + co) = d;
    fma4(acc, dt, relu4(v));
    if (lig == 0) dsum += dsr;
  }
  s_part[threadIdx.x] = acc;
  s_ds[threadIdx.x] = (lig == 0) ? dsum : 0.f;
  __syncthreads();
  // block partial: thread c < nvec sums the RPB groups' column vectors in group order
  if ((int)threadIdx.x < nvec) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < RPB; ++q) {
      const float4 v = s_part[q * G + threadIdx.x];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * (F + 4) + 4 * threadIdx.x) = s;
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int q = 0; q < RPB; ++q) s += s_ds[q * G];
    part[(int64_t)blockIdx.x * (F + 4) + F] = s;
  }
}


// dws[0:F], dbs[0] = column sums of part[nb, F + 4]: 8 slices of blocks per float4 column, then the slices in order
__global__ __launch_bounds__(256) void sag_du_reduce(float* __restrict__ part, int nb, int F, float* __restrict__ dws,
                                                     float* __restrict__ dbs, int chunk, int stride) {
  __shared__ float4 s_part[256];
  du_reduce_body(part, nb, F, dws, dbs, chunk, stride, (int)blockIdx.x, s_part);
}

constexpr int DU_REDUCE_CHUNK = 64;
// fixed-order sum of nb partial rows: one block up to 256 rows, two stages above (8,192 rows by one block: 150 us)
inline void launch_du_reduce(float* part, int nb, int F, float* dws, float* dbs, hipStream_t stream) {
  if (nb <= 256) {
    sag_du_reduce<<<1, 256, 0, stream>>>(part, nb, F, dws, dbs, nb, 1);
    return;
  }
  const int nb2 = (nb + DU_REDUCE_CHUNK - 1) / DU_REDUCE_CHUNK;
  sag_du_reduce<<<(unsigned)nb2, 256, 0, stream>>>(part, nb, F, nullptr, nullptr, DU_REDUCE_CHUNK, 1);
  sag_du_reduce<<<1, 256, 0, stream>>>(part, nb2, F, dws, dbs, nb2, DU_REDUCE_CHUNK);
}

// ---------------------------------------------------------------- backward of the level tail, one workgroup per graph
// pool_bwd -> dt = A^ dscore -> du also touch one graph's rows only: with dscore and dt in LDS the three phases are one launch
// (sag_pool_bwd + sag_du otherwise), followed by the fixed-order reduction of the per-graph partial sums of dw_s / db_s.
struct PoolGraphBwdArgs {
  const float* y; int64_t ldy; const float* score; const int* new_id;
  const int* gp; const int* gp_new; const int* arg;
  const float* dxp; int64_t lddxp; const float* dread; int64_t lddr;
  const int* rowptr; const int* rowend; const int* col; const float* dinv; const float* self_w; const float* w_s;
  float* du; int64_t lddu; float* part; int F;
  // instead of dxp: the next level's dagg (gradient of ITS aggregation) and that level's CSR / coefficients — dxp = A^' dagg is
  // formed per kept row here (symmetric adjacency), one launch less per level
  const float* dagg_next; int64_t lddagg;
  const int* rowptr_n; const int* rowend_n; const int* col_n; const float* dinv_n; const float* self_w_n;
};
template <int G, int BT>
__global__ __launch_bounds__(BT) void sag_pool_graph_bwd_kernel(PoolGraphBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pb_smem[];
  constexpr int NG = BT / G;
  constexpr int EG = 8;                       // lanes per row of the adjacency walk (B), as in the forward kernel
  const int b = blockIdx.x, tid = threadIdx.x;
  const int g0 = a.gp[b], n = a.gp[b + 1] - g0;
  const int k0 = a.gp_new[b], kb = a.gp_new[b + 1] - k0;
  const int F = a.F, nvec = F >> 2;
  const int n4 = (max(n, 0) + 3) & ~3;
  float* ds = pb_smem;                        // [n] dscore
  float* dt = ds + n4;                        // [n]
  int* nidl = reinterpret_cast<int*>(dt + n4);   // [n] new id of old row j, -1: dropped
  int* rowof = nidl + n4;                     // [kb] old row (offset in the graph) of kept row q
  float* dvn = reinterpret_cast<float*>(rowof + n4);   // [kb] next level's dinv of kept row q
  float* racc = dvn + n4;                     // [NG][F] partial dw_s
  const int lig = tid & (G - 1), grp = tid / G;
  const bool live = lig < nvec;
  const int co = live ? 4 * lig : 0;
  const float inv_k = 1.0f / (float)max(kb, 1);
  TR(0);
  float4 dm = make_float4(0.f, 0.f, 0.f, 0.f), dx = dm;
  int4 am = make_int4(-1, -1, -1, -1);
  if (live) {
    dm = ld4(a.dread + (int64_t)b * a.lddr + F + co);
    dx = ld4(a.dread + (int64_t)b * a.lddr + co);
    am = *reinterpret_cast<const int4*>(a.arg + (int64_t)b * F + co);
  }
  // the relabelling map of the graph, its inverse and the kept rows' coefficients: two round trips for the whole graph
  for (int j = tid; j < n; j += BT) {
    const int p = a.new_id[g0 + j];
    nidl[j] = p;
    if (p >= 0) {
      rowof[p - k0] = j;
      if (a.dagg_next != nullptr) dvn[p - k0] = a.dinv_n[p];
    } else {
      ds[j] = 0.f;                                                       // dropped rows carry no score gradient
    }
  }
  __syncthreads();
  // (A) kept rows only: gradient of the gated gather + readouts (dxp = A^' dagg' formed here when the next level hands over
  // its dagg), du <- dtot * gate (for now), dscore -> LDS.  Dropped rows have du = 0: (C) does not read them.
  for (int q = grp; q < kb; q += NG) {
    const int j = rowof[q], r = g0 + j, p = k0 + q;
    const float sc = a.score[r];
    const float4 yv = live ? ld4(a.y + (int64_t)r * a.ldy + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.dagg_next != nullptr) {
      const int e0 = a.rowptr_n[p], e1 = a.rowend_n[p];
      const float sw = a.self_w_n[p];
      const float4 xs = ld4(a.dagg_next + (int64_t)p * a.lddagg + co);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int eb = e0; eb < e1; eb += G) {                              // prop_row's arithmetic, coefficients from LDS
        const int me = eb + lig;
        const int cj = (me < e1) ? a.col_n[me] : k0;
        const float dj = (me < e1) ? dvn[cj - k0] : 0.f;
        const int cnt = min(G, e1 - eb);
        int u = 0;
        for (; u + 4 <= cnt; u += 4) {
          const int j0 = __shfl(cj, u, G), j1 = __shfl(cj, u + 1, G), j2 = __shfl(cj, u + 2, G), j3 = __shfl(cj, u + 3, G);
          const float4 v0 = ld4(a.dagg_next + (int64_t)j0 * a.lddagg + co);
          const float4 v1 = ld4(a.dagg_next + (int64_t)j1 * a.lddagg + co);
          const float4 v2 = ld4(a.dagg_next + (int64_t)j2 * a.lddagg + co);
          const float4 v3 = ld4(a.dagg_next + (int64_t)j3 * a.lddagg + co);
          fma4(acc, __shfl(dj, u, G), v0); fma4(acc, __shfl(dj, u + 1, G), v1);
          fma4(acc, __shfl(dj, u + 2, G), v2); fma4(acc, __shfl(dj, u + 3, G), v3);
        }
        for (; u < cnt; ++u) {
          const int jj = __shfl(cj, u, G);
          fma4(acc, __shfl(dj, u, G), ld4(a.dagg_next + (int64_t)jj * a.lddagg + co));
        }
      }
      const float di = dvn[q];
      d = make_float4(fmaf(di, acc.x, sw * xs.x), fmaf(di, acc.y, sw * xs.y), fmaf(di, acc.z, sw * xs.z), fmaf(di, acc.w, sw * xs.w));
    } else if (a.dxp != nullptr) {
      d = ld4(a.dxp + (int64_t)p * a.lddxp + co);
    }
    d.x += dm.x * inv_k + (am.x == p ? dx.x : 0.f);
    d.y += dm.y * inv_k + (am.y == p ? dx.y : 0.f);
    d.z += dm.z * inv_k + (am.z == p ? dx.z : 0.f);
    d.w += dm.w * inv_k + (am.w == p ? dx.w : 0.f);
    const float gate = tanhf(sc);
    float dot = live ? dot4(d, relu4(yv)) : 0.f;
    dot = group_sum<G>(dot);
    if (live) *reinterpret_cast<float4*>(a.du + (int64_t)r * a.lddu + co) = make_float4(d.x * gate, d.y * gate, d.z * gate, d.w * gate);
    if (lig == 0) ds[j] = dot * (1.f - gate * gate);
  }
  __syncthreads();
  TR(1);
  // (B) dt = A^ dscore (the score layer's propagate; symmetric adjacency), EG lanes per row
  {
    const int sub = tid & (EG - 1), jr = tid / EG;
    for (int j = jr; j < n; j += BT / EG) {
      const int r = g0 + j;
      const int e1 = a.rowend ? a.rowend[r] : a.rowptr[r + 1];
      float acc = 0.f;
      for (int e = a.rowptr[r] + sub; e < e1; e += EG) {
        const int c = a.col[e];
        if ((unsigned)(c - g0) < (unsigned)n) acc = fmaf(a.dinv[c], ds[c - g0], acc);
      }
      acc = group_sum<EG>(acc);
      if (sub == 0) dt[j] = fmaf(a.dinv[r], acc, a.self_w[r] * ds[j]);
    }
  }
  __syncthreads();
  TR(2);
  // (C) du = (du + dt w_s) [y > 0]; partial sums of dw_s = sum dt relu(y)   (two rows per lane group in flight)
  const float4 w = live ? ld4(a.w_s + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = grp; j0 < n; j0 += 2 * NG) {
    float4 v[2], d[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = j0 + u * NG;
      const bool ok = live && j < n;
      v[u] = ok ? ld4(a.y + (int64_t)(g0 + j) * a.ldy + co) : make_float4(0.f, 0.f, 0.f, 0.f);
      d[u] = (ok && nidl[j] >= 0) ? ld4(a.du + (int64_t)(g0 + j) * a.lddu + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = j0 + u * NG;
      if (live && j < n) {
        const float t = dt[j];
        float4 o;
        o.x = v[u].x > 0.f ? fmaf(t, w.x, d[u].x) : 0.f;
        o.y = v[u].y > 0.f ? fmaf(t, w.y, d[u].y) : 0.f;
        o.z = v[u].z > 0.f ? fmaf(t, w.z, d[u].z) : 0.f;
        o.w = v[u].w > 0.f ? fmaf(t, w.w, d[u].w) : 0.f;
        *reinterpret_cast<float4*>(a.du + (int64_t)(g0 + j) * a.lddu + co) = o;
        fma4(acc, t, relu4(v[u]));
      }
    }
  }
  if (live) *reinterpret_cast<float4*>(racc + grp * F + co) = acc;
  __syncthreads();
  TR(3);
  // (D) this graph's partial sums, groups in order
  for (int f = tid; f < F; f += BT) {
    float sum = 0.f;
    for (int q = 0; q < NG; ++q) sum += racc[q * F + f];
    a.part[(int64_t)b * (F + 4) + f] = sum;
  }
  if (tid == 0) {
    float sum = 0.f;
    for (int j = 0; j < n; ++j) sum += ds[j];
    a.part[(int64_t)b * (F + 4) + F] = sum;
  }
  TR(4);
  TR_END();
}

// single-launch scan of a short array (the per-row counts of one pooled level)
__global__ __launch_bounds__(1024) void scan_short_kernel(const int* __restrict__ in, int n, int* __restrict__ out) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 4096) {
    const int i0 = base + threadIdx.x * 4;
    int v[4];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (i0 + k < n) ? in[i0 + k] : 0; s += v[k]; }
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int wbase = carry_s, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int x = wsum[w];
      if (w < wid) wbase += x;
      tot += x;
    }
    int ex = wbase + inc - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) out[i0 + k] = ex;
      ex += v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carry_s;
}

template <int G>
void launch_prop(const PropArgs& a, hipStream_t s) {
  const unsigned nblk = (unsigned)ceil_div64(a.n_rows, 256 / G);
  gcn_propagate_vec4<G><<<nblk, 256, 0, s>>>(a, nblk);
}

constexpr int64_t PROP_RB_MIN_ROWS = 262144;   // below this the batch is cache-resident: one row per lane group wins

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int group_of(int F) { const int nv = F / 4; return nv <= 8 ? 8 : nv <= 16 ? 16 : nv <= 32 ? 32 : 64; }

}  // namespace

extern "C" {

int tsgnn_gcn_coef_f32(const int* rowptr, const int* col, int64_t n_rows, float* dinv, float* self_w, tsgnn_stream_t stream) {
  if (n_rows < 0 || !rowptr || !dinv || !self_w) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  gcn_coef_kernel<<<(unsigned)ceil_div64(n_rows, 256), 256, 0, stream>>>(rowptr, col, n_rows, dinv, self_w);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gcn_propagate_f32(const int* rowptr, const int* col, const float* dinv, const float* self_w, const float* x,
                            int64_t ldx, int relu_in, const float* bias, const float* w_dot, const float* dot_bias, float* y,
                            int64_t ldy, float* t, int64_t n_rows, int feat, tsgnn_stream_t stream) {
  return tsgnn_gcn_propagate_re_f32(rowptr, nullptr, col, dinv, self_w, x, ldx, relu_in, bias, w_dot, dot_bias, y, ldy, t, n_rows, feat,
                                    stream);
}

int tsgnn_gcn_propagate_re_f32(const int* rowptr, const int* rowend, const int* col, const float* dinv, const float* self_w,
                               const float* x, int64_t ldx, int relu_in, const float* bias, const float* w_dot, const float* dot_bias,
                               float* y, int64_t ldy, float* t, int64_t n_rows, int feat, tsgnn_stream_t stream) {
  if (n_rows < 0 || feat <= 0 || !rowptr || !dinv || !self_w || !x || ldx < feat) return TSGNN_EINVAL;
  if (!y && !w_dot) return TSGNN_EINVAL;
  if (y && ldy < feat) return TSGNN_EINVAL;
  if (w_dot && !t) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  PropArgs a{rowptr, rowend, col, dinv, self_w, x, ldx, bias, w_dot, dot_bias, y, ldy, t, n_rows, feat, relu_in};
  const bool vec_ok = feat % 4 == 0 && feat <= 256 && ldx % 4 == 0 && aligned16(x) && (!y || (ldy % 4 == 0 && aligned16(y))) &&
                      (!bias || aligned16(bias)) && (!w_dot || aligned16(w_dot));
  if (vec_ok && !rowend && n_rows >= PROP_RB_MIN_ROWS && (group_of(feat) == 16 || group_of(feat) == 32)) {
    if (group_of(feat) == 16) {
      const unsigned nblk = (unsigned)ceil_div64(n_rows, (256 / 16) * 4);
      gcn_propagate_vec4_rb<16, 4><<<nblk, 256, 0, stream>>>(a, nblk);
    } else {
      const unsigned nblk = (unsigned)ceil_div64(n_rows, (256 / 32) * 4);
      gcn_propagate_vec4_rb<32, 4><<<nblk, 256, 0, stream>>>(a, nblk);
    }
  } else if (vec_ok) {
    switch (group_of(feat)) {
      case 8: launch_prop<8>(a, stream); break;
      case 16: launch_prop<16>(a, stream); break;
      case 32: launch_prop<32>(a, stream); break;
      default: launch_prop<64>(a, stream); break;
    }
  } else if (feat <= PROP_NARROW_MAX && n_rows >= PROP_NARROW_MIN_ROWS) {
    gcn_propagate_narrow<<<(unsigned)ceil_div64(n_rows, 256), 256, 0, stream>>>(a);
  } else {
    gcn_propagate_generic<<<(unsigned)ceil_div64(n_rows, 4), 256, 0, stream>>>(a);
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* mean aggregation with the coefficients taken from the row lengths (no coefficient arrays):
 *   transpose = 0:  y[i] = (1 / max(len_i, 1)) sum_{j in row i} x[j]  (+ xself[i])      PyG SAGEConv's aggregation
 *   transpose = 1:  y[i] = sum_{j in row i} x[j] / max(len_j, 1)      (+ xself[i])      its adjoint on a symmetric edge list
 * xself (nullable, [n_rows, >= feat], leading dimension ldxs) is added with weight 1: the input gradient of lin_l(mean) + lin_r(x)
 * is ONE launch, the gathered half and the self half of d[agg || x] read from their own columns. */
int tsgnn_propagate_mean_f32(const int* rowptr, const int* rowend, const int* col, int transpose, const float* x, int64_t ldx,
                             const float* xself, int64_t ldxs, float* y, int64_t ldy, int64_t n_rows, int feat, tsgnn_stream_t stream) {
  if (n_rows < 0 || feat <= 0 || !rowptr || !col || !x || !y || ldx < feat || ldy < feat || (xself && ldxs < feat)) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  PropArgs a{rowptr, rowend, col, nullptr, nullptr, x, ldx, nullptr, nullptr, nullptr, y, ldy, nullptr, n_rows, feat, 0};
  a.scale_mode = transpose ? 2 : 1;
  a.xself = xself; a.ldxs = ldxs;
  const bool vec_ok = feat % 4 == 0 && feat <= 256 && ldx % 4 == 0 && aligned16(x) && ldy % 4 == 0 && aligned16(y) &&
                      (!xself || (ldxs % 4 == 0 && aligned16(xself)));
  if (vec_ok) {
    switch (group_of(feat)) {
      case 8: launch_prop<8>(a, stream); break;
      case 16: launch_prop<16>(a, stream); break;
      case 32: launch_prop<32>(a, stream); break;
      default: launch_prop<64>(a, stream); break;
    }
  } else {
    gcn_propagate_generic<<<(unsigned)ceil_div64(n_rows, 4), 256, 0, stream>>>(a);
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* narrow inputs (feat <= 8): agg = A^ x AND y = agg . w + bias (GCNConv's transform, w [feat, n_out] row-major) in one launch */
int tsgnn_gcn_propagate_affine_f32(const int* rowptr, const int* rowend, const int* col, const float* dinv, const float* self_w,
                                   const float* x, int64_t ldx, float* agg, int64_t ldagg, int64_t n_rows, int feat, const float* w,
                                   int64_t ldw, const float* bias, float* y, int64_t ldy, int n_out, tsgnn_stream_t stream) {
  if (n_rows < 0 || feat <= 0 || n_out <= 0 || !rowptr || !dinv || !self_w || !x || !agg || !w || !y || ldx < feat || ldagg < feat ||
      ldw < n_out || ldy < n_out)
    return TSGNN_EINVAL;
  if (feat > 8) return TSGNN_EUNSUPPORTED;
  if (n_rows == 0) return TSGNN_OK;
  PropArgs a{rowptr, rowend, col, dinv, self_w, x, ldx, nullptr, nullptr, nullptr, agg, ldagg, nullptr, n_rows, feat, 0,
             w, ldw, bias, y, ldy, n_out};
  gcn_propagate_generic<<<(unsigned)ceil_div64(n_rows, 4), 256, 0, stream>>>(a);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

#define SAG_DISPATCH(F, CALL)            \
  switch (group_of(F)) {                 \
    case 8: { constexpr int G = 8; CALL; } break;   \
    case 16: { constexpr int G = 16; CALL; } break; \
    case 32: { constexpr int G = 32; CALL; } break; \
    default: { constexpr int G = 64; CALL; } break; \
  }

int tsgnn_sag_supported(int F) { return (F > 0 && F % 4 == 0 && F <= 256) ? 1 : 0; }

int tsgnn_sag_pool_gather_f32(const float* y, int64_t ldy, const float* score, const int* perm, const int* new_id,
                              const int* rowptr, const int* col, int64_t K, int F, int relu_in, float* xp, int64_t ldo, int* cnt,
                              tsgnn_stream_t stream) {
  if (K < 0 || !score || !perm || !new_id || !rowptr || !cnt || ((y == nullptr) != (xp == nullptr))) return TSGNN_EINVAL;
  if (!tsgnn_sag_supported(F)) return TSGNN_EUNSUPPORTED;
  if (y && (ldy % 4 || ldo % 4 || !aligned16(y) || !aligned16(xp) || ldy < F || ldo < F)) return TSGNN_EUNSUPPORTED;
  if (K == 0) return TSGNN_OK;
  SAG_DISPATCH(F, (sag_pool_gather<G><<<(unsigned)ceil_div64(K, 256 / G), 256, 0, stream>>>(y, ldy, score, perm, new_id, rowptr, col, K,
                                                                                            F, relu_in, xp, ldo, cnt)));
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sag_pool_graph_max_nodes(void) { return PG_MAX_NODES; }

int tsgnn_sag_pool_graph_f32(const float* y, int64_t ldy, const int* rowptr, const int* rowend, const int* col, const float* dinv,
                             const float* self_w, const float* w_s, const float* b_s, const int* graph_ptr, const int* graph_ptr_new,
                             int B, int max_seg, int F, float* score, int* perm, int* new_id, float* xp, int64_t ldo, int* cnt,
                             float* out, int64_t ldout, int* arg, int accumulate, int* rowptr_new, int* rowend_new, int* col_new,
                             float* dinv_new, float* self_w_new, float* agg_next, int64_t ldagg, tsgnn_stream_t stream) {
  if (agg_next && (!col_new || ldagg < F)) return TSGNN_EINVAL;
  if (agg_next && (ldagg % 4 || !aligned16(agg_next))) return TSGNN_EUNSUPPORTED;
  if ((col_new != nullptr) != (rowptr_new != nullptr) || (col_new != nullptr) != (rowend_new != nullptr) ||
      (col_new != nullptr) != (dinv_new != nullptr) || (col_new != nullptr) != (self_w_new != nullptr))
    return TSGNN_EINVAL;
  if (!y || !rowptr || !dinv || !self_w || !w_s || !graph_ptr || !graph_ptr_new || !score || !perm || !new_id || !xp || !cnt || !out ||
      !arg || B <= 0 || max_seg < 0 || ldy < F || ldo < F || ldout < 2 * F)
    return TSGNN_EINVAL;
  if (!tsgnn_sag_supported(F) || max_seg > PG_MAX_NODES || ldy % 4 || ldo % 4 || !aligned16(y) || !aligned16(xp) || !aligned16(w_s))
    return TSGNN_EUNSUPPORTED;
  if (max_seg == 0) return TSGNN_OK;
  int np = 1;
  while (np < max_seg) np <<= 1;
  const size_t lds = (size_t)np * (8 + 4 + 4) + (size_t)PG_RGROUPS * F * 12 + (np <= PG_RANK_MAX ? (size_t)np * 8 : 0) + (size_t)np * 16;
  PoolGraphArgs a{y, ldy, rowptr, rowend, col, dinv, self_w, w_s, b_s, graph_ptr, graph_ptr_new, score, perm, new_id, xp, ldo, cnt,
                  out, ldout, arg, accumulate, F, rowptr_new, rowend_new, col_new, dinv_new, self_w_new, agg_next, ldagg};
#define PG_LAUNCH(GG)                                                                                                          \
  do {                                                                                                                         \
    if (pg_small_block(max_seg, B)) {                                                                                          \
      sag_pool_graph_kernel<GG, 256><<<(unsigned)B, 256, lds, stream>>>(a);                                                    \
    } else {                                                                                                                   \
      if (lds > 64 * 1024)                                                                                                     \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sag_pool_graph_kernel<GG, PG_THREADS>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
      sag_pool_graph_kernel<GG, PG_THREADS><<<(unsigned)B, PG_THREADS, lds, stream>>>(a);                                      \
    }                                                                                                                          \
  } while (0)
  switch (group_of(F)) {
    case 8: PG_LAUNCH(8); break;
    case 16: PG_LAUNCH(16); break;
    case 32: PG_LAUNCH(32); break;
    default: PG_LAUNCH(64); break;
  }
#undef PG_LAUNCH
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sag_readout_f32(const float* xp, int64_t ld, const int* graph_ptr, int B, int F, int accumulate, float* out, int64_t ldo,
                          int* arg, tsgnn_stream_t stream) {
  if (B <= 0 || F <= 0 || !xp || !graph_ptr || !out || !arg || ld < F || ldo < 2 * F) return TSGNN_EINVAL;
  sag_readout_kernel<<<dim3((unsigned)B, (unsigned)((F + 63) / 64)), 256, 0, stream>>>(xp, ld, graph_ptr, F, accumulate, out, ldo, arg);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_csr_filter_fill(const int* rowptr, const int* col, const int* perm, const int* new_id, int64_t K, const int* rowptr_new,
                          int* col_new, float* dinv_new, float* self_w_new, tsgnn_stream_t stream) {
  if (K < 0 || !rowptr || !perm || !new_id || !rowptr_new || !col_new) return TSGNN_EINVAL;
  if ((dinv_new == nullptr) != (self_w_new == nullptr)) return TSGNN_EINVAL;
  if (K == 0) return TSGNN_OK;
  csr_filter_fill_kernel<<<(unsigned)ceil_div64(K, 4), 256, 0, stream>>>(rowptr, col, perm, new_id, K, rowptr_new, col_new, dinv_new,
                                                                        self_w_new);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_scan_short_i32(const int* in, int64_t n, int* out, tsgnn_stream_t stream) {
  if (n < 0 || n > (1 << 20) || !out || (n > 0 && !in)) return TSGNN_EINVAL;
  scan_short_kernel<<<1, 1024, 0, stream>>>(in, (int)n, out);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sag_pool_bwd_f32(const float* y, int64_t ldy, const float* score, const int* new_id, const int* row_graph_new,
                           const int* graph_ptr_new, const int* arg, const float* dxp, int64_t lddxp, const float* dread,
                           int64_t lddr, int64_t N, int F, int relu_in, float* dyb, int64_t lddy, float* dscore,
                           tsgnn_stream_t stream) {
  if (N < 0 || !y || !score || !new_id || !row_graph_new || !graph_ptr_new || !arg || !dread || !dyb || !dscore) return TSGNN_EINVAL;
  if (!tsgnn_sag_supported(F) || ldy % 4 || lddy % 4 || lddr % 4 || (dxp && (lddxp % 4 || !aligned16(dxp))) || !aligned16(y) ||
      !aligned16(dyb) || !aligned16(dread) || !aligned16(arg))
    return TSGNN_EUNSUPPORTED;
  if (N == 0) return TSGNN_OK;
  SAG_DISPATCH(F, (sag_pool_bwd<G><<<(unsigned)ceil_div64(N, 256 / G), 256, 0, stream>>>(y, ldy, score, new_id, row_graph_new,
                                                                                         graph_ptr_new, arg, dxp, lddxp, dread, lddr, N,
                                                                                         F, relu_in, dyb, lddy, dscore)));
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sag_pool_graph_bwd_f32(const float* y, int64_t ldy, const float* score, const int* new_id, const int* graph_ptr,
                                 const int* graph_ptr_new, const int* arg, const float* dxp, int64_t lddxp, const float* dread,
                                 int64_t lddr, const int* rowptr, const int* rowend, const int* col, const float* dinv,
                                 const float* self_w, const float* w_s, int B, int max_seg, int F, float* du, int64_t lddu, float* part,
                                 float* dws, float* dbs, const float* dagg_next, int64_t lddagg, const int* rowptr_n,
                                 const int* rowend_n, const int* col_n, const float* dinv_n, const float* self_w_n,
                                 tsgnn_stream_t stream) {
  if (dagg_next && (dxp || !rowptr_n || !rowend_n || !col_n || !dinv_n || !self_w_n || lddagg < F)) return TSGNN_EINVAL;
  if (dagg_next && (lddagg % 4 || !aligned16(dagg_next))) return TSGNN_EUNSUPPORTED;
  if (!y || !score || !new_id || !graph_ptr || !graph_ptr_new || !arg || !dread || !rowptr || !dinv || !self_w || !w_s || !du || !part ||
      ((dws == nullptr) != (dbs == nullptr)) || B <= 0 || max_seg < 0)
    return TSGNN_EINVAL;
  if (!tsgnn_sag_supported(F) || max_seg > PG_MAX_NODES || ldy % 4 || lddu % 4 || lddr % 4 || (dxp && (lddxp % 4 || !aligned16(dxp))) ||
      !aligned16(y) || !aligned16(du) || !aligned16(dread) || !aligned16(arg) || !aligned16(w_s) || !aligned16(part))
    return TSGNN_EUNSUPPORTED;
  if (max_seg == 0) return TSGNN_OK;
  PoolGraphBwdArgs a{y, ldy, score, new_id, graph_ptr, graph_ptr_new, arg, dxp, lddxp, dread, lddr, rowptr, rowend, col, dinv, self_w,
                     w_s, du, lddu, part, F, dagg_next, lddagg, rowptr_n, rowend_n, col_n, dinv_n, self_w_n};
  const int G_ = group_of(F);
  const int bt = pg_small_block(max_seg, B) ? 256 : PG_THREADS;
  const size_t lds = sizeof(float) * (5 * (size_t)((max_seg + 3) & ~3) + (size_t)(bt / G_) * F);
#define PGB_LAUNCH(GG)                                                                                                             \
  do {                                                                                                                             \
    if (bt == 256) {                                                                                                               \
      sag_pool_graph_bwd_kernel<GG, 256><<<(unsigned)B, 256, lds, stream>>>(a);                                                    \
    } else {                                                                                                                       \
      if (lds > 64 * 1024)                                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sag_pool_graph_bwd_kernel<GG, PG_THREADS>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                           \
      sag_pool_graph_bwd_kernel<GG, PG_THREADS><<<(unsigned)B, PG_THREADS, lds, stream>>>(a);                                      \
    }                                                                                                                              \
  } while (0)
  switch (G_) {
    case 8: PGB_LAUNCH(8); break;
    case 16: PGB_LAUNCH(16); break;
    case 32: PGB_LAUNCH(32); break;
    default: PGB_LAUNCH(64); break;
  }
#undef PGB_LAUNCH
  if (dws != nullptr) launch_du_reduce(part, (int)B, F, dws, dbs, stream);   // NULL: reduced by tsgnn_linear_wgrad_du_f32
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* fixed-order sum of nb partial rows part[nb][F + 4] -> dws[F], dbs[1] (what tsgnn_sag_pool_graph_bwd_f32 runs itself unless it
 * is called with dws = dbs = NULL) */
int tsgnn_sag_du_reduce_f32(float* part, int nb, int F, float* dws, float* dbs, tsgnn_stream_t stream) {
  if (!part || !dws || !dbs || nb <= 0 || !tsgnn_sag_supported(F) || !aligned16(part) || !aligned16(dws)) return TSGNN_EINVAL;
  launch_du_reduce(part, nb, F, dws, dbs, stream);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* blocks of the du kernel for N rows of F features (= rows of the partial-sum workspace, each F + 4 floats) */
int tsgnn_sag_du_blocks(int64_t N, int F) {
  if (N <= 0 || !tsgnn_sag_supported(F)) return 0;
  const int64_t nb = ceil_div64(N, 256 / group_of(F));
  return (int)(nb < 256 ? nb : 256);
}

int tsgnn_sag_du_f32(const int* rowptr, const int* rowend, const int* col, const float* dinv, const float* self_w, const float* dscore, const float* y,
                     int64_t ldy, const float* w_s, float* dyb, int64_t lddy, int64_t N, int F, float* part, float* dws, float* dbs,
                     tsgnn_stream_t stream) {
  if (N <= 0 || !rowptr || !dinv || !self_w || !dscore || !y || !w_s || !dyb || !part || !dws || !dbs) return TSGNN_EINVAL;
  if (!tsgnn_sag_supported(F) || ldy % 4 || lddy % 4 || !aligned16(y) || !aligned16(dyb) || !aligned16(w_s) || !aligned16(part))
    return TSGNN_EUNSUPPORTED;
  const unsigned nb = (unsigned)tsgnn_sag_du_blocks(N, F);
  SAG_DISPATCH(F, (sag_du_kernel<G><<<nb, 256, 0, stream>>>(rowptr, rowend, col, dinv, self_w, dscore, y, ldy, w_s, dyb, lddy, N, F, part)));
  launch_du_reduce(part, (int)nb, F, dws, dbs, stream);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
