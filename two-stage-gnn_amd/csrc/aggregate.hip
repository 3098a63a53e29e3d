// Neighbour aggregation: Y = A.X (+ self term) over a CSR adjacency.
//
// Replaces the reference's dense  torch.matmul(adj[B,Nmax,Nmax], x[B,Nmax,F])  (encoders.py:33-35)
// and PyG's scatter-add propagate (Code/sag/network.py:34).  HBM-bound gather:
//   * one aligned lane group of G = F/4 lanes (power of two, <= 64) owns one output row, each lane
//     one float4 (16 B) of it -> a 128-float row is read/written as one 512-B coalesced segment,
//     two rows per wave;
//   * the group's lanes fetch the row's column indices with ONE coalesced load and broadcast them
//     with wave shuffles; four neighbour rows are kept in flight per lane;
//   * no atomics (row-gather, not scatter): results are bitwise reproducible;
//   * blocks are remapped so each XCD's L2 sees a contiguous range of rows (= whole graphs, whose
//     neighbour rows are the rows being gathered).
// Algorithmic bytes per pass: 4NF (read X) + 4NF (write Y) + 4E (col) + 4(N+1) (rowptr) [+4E val].
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

constexpr int64_t RB_MIN_ROWS = 262144;  // below this the batch is cache-resident (measured: B=256 DD graphs 52% of HBM peak one row per group)

struct SpmmArgs {
  const int* rowptr;
  const int* col;
  const float* val;        // nullable: unit weights
  const float* self_w;     // nullable: per-row self coefficient (GCN: 1/deg_i)
  const float* x;
  float* y;
  int64_t ldx, ldy;
  int64_t n_rows;
  int feat;
  float self_scalar;       // add_self (encoders.py:34-35): y += self_scalar * x[row]
  int relu_in;             // consume relu(x) instead of x (activation folded into the gather)
  int accumulate;          // y += result instead of y = result
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ void fma4(float4& a, float w, float4 v) {
  a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
}

// G lanes per row, float4 per lane, NCH feature chunks of 4*G floats (NCH > 1 only when G == 64)
template <int G, bool WEIGHTED, bool RELU>
__global__ __launch_bounds__(256) void spmm_vec4(SpmmArgs a, unsigned nblk) {
  constexpr int ROWS_PER_BLOCK = 256 / G;
  const unsigned lb = xcd_remap(blockIdx.x, nblk);
  const int lane_in_group = threadIdx.x & (G - 1);
  const int64_t row = (int64_t)lb * ROWS_PER_BLOCK + threadIdx.x / G;
  if (row >= a.n_rows) return;
  const int e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
  const int nvec = a.feat >> 2;                         // float4 per row
  for (int cb = 0; cb < nvec; cb += G) {                // group-uniform trip count (shuffles below)
    const int c = cb + lane_in_group;
    const bool live = c < nvec;
    const int64_t co = live ? 4 * c : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int eb = e0; eb < e1; eb += G) {
      // cooperative, coalesced index (and weight) fetch for up to G neighbours
      const int me = eb + lane_in_group;
      int cj = (me < e1) ? a.col[me] : 0;
      float wj = 0.f;
      if (WEIGHTED) wj = (me < e1) ? a.val[me] : 0.f;
      const int cnt = min(G, e1 - eb);
      int k = 0;
      for (; k + 4 <= cnt; k += 4) {
        const int j0 = __shfl(cj, k, G), j1 = __shfl(cj, k + 1, G), j2 = __shfl(cj, k + 2, G), j3 = __shfl(cj, k + 3, G);
        float4 v0 = ld4(a.x + (int64_t)j0 * a.ldx + co);
        float4 v1 = ld4(a.x + (int64_t)j1 * a.ldx + co);
        float4 v2 = ld4(a.x + (int64_t)j2 * a.ldx + co);
        float4 v3 = ld4(a.x + (int64_t)j3 * a.ldx + co);
        if (RELU) { v0 = relu4(v0); v1 = relu4(v1); v2 = relu4(v2); v3 = relu4(v3); }
        if (WEIGHTED) {
          fma4(acc, __shfl(wj, k, G), v0); fma4(acc, __shfl(wj, k + 1, G), v1);
          fma4(acc, __shfl(wj, k + 2, G), v2); fma4(acc, __shfl(wj, k + 3, G), v3);
        } else {
          acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
          acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
          acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
          acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
        }
      }
      for (; k < cnt; ++k) {
        const int j = __shfl(cj, k, G);
        float4 v = ld4(a.x + (int64_t)j * a.ldx + co);
        if (RELU) v = relu4(v);
        if (WEIGHTED) fma4(acc, __shfl(wj, k, G), v);
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
      }
    }
    if (!live) continue;
    if (a.self_w != nullptr || a.self_scalar != 0.f) {
      float4 v = ld4(a.x + row * a.ldx + co);
      if (RELU) v = relu4(v);
      const float s = (a.self_w ? a.self_w[row] : 0.f) + a.self_scalar;
      fma4(acc, s, v);
    }
    float* yp = a.y + row * a.ldy + co;
    if (a.accumulate) {
      const float4 o = ld4(yp);
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    *reinterpret_cast<float4*>(yp) = acc;
  }
}


// ---------------------------------------------------------------- row-batched CSR gather (large batches)
// With one row per lane group the gather is a chain of dependent memory round trips (rowptr -> col -> rows ->
// store) and, once the batch no longer fits the caches, latency- rather than bandwidth-bound.  Here a lane group
// owns RB consecutive rows: ONE rowptr fetch and ONE coalesced index fetch cover all their neighbours, and the
// feature rows of the whole batch are gathered eight at a time, so ~RB x fewer dependent hops per row and
// eight 16-byte loads in flight per lane.  Per-row summation order is unchanged (bitwise equal to spmm_vec4).
template <int G, int RB, bool WEIGHTED>
__global__ __launch_bounds__(256) void spmm_vec4_rb(SpmmArgs a, unsigned nblk) {
  constexpr int GROUPS = 256 / G;
  const unsigned lb = xcd_remap(blockIdx.x, nblk);
  const int lig = threadIdx.x & (G - 1);
  const int64_t row0 = ((int64_t)lb * GROUPS + threadIdx.x / G) * RB;
  if (row0 >= a.n_rows) return;
  const int nvec = a.feat >> 2;
  const bool live = lig < nvec;                          // nvec <= G on this path
  const int64_t co = live ? 4 * lig : 0;
  // rowptr[row0 .. row0+RB] with one coalesced load (clamped at n_rows -> empty rows)
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
    float wj = 0.f;
    if (WEIGHTED) wj = (me < eb[RB]) ? a.val[me] : 0.f;
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
          const float w = WEIGHTED ? __shfl(wj, (k + u) & (G - 1), G) : 1.f;
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            if (e >= eb[r] && e < eb[r + 1]) {
              if (WEIGHTED) fma4(acc[r], w, v[u]);
              else { acc[r].x += v[u].x; acc[r].y += v[u].y; acc[r].z += v[u].z; acc[r].w += v[u].w; }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int64_t row = row0 + r;
    if (row < a.n_rows && live) {
      if (a.self_w != nullptr || a.self_scalar != 0.f)
        fma4(acc[r], (a.self_w ? a.self_w[row] : 0.f) + a.self_scalar, ld4(a.x + row * a.ldx + co));
      float* yp = a.y + row * a.ldy + co;
      if (a.accumulate) { const float4 o = ld4(yp); acc[r].x += o.x; acc[r].y += o.y; acc[r].z += o.z; acc[r].w += o.w; }
      *reinterpret_cast<float4*>(yp) = acc[r];
    }
  }
}

// scalar fallback: any F / any leading dimension (e.g. the raw 89-wide DD input, F = 1 scores).
// One wave per row, lane f handles features f, f+64, ...
template <bool WEIGHTED, bool RELU>
__global__ __launch_bounds__(256) void spmm_scalar(SpmmArgs a, unsigned nblk) {
  const unsigned lb = xcd_remap(blockIdx.x, nblk);
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)lb * 4 + (threadIdx.x >> 6);
  if (row >= a.n_rows) return;
  const int e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
  for (int fb = 0; fb < a.feat; fb += 64) {          // wave-uniform trip count (shuffles below)
    const int f = fb + lane;
    const bool live = f < a.feat;
    const int fo = live ? f : 0;
    float acc = 0.f;
    for (int eb = e0; eb < e1; eb += 64) {
      const int me = eb + lane;
      int cj = (me < e1) ? a.col[me] : 0;
      float wj = 1.f;
      if (WEIGHTED) wj = (me < e1) ? a.val[me] : 0.f;
      const int cnt = min(64, e1 - eb);
      for (int k = 0; k < cnt; ++k) {
        const int j = __shfl(cj, k, 64);
        float v = a.x[(int64_t)j * a.ldx + fo];
        if (RELU) v = fmaxf(v, 0.f);
        acc = WEIGHTED ? fmaf(__shfl(wj, k, 64), v, acc) : acc + v;
      }
    }
    if (!live) continue;
    if (a.self_w != nullptr || a.self_scalar != 0.f) {
      float v = a.x[row * a.ldx + f];
      if (RELU) v = fmaxf(v, 0.f);
      acc = fmaf((a.self_w ? a.self_w[row] : 0.f) + a.self_scalar, v, acc);
    }
    float* yp = a.y + row * a.ldy + f;
    *yp = a.accumulate ? (*yp + acc) : acc;
  }
}

// F == 1 (GCN score layer C->1 after the transform, Code/sag/layers.py:18): one lane per row
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void spmv_rows(SpmmArgs a) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= a.n_rows) return;
  float acc = 0.f;
  for (int e = a.rowptr[row]; e < a.rowptr[row + 1]; ++e) {
    float v = a.x[(int64_t)a.col[e] * a.ldx];
    if (a.relu_in) v = fmaxf(v, 0.f);
    acc = WEIGHTED ? fmaf(a.val[e], v, acc) : acc + v;
  }
  if (a.self_w != nullptr || a.self_scalar != 0.f) {
    float v = a.x[row * a.ldx];
    if (a.relu_in) v = fmaxf(v, 0.f);
    acc = fmaf((a.self_w ? a.self_w[row] : 0.f) + a.self_scalar, v, acc);
  }
  float* yp = a.y + row * a.ldy;
  *yp = a.accumulate ? (*yp + acc) : acc;
}


// ---------------------------------------------------------------- ELL (fixed-width) fast path
// The CSR gather has three dependent memory round trips before the store (rowptr -> col -> rows).  For the
// small, nearly regular degrees of the TU graphs (DD: mean 5, max ~15) a fixed-width index table
// ell[row][W] (-1 padded) removes the rowptr hop and lets a lane group issue ALL of its row's gathers
// back to back (W float4 loads in flight per lane).  Rows with more than W neighbours keep their tail in CSR
// and are finished by the CSR kernel with accumulate=1.  Same summation order as the CSR kernel (bitwise equal).
template <int G, int W>
__global__ __launch_bounds__(256) void spmm_ell_vec4(const int* __restrict__ ell, const float* __restrict__ x, int64_t ldx,
                                                     float* __restrict__ y, int64_t ldy, int64_t n_rows, int nvec,
                                                     float self_scalar, unsigned nblk) {
  constexpr int ROWS_PER_BLOCK = 256 / G;
  const unsigned lb = xcd_remap(blockIdx.x, nblk);
  const int lig = threadIdx.x & (G - 1);
  const int64_t row = (int64_t)lb * ROWS_PER_BLOCK + threadIdx.x / G;
  if (row >= n_rows) return;
  // W <= G: lane k of the group fetches index k (one coalesced 4*W-byte segment per row)
  int cj = (lig < W) ? ell[row * W + lig] : -1;
  const bool live = lig < nvec;
  const int64_t co = live ? 4 * lig : 0;
  float4 v[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int j = __shfl(cj, k, G);
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j >= 0) v[k] = ld4(x + (int64_t)j * ldx + co);
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < W; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
  if (self_scalar != 0.f) fma4(acc, self_scalar, ld4(x + row * ldx + co));
  if (live) *reinterpret_cast<float4*>(y + row * ldy + co) = acc;
}


__global__ void csr_to_ell_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int64_t n_rows, int W,
                                  int* __restrict__ ell, int* __restrict__ tail_cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * W) return;
  const int64_t r = i / W;
  const int k = (int)(i % W);
  const int e0 = rowptr[r], d = rowptr[r + 1] - e0;
  ell[i] = k < d ? col[e0 + k] : -1;
  if (k == 0 && tail_cnt) tail_cnt[r] = d > W ? d - W : 0;
}
__global__ void csr_tail_fill(const int* __restrict__ rowptr, const int* __restrict__ col, const int* __restrict__ tail_ptr,
                              int64_t n_rows, int W, int* __restrict__ tail_col) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int e0 = rowptr[r] + W, e1 = rowptr[r + 1];
  int o = tail_ptr[r];
  for (int e = e0; e < e1; ++e) tail_col[o++] = col[e];
}

template <int G>
void launch_vec4(const SpmmArgs& a, hipStream_t s) {
  const unsigned nblk = (unsigned)ceil_div64(a.n_rows, 256 / G);
  TSGNN_KNAME("spmm_vec4<%d,%s,%s>", G, a.val ? "true" : "false", a.relu_in ? "true" : "false");
  if (a.val) {
    if (a.relu_in) spmm_vec4<G, true, true><<<nblk, 256, 0, s>>>(a, nblk);
    else spmm_vec4<G, true, false><<<nblk, 256, 0, s>>>(a, nblk);
  } else {
    if (a.relu_in) spmm_vec4<G, false, true><<<nblk, 256, 0, s>>>(a, nblk);
    else spmm_vec4<G, false, false><<<nblk, 256, 0, s>>>(a, nblk);
  }
}

// per-row degree / GCN symmetric normalisation -------------------------------------------------
// deg_i = sum_e val[e] (+1 for the added self loop); dinv = deg^-1/2 (0 if deg == 0);
// val_out[e] = dinv[i] * val[e] * dinv[col[e]],  self_w[i] = dinv[i]^2 * self_fill
__global__ void gcn_deg_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                               const float* __restrict__ val, int64_t n, float self_fill,
                               float* __restrict__ dinv, float* __restrict__ self_w) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = 0.f;
  bool has_self = false;                    // add_remaining_self_loops: keep an existing self loop
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    d += val ? val[e] : 1.f;
    has_self |= (col[e] == i);
  }
  if (!has_self) d += self_fill;
  dinv[i] = d > 0.f ? 1.0f / sqrtf(d) : 0.f;
  self_w[i] = has_self ? 0.f : self_fill;   // finished (x dinv^2) by gcn_norm_kernel
}
__global__ void gcn_norm_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                const float* __restrict__ val, const float* __restrict__ dinv, int64_t n,
                                float self_fill, float* __restrict__ val_out, float* __restrict__ self_w) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float di = dinv[i];
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) val_out[e] = di * (val ? val[e] : 1.f) * dinv[col[e]];
  self_w[i] = di * di * self_w[i];
}


// gradient of the weighted aggregation with respect to its weights: dval[e] = dy[i] . x[col[e]] for the entries e of row i,
// dself[i] = dy[i] . x[i]; one wave per row, the row of dy held in registers (feat <= 1024)
__global__ __launch_bounds__(256) void sddmm_rows_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                        const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                        int64_t ldx, int64_t n, int feat, float* __restrict__ dval,
                                                        float* __restrict__ dself) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  float g[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) g[u] = lane + 64 * u < feat ? dy[i * lddy + lane + 64 * u] : 0.f;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int e = e0 - (dself ? 1 : 0); e < e1; ++e) {
    const int64_t c = e < e0 ? i : (int64_t)col[e];
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (lane + 64 * u < feat) acc = fmaf(g[u], x[c * ldx + lane + 64 * u], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      if (e < e0) dself[i] = acc;
      else dval[e] = acc;
    }
  }
}

}  // namespace

extern "C" {

int tsgnn_csr_spmm_f32(const int* rowptr, const int* col, const float* val, const float* self_w,
                       const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows, int feat,
                       float self_scalar, int relu_in, int accumulate, hipStream_t stream) {
  if (n_rows < 0 || feat <= 0 || !rowptr || !x || !y || ldx < feat || ldy < feat) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  SpmmArgs a{rowptr, col, val, self_w, x, y, ldx, ldy, n_rows, feat, self_scalar, relu_in, accumulate};
  const bool vec_ok = (feat % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  if (feat == 1) {
    const unsigned nblk = (unsigned)ceil_div64(n_rows, 256);
    if (val) spmv_rows<true><<<nblk, 256, 0, stream>>>(a);
    else spmv_rows<false><<<nblk, 256, 0, stream>>>(a);
  } else if (vec_ok && !relu_in && n_rows >= RB_MIN_ROWS && feat / 4 <= 32 && feat / 4 > 8) {
    // large batches: row-batched gather (RB = 4 rows per lane group)
    const int nvec = feat / 4;
    TSGNN_KNAME("spmm_vec4_rb<%d,4,%s>", nvec <= 16 ? 16 : 32, val ? "true" : "false");
    if (nvec <= 16) {
      const unsigned nblk = (unsigned)ceil_div64(n_rows, (256 / 16) * 4);
      if (val) spmm_vec4_rb<16, 4, true><<<nblk, 256, 0, stream>>>(a, nblk);
      else spmm_vec4_rb<16, 4, false><<<nblk, 256, 0, stream>>>(a, nblk);
    } else {
      const unsigned nblk = (unsigned)ceil_div64(n_rows, (256 / 32) * 4);
      if (val) spmm_vec4_rb<32, 4, true><<<nblk, 256, 0, stream>>>(a, nblk);
      else spmm_vec4_rb<32, 4, false><<<nblk, 256, 0, stream>>>(a, nblk);
    }
  } else if (vec_ok) {
    const int nvec = feat / 4;
    if (nvec <= 4) launch_vec4<4>(a, stream);
    else if (nvec <= 8) launch_vec4<8>(a, stream);
    else if (nvec <= 16) launch_vec4<16>(a, stream);
    else if (nvec <= 32) launch_vec4<32>(a, stream);
    else launch_vec4<64>(a, stream);
  } else {
    const unsigned nblk = (unsigned)ceil_div64(n_rows, 4);
    if (val) {
      if (relu_in) spmm_scalar<true, true><<<nblk, 256, 0, stream>>>(a, nblk);
      else spmm_scalar<true, false><<<nblk, 256, 0, stream>>>(a, nblk);
    } else {
      if (relu_in) spmm_scalar<false, true><<<nblk, 256, 0, stream>>>(a, nblk);
      else spmm_scalar<false, false><<<nblk, 256, 0, stream>>>(a, nblk);
    }
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}


int tsgnn_csr_to_ell(const int* rowptr, const int* col, int64_t n_rows, int W, int* ell, int* tail_cnt, hipStream_t stream) {
  if (!rowptr || !ell || n_rows < 0 || (W != 4 && W != 8 && W != 16)) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  csr_to_ell_kernel<<<(unsigned)ceil_div64(n_rows * W, 256), 256, 0, stream>>>(rowptr, col, n_rows, W, ell, tail_cnt);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_csr_tail_fill(const int* rowptr, const int* col, const int* tail_ptr, int64_t n_rows, int W, int* tail_col,
                        hipStream_t stream) {
  if (!rowptr || !col || !tail_ptr || !tail_col || n_rows < 0 || W <= 0) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  csr_tail_fill<<<(unsigned)ceil_div64(n_rows, 256), 256, 0, stream>>>(rowptr, col, tail_ptr, n_rows, W, tail_col);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_ell_spmm_f32(const int* ell, int W, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows, int feat,
                       float self_scalar, hipStream_t stream) {
  if (!ell || !x || !y || n_rows < 0 || feat <= 0 || ldx < feat || ldy < feat) return TSGNN_EINVAL;
  if ((feat % 4) || (ldx % 4) || (ldy % 4) || feat > 256 || (reinterpret_cast<uintptr_t>(x) & 15) ||
      (reinterpret_cast<uintptr_t>(y) & 15) || (W != 4 && W != 8 && W != 16))
    return TSGNN_EUNSUPPORTED;
  if (n_rows == 0) return TSGNN_OK;
  const int nvec = feat / 4;
#define TSGNN_ELL(G, WW)                                                                               \
  {                                                                                                    \
    const unsigned nblk = (unsigned)ceil_div64(n_rows, 256 / G);                                       \
    TSGNN_KNAME("spmm_ell_vec4<%d,%d>", G, WW);                                                        \
    spmm_ell_vec4<G, WW><<<nblk, 256, 0, stream>>>(ell, x, ldx, y, ldy, n_rows, nvec, self_scalar, nblk); \
  }
  if (nvec <= 16) {
    if (W == 4) TSGNN_ELL(16, 4) else if (W == 8) TSGNN_ELL(16, 8) else TSGNN_ELL(16, 16)
  } else if (nvec <= 32) {
    if (W == 4) TSGNN_ELL(32, 4) else if (W == 8) TSGNN_ELL(32, 8) else TSGNN_ELL(32, 16)
  } else {
    if (W == 4) TSGNN_ELL(64, 4) else if (W == 8) TSGNN_ELL(64, 8) else TSGNN_ELL(64, 16)
  }
#undef TSGNN_ELL
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gcn_norm_f32(const int* rowptr, const int* col, const float* val, int64_t n_rows, float self_fill,
                       float* dinv, float* val_out, float* self_w, hipStream_t stream) {
  if (n_rows < 0 || !rowptr || !dinv || !val_out || !self_w) return TSGNN_EINVAL;
  if (n_rows == 0) return TSGNN_OK;
  const unsigned nblk = (unsigned)ceil_div64(n_rows, 256);
  gcn_deg_kernel<<<nblk, 256, 0, stream>>>(rowptr, col, val, n_rows, self_fill, dinv, self_w);
  gcn_norm_kernel<<<nblk, 256, 0, stream>>>(rowptr, col, val, dinv, n_rows, self_fill, val_out, self_w);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sddmm_rows_f32(const int* rowptr, const int* col, const float* dy, int64_t lddy, const float* x, int64_t ldx,
                         int64_t n_rows, int feat, float* dval, float* dself, hipStream_t stream) {
  if (n_rows < 0 || feat <= 0 || !rowptr || !col || !dy || !x || !dval || lddy < feat || ldx < feat) return TSGNN_EINVAL;
  if (feat > 1024) return TSGNN_EUNSUPPORTED;
  if (n_rows == 0) return TSGNN_OK;
  TSGNN_KNAME("sddmm_rows_kernel");
  sddmm_rows_kernel<<<(unsigned)ceil_div64(n_rows, 4), 256, 0, stream>>>(rowptr, col, dy, lddy, x, ldx, n_rows, feat, dval, dself);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
