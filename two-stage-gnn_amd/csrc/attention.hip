// Edge-softmax attention (SURVEY §8 a6/a7 and GATConv of a15) without ever materialising the
// reference's [N,N,2F] pair tensor (encoders_GAT.py:35) or its dense [N,N] attention matrix.
//
//   score of entry e in softmax group g (a CSR row):  t = s_grp[g,h] + s_oth[col[e] % mod, h]
//   alpha[e,h] = softmax over the group's entries of LeakyReLU(t)
//
// Reference DGATHead (encoders_GAT.py:36-43): softmax(dim=1) normalises over the ROW index i for each
// column j (trap T3) -> groups are the rows of A^T, s_grp = a2.h_j, s_oth = a1.h_i; the aggregation
// out_i = sum_j alpha_ij h_j then runs on A with the permuted alphas.  PyG GATConv: groups are the
// target rows of A, s_grp = att_r.h_i, s_oth = att_l.h_j, same-CSR aggregation.
// One sub-wave lane group per CSR row; per-node scalars only in the softmax pass (HBM-light), feature
// rows are gathered once, by the head-weighted SpMM.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

__device__ __forceinline__ float lrelu(float t, float slope) { return t > 0.f ? t : slope * t; }

// per-node scalars s[r,h] = sum_f x[r, h*Fh + f] * a[h*lda + f]    (a1.h_i / a2.h_j for every head)
// (a2 / s2 nullable: both score vectors of a GAT head, a1.h_i and a2.h_j, from ONE pass over the rows)
__global__ __launch_bounds__(256) void node_scores_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int H, int Fh,
                                                          const float* __restrict__ a, int64_t lda, float* __restrict__ s,
                                                          const float* __restrict__ a2, int64_t lda2, float* __restrict__ s2) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  for (int h = 0; h < H; ++h) {
    float acc = 0.f, acc2 = 0.f;
    for (int f = lane; f < Fh; f += 64) {
      const float xv = x[r * ldx + (int64_t)h * Fh + f];
      acc = fmaf(xv, a[(int64_t)h * lda + f], acc);
      if (a2) acc2 = fmaf(xv, a2[(int64_t)h * lda2 + f], acc2);
    }
    acc = wave_sum(acc);
    if (a2) acc2 = wave_sum(acc2);
    if (lane == 0) {
      s[r * H + h] = acc;
      if (a2) s2[r * H + h] = acc2;
    }
  }
}

// the same with 16-byte loads: lane l holds floats 4l..4l+3 of the row (all heads at once, H * Fh <= 256), the Fh / 4 lanes of a
// head reduce with DPP inside their 16-lane row — one load per row instead of H dependent-on-nothing but serial passes
template <int LPH>
__global__ __launch_bounds__(256) void node_scores_vec4_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int H, int Fh,
                                                               const float* __restrict__ a, int64_t lda, float* __restrict__ s,
                                                               const float* __restrict__ a2, int64_t lda2, float* __restrict__ s2) {
  const int lane = threadIdx.x & 63;
  const int64_t r0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool rok = r0 < rows;                               // whole waves stay active for the DPP reductions
  const int64_t r = rok ? r0 : 0;
  const int h = lane / LPH, f = 4 * (lane % LPH);
  const bool live = h < H;
  const float4 xv = live ? *reinterpret_cast<const float4*>(x + r * ldx + (int64_t)h * Fh + f) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 av = live ? *reinterpret_cast<const float4*>(a + (int64_t)h * lda + f) : make_float4(0.f, 0.f, 0.f, 0.f);
  float acc = (xv.x * av.x + xv.y * av.y) + (xv.z * av.z + xv.w * av.w);
  acc = group_sum<LPH>(acc);
  float acc2 = 0.f;
  if (a2) {
    const float4 bv = live ? *reinterpret_cast<const float4*>(a2 + (int64_t)h * lda2 + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc2 = group_sum<LPH>((xv.x * bv.x + xv.y * bv.y) + (xv.z * bv.z + xv.w * bv.w));
  }
  if (rok && live && (lane % LPH) == 0) {
    s[r * H + h] = acc;
    if (a2) s2[r * H + h] = acc2;
  }
}

inline bool node_scores_vec4_ok(const float* x, int64_t ldx, int H, int Fh, const float* a, int64_t lda, const float* a2, int64_t lda2) {
  const int lph = Fh / 4;
  if (Fh % 4 || (lph != 1 && lph != 2 && lph != 4 && lph != 8 && lph != 16) || H * lph > 64) return false;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(a2)) & 15) return false;
  return ldx % 4 == 0 && lda % 4 == 0 && (!a2 || lda2 % 4 == 0);
}

template <class... A>
inline void launch_node_scores_vec4(int lph, unsigned grid, hipStream_t stream, A... args) {
  switch (lph) {
    case 1: node_scores_vec4_kernel<1><<<grid, 256, 0, stream>>>(args...); break;
    case 2: node_scores_vec4_kernel<2><<<grid, 256, 0, stream>>>(args...); break;
    case 4: node_scores_vec4_kernel<4><<<grid, 256, 0, stream>>>(args...); break;
    case 8: node_scores_vec4_kernel<8><<<grid, 256, 0, stream>>>(args...); break;
    default: node_scores_vec4_kernel<16><<<grid, 256, 0, stream>>>(args...); break;
  }
}

// EG lanes per (row, head): numerically-stable softmax over the row's entries.  A DD row has ~5 entries: one pass of the lane
// group covers it with two dependent round trips (entry ids -> their scores); a thread walking the row alone paid those two
// trips per entry, three times over (max, sum, normalise).  The first entry of every lane stays in registers; rows longer
// than EG entries take the strided loops.
constexpr int SM_EG = 8;
__global__ __launch_bounds__(256) void edge_softmax_fwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int64_t rows,
                                                               int H, const float* __restrict__ s_grp, const float* __restrict__ s_oth,
                                                               int mod, float slope, float* __restrict__ alpha) {
  const int sub = threadIdx.x & (SM_EG - 1);
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SM_EG;      // (row, head) pair of this lane group
  if (i >= rows * H) return;
  const int64_t g = i / H;
  const int h = (int)(i % H);
  const int e0 = rowptr[g], e1 = rowptr[g + 1];
  if (e0 == e1) return;
  const float sg = s_grp[(mod ? g % mod : g) * H + h];
  auto score = [&](int e) {
    const int64_t c = mod ? col[e] % mod : col[e];
    return lrelu(sg + s_oth[c * H + h], slope);
  };
  const int ef = e0 + sub;
  const bool has = ef < e1;
  const float tf = has ? score(ef) : -INFINITY;
  float m = tf;
  for (int e = ef + SM_EG; e < e1; e += SM_EG) m = fmaxf(m, score(e));
  m = group_max<SM_EG>(m);
  const float pf = has ? expf(tf - m) : 0.f;
  float d = pf;
  for (int e = ef + SM_EG; e < e1; e += SM_EG) d += expf(score(e) - m);
  d = group_sum<SM_EG>(d);
  const float inv = 1.f / d;
  if (has) alpha[(int64_t)ef * H + h] = pf * inv;
  for (int e = ef + SM_EG; e < e1; e += SM_EG) alpha[(int64_t)e * H + h] = expf(score(e) - m) * inv;
}

// softmax + LeakyReLU backward inside a group:  dt[e] = alpha (dalpha - sum alpha dalpha) * lrelu'(t)
// also ds_grp[g,h] = sum_e dt[e,h]                      (same lane-group layout)
__global__ __launch_bounds__(256) void edge_softmax_bwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int64_t rows,
                                                               int H, const float* __restrict__ s_grp, const float* __restrict__ s_oth,
                                                               int mod, float slope, const float* __restrict__ alpha,
                                                               const float* __restrict__ dalpha, float* __restrict__ dt,
                                                               float* __restrict__ ds_grp) {
  const int sub = threadIdx.x & (SM_EG - 1);
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SM_EG;
  if (i >= rows * H) return;
  const int64_t g = i / H;
  const int h = (int)(i % H);
  const int e0 = rowptr[g], e1 = rowptr[g + 1];
  const int ef = e0 + sub;
  const bool has = ef < e1;
  const float af = has ? alpha[(int64_t)ef * H + h] : 0.f, daf = has ? dalpha[(int64_t)ef * H + h] : 0.f;
  float c = af * daf;
  for (int e = ef + SM_EG; e < e1; e += SM_EG) c = fmaf(alpha[(int64_t)e * H + h], dalpha[(int64_t)e * H + h], c);
  c = group_sum<SM_EG>(c);
  const float sg = (e0 < e1) ? s_grp[(mod ? g % mod : g) * H + h] : 0.f;
  float acc = 0.f;
  if (has) {
    const int64_t cc = mod ? col[ef] % mod : col[ef];
    const float t = sg + s_oth[cc * H + h];
    const float v = af * (daf - c) * (t > 0.f ? 1.f : slope);
    dt[(int64_t)ef * H + h] = v;
    acc = v;
  }
  for (int e = ef + SM_EG; e < e1; e += SM_EG) {
    const int64_t cc = mod ? col[e] % mod : col[e];
    const float t = sg + s_oth[cc * H + h];
    const float v = alpha[(int64_t)e * H + h] * (dalpha[(int64_t)e * H + h] - c) * (t > 0.f ? 1.f : slope);
    dt[(int64_t)e * H + h] = v;
    acc += v;
  }
  acc = group_sum<SM_EG>(acc);
  if (sub == 0) ds_grp[i] = acc;
}

// out[p, :] = src[perm[p], :]   (edge-value permutation between a CSR and its transpose)
__global__ void gather_rows_small(const float* __restrict__ src, const int* __restrict__ perm, int64_t n, int H, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * H) return;
  dst[i] = src[(int64_t)perm[i / H] * H + (i % H)];
}
__global__ void scatter_rows_small(const float* __restrict__ src, const int* __restrict__ perm, int64_t n, int H, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * H) return;
  dst[(int64_t)perm[i / H] * H + (i % H)] = src[i];
}
// out[r, h] = sum over the row's entries of val[e, h]
// eperm (nullable): entry e's value lives at val[eperm[e]] (values stored in the transposed structure's entry order)
__global__ __launch_bounds__(256) void csr_row_sum_kernel(const int* __restrict__ rowptr, const float* __restrict__ val, int64_t rows, int H,
                                                          float* __restrict__ out, const int* __restrict__ eperm) {
  // SM_EG lanes per (row, head), as the edge softmax: entry -> (permuted) value is two dependent round trips per ROW, not per entry
  const int sub = threadIdx.x & (SM_EG - 1);
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SM_EG;
  if (i >= rows * H) return;
  const int64_t r = i / H;
  const int h = (int)(i % H);
  const int e1 = rowptr[r + 1];
  float acc = 0.f;
  for (int e = rowptr[r] + sub; e < e1; e += SM_EG) acc += val[(int64_t)(eperm ? eperm[e] : e) * H + h];
  acc = group_sum<SM_EG>(acc);
  if (sub == 0) out[i] = acc;
}

// optional epilogue of the head-weighted aggregation (folds the element-wise passes that used to follow it):
//   y[r, h*Fh+f] += w1[r,h] * a1[h,f] + w2[r,h] * a2[h,f] + uscale * (uw ? uw[r,h] : 1) * u[r / rows_per_seg, h*Fh+f]
// (score-gradient outer products ds (x) a of dh, and the uniform 1/N term of all-masked softmax columns)
struct HeadsEpi {
  const float* w1; const float* a1; int64_t lda1;
  const float* w2; const float* a2; int64_t lda2;
  const float* u; int64_t ldu; const float* uw; int rows_per_seg; float uscale;
  const int* row_seg;      // nullable: segment (graph) of every row for ragged batches; else r / rows_per_seg
  const int* eperm;        // nullable: entry e's weights live at alpha[eperm[e]] (alpha stored in the transposed entry order)
};
__device__ __forceinline__ float heads_epi(const HeadsEpi& e, int64_t r, int H, int Fh, int h, int f /*within head*/) {
  float add = 0.f;
  if (e.w1) add = fmaf(e.w1[r * H + h], e.a1[(int64_t)h * e.lda1 + f], add);
  if (e.w2) add = fmaf(e.w2[r * H + h], e.a2[(int64_t)h * e.lda2 + f], add);
  if (e.u) add = fmaf(e.uscale * (e.uw ? e.uw[r * H + h] : 1.f), e.u[(e.row_seg ? (int64_t)e.row_seg[r] : r / e.rows_per_seg) * e.ldu + (int64_t)h * Fh + f], add);
  return add;
}

// head-weighted aggregation: y[r, h*Fh+f] = sum_e alpha[e,h] * x[col[e] % mod, h*Fh+f];  one wave per row
__global__ __launch_bounds__(256) void spmm_heads_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                         const float* __restrict__ alpha, int H, int Fh,
                                                         const float* __restrict__ x, int64_t ldx, int mod,
                                                         float* __restrict__ y, int64_t ldy, int64_t rows, unsigned nblk, HeadsEpi epi) {
  const unsigned lb = xcd_remap(blockIdx.x, nblk);
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)lb * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int F = H * Fh;
  const int e0 = rowptr[r], e1 = rowptr[r + 1];
  for (int fb = 0; fb < F; fb += 64) {
    const int f = fb + lane;
    const bool live = f < F;
    const int h = live ? f / Fh : 0;
    float acc = 0.f;
    for (int eb = e0; eb < e1; eb += 64) {
      const int me = eb + lane;
      const int cj = (me < e1) ? (mod ? col[me] % mod : col[me]) : 0;
      const int cnt = min(64, e1 - eb);
      for (int k = 0; k < cnt; ++k) {
        const int j = __shfl(cj, k, 64);
        if (live) acc = fmaf(alpha[(int64_t)(epi.eperm ? epi.eperm[eb + k] : eb + k) * H + h], x[(int64_t)j * ldx + f], acc);
      }
    }
    if (live) y[r * ldy + f] = acc + heads_epi(epi, r, H, Fh, h, f - h * Fh);
  }
}

// float4 variants (Fh % 4 == 0, F = H*Fh <= 256, 16-byte rows): lane l of a row's lane group owns features 4l..4l+3 (one head,
// since Fh % 4 == 0), G = F/4 lanes per row (64 / G rows per wave), edges in batches of 8 with independent 16-byte loads.
template <int G>
__global__ __launch_bounds__(256) void spmm_heads_vec4(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                       const float* __restrict__ alpha, int H, int Fh, const float* __restrict__ x,
                                                       int64_t ldx, int mod, float* __restrict__ y, int64_t ldy, int64_t rows, int nvec,
                                                       HeadsEpi epi) {
  const int lig = threadIdx.x % G;
  // XCD-aware row order (blocks b, b + 8, ... share an L2): neighbouring rows gather the same x rows
  const int64_t r = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * (256 / G) + threadIdx.x / G;
  if (r >= rows) return;
  const bool live = lig < nvec;
  const int h = live ? (4 * lig) / Fh : 0;
  const int64_t co = live ? 4 * lig : 0;
  const int e0 = rowptr[r], e1 = rowptr[r + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int eb = e0; eb < e1; eb += 8) {
    float4 v[8];
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = eb + k;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      a[k] = 0.f;
      if (e < e1) {
        const int j = mod ? col[e] % mod : col[e];
        v[k] = *reinterpret_cast<const float4*>(x + (int64_t)j * ldx + co);
        a[k] = alpha[(int64_t)(epi.eperm ? epi.eperm[e] : e) * H + h];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc.x = fmaf(a[k], v[k].x, acc.x); acc.y = fmaf(a[k], v[k].y, acc.y);
      acc.z = fmaf(a[k], v[k].z, acc.z); acc.w = fmaf(a[k], v[k].w, acc.w);
    }
  }
  if (live) {
    const int f0 = (int)co - h * Fh;
    acc.x += heads_epi(epi, r, H, Fh, h, f0); acc.y += heads_epi(epi, r, H, Fh, h, f0 + 1);
    acc.z += heads_epi(epi, r, H, Fh, h, f0 + 2); acc.w += heads_epi(epi, r, H, Fh, h, f0 + 3);
    *reinterpret_cast<float4*>(y + r * ldy + co) = acc;
  }
}

// SDDMM, float4: the Fh/4 lanes of a head reduce their partial dot products with DPP inside a 16-lane row
// (Fh/4 in {1,2,4,8,16}); one lane per head writes dalpha[e, h]
template <int G>
__global__ __launch_bounds__(256) void sddmm_heads_vec4(const int* __restrict__ rowptr, const int* __restrict__ col, int H, int Fh,
                                                        const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                        int64_t ldx, int mod, float* __restrict__ dalpha, int64_t rows, int nvec) {
  const int lig = threadIdx.x % G;
  const int64_t r0 = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * (256 / G) + threadIdx.x / G;   // XCD-aware row order, as the aggregation
  const bool rok = r0 < rows;                           // whole waves stay active: the DPP reductions need every lane
  const int64_t r = rok ? r0 : 0;
  const bool live = lig < nvec;
  const int lph = Fh / 4;                               // lanes per head
  const int h = live ? lig / lph : 0;
  const int64_t co = live ? 4 * lig : 0;
  const int e0 = rok ? rowptr[r] : 0, e1 = rok ? rowptr[r + 1] : 0;
  const float4 d = live ? *reinterpret_cast<const float4*>(dy + r * lddy + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  int emax = e1 - e0;                                   // uniform trip count over the wave
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) emax = max(emax, __shfl_xor(emax, o, 64));
  for (int eb = 0; eb < emax; eb += 4) {
    float p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = e0 + eb + k;
      p[k] = 0.f;
      if (e < e1 && live) {
        const int j = mod ? col[e] % mod : col[e];
        const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)j * ldx + co);
        p[k] = (d.x * v.x + d.y * v.y) + (d.z * v.z + d.w * v.w);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = p[k];
      if (lph >= 2) t += dpp_f32<0xB1>(t);
      if (lph >= 4) t += dpp_f32<0x4E>(t);
      if (lph >= 8) t += dpp_f32<0x141>(t);              // row_half_mirror: with the two quad steps = all 8 lanes
      if (lph >= 16) t += dpp_f32<0x140>(t);             // row_mirror: all 16 lanes
      const int e = e0 + eb + k;
      if (live && e < e1 && (lig % lph) == 0) dalpha[(int64_t)e * H + h] = t;
    }
  }
}

// SDDMM: dalpha[e,h] = dot(dy[r, head h], x[col[e] % mod, head h]) for every entry e of row r; one wave per row
__global__ __launch_bounds__(256) void sddmm_heads_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int H, int Fh,
                                                          const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                          int64_t ldx, int mod, float* __restrict__ dalpha, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int e0 = rowptr[r], e1 = rowptr[r + 1];
  for (int e = e0; e < e1; ++e) {
    const int64_t j = mod ? col[e] % mod : col[e];
    for (int h = 0; h < H; ++h) {
      float acc = 0.f;
      for (int f = lane; f < Fh; f += 64) acc = fmaf(dy[r * lddy + (int64_t)h * Fh + f], x[j * ldx + (int64_t)h * Fh + f], acc);
      acc = wave_sum(acc);
      if (lane == 0) dalpha[(int64_t)e * H + h] = acc;
    }
  }
}

// out[s, c] = scale * sum_{r in segment s} w[r, c / Fh] * x[r, c]   (w nullable = 1; mean: / segment length).
// Two levels so that one long segment (all rows of a 1000-node graph) still fills the chip: grid (ceil(C/64), S, chunks of
// 128 rows) writes partials, segment_wsum_final adds the chunks in fixed order (reproducible, no float atomics).
constexpr int SEG_CHUNK = 128;
// w2 / part2 (nullable pair): a second weighted sum of the same rows in the same pass (x is read once)
__global__ __launch_bounds__(256) void segment_wsum_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, int H,
                                                           int Fh, const int* __restrict__ seg_ptr, int64_t rows_if_one, int C,
                                                           float* __restrict__ part, int nseg, const float* __restrict__ w2,
                                                           float* __restrict__ part2) {
  __shared__ float lds[4][64];
  __shared__ float lds2[4][64];
  const int s = blockIdx.y;
  const int64_t s0 = seg_ptr ? seg_ptr[s] : 0, s1 = seg_ptr ? seg_ptr[s + 1] : rows_if_one;
  const int64_t r0 = s0 + (int64_t)blockIdx.z * SEG_CHUNK, r1 = min(s1, r0 + SEG_CHUNK);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float acc = 0.f, acc2 = 0.f;
  if (c < C && r0 < r1) {
    const int h = c / Fh;
    for (int64_t rb = r0 + rl; rb < r1; rb += 32) {       // eight independent row loads in flight per thread
      float xv[8], wv[8], wv2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = rb + 4 * u;
        const bool ok = r < r1;
        xv[u] = ok ? x[r * ldx + c] : 0.f;
        wv[u] = ok ? (w ? w[r * H + h] : 1.f) : 0.f;
        wv2[u] = (ok && w2) ? w2[r * H + h] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc = fmaf(wv[u], xv[u], acc); acc2 = fmaf(wv2[u], xv[u], acc2); }
    }
  }
  lds[rl][threadIdx.x & 63] = acc;
  lds2[rl][threadIdx.x & 63] = acc2;
  __syncthreads();
  if (rl == 0 && c < C) {
    part[((int64_t)blockIdx.z * nseg + s) * C + c] = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
    if (part2)
      part2[((int64_t)blockIdx.z * nseg + s) * C + c] = (lds2[0][threadIdx.x] + lds2[1][threadIdx.x]) + (lds2[2][threadIdx.x] + lds2[3][threadIdx.x]);
  }
}
// grid (ceil(C / 64), nseg), 64 columns x 16 chunk lanes: lane q adds chunks q, q + 16, ... (a 32,000-row segment has 250
// chunks: one thread per output summed them in a 32 us latency chain), then the 16 lanes are added in order.
__global__ __launch_bounds__(1024) void segment_wsum_final(const float* __restrict__ part, int nchunk, int nseg, int C,
                                                           const int* __restrict__ seg_ptr, int64_t rows_if_one, float scale, int mean,
                                                           float* __restrict__ out, int64_t ldo, const float* __restrict__ part2,
                                                           float* __restrict__ out2) {
  __shared__ float lds[16][64];
  if (blockIdx.z == 1) { part = part2; out = out2; }             // the second sum of a two-weight call
  const int s = blockIdx.y, cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t len = seg_ptr ? (int64_t)(seg_ptr[s + 1] - seg_ptr[s]) : rows_if_one;
  const int used = min((int)((len + SEG_CHUNK - 1) / SEG_CHUNK), nchunk);
  float acc = 0.f;
  if (c < C)
    for (int k = q; k < used; k += 16) acc += part[((int64_t)k * nseg + s) * C + c];
  lds[q][cl] = acc;
  __syncthreads();
  if (q != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) acc += lds[k][cl];
  float sc = scale;
  if (mean) sc = len > 0 ? scale / (float)len : 0.f;
  out[(int64_t)s * ldo + c] = sc * acc;
}

// out[s, c] = scale * sum_{e in [seg_ptr[s], seg_ptr[s+1])} w[e] * x[idx[e], c]: weighted sum of a FEW listed rows per segment (the
// isolated / padded columns of a graph in the uniform softmax term) without scanning all rows.  One block per segment.
__global__ __launch_bounds__(256) void gather_wsum_kernel(const float* __restrict__ x, int64_t ldx, const int* __restrict__ idx,
                                                          const float* __restrict__ w, const int* __restrict__ seg_ptr, int C,
                                                          float scale, float* __restrict__ out, int64_t ldo) {
  const int s = blockIdx.x;
  const int e0 = seg_ptr[s], e1 = seg_ptr[s + 1];
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
#pragma unroll 4
    for (int e = e0; e < e1; ++e) acc = fmaf(w[e], x[(int64_t)idx[e] * ldx + c], acc);
    out[(int64_t)s * ldo + c] = scale * acc;
  }
}

// y[r, c] (+)= alpha * w[r, c/Fh] * v[seg(r)?, c] ... generic rank-1 style broadcast add:
//   mode 0: y[r,c] += w[r, c/Fh] * a[(c/Fh)*lda + c%Fh]         (ds (x) a  terms of dh)
//   mode 1: y[r,c] += w[r, c/Fh] * u[seg_of_row(r), c]           (uniform term; w nullable = 1)
__global__ void broadcast_add_kernel(float* __restrict__ y, int64_t ldy, int64_t rows, int H, int Fh, const float* __restrict__ w,
                                     const float* __restrict__ a, int64_t lda, const float* __restrict__ u, int64_t ldu,
                                     int rows_per_seg, float scale) {
  const int C = H * Fh;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const int64_t r = i / C;
  const int c = (int)(i % C), h = c / Fh;
  const float wv = w ? w[r * H + h] : 1.f;
  float add;
  if (a) add = wv * a[(int64_t)h * lda + (c - h * Fh)];
  else add = wv * u[(r / rows_per_seg) * ldu + c];
  y[r * ldy + c] += scale * add;
}

// ELU forward / backward (encoders_GAT.py:47,83), optional mean over heads first (concat=False, :78-83)
// concat heads (no mean over heads): plain element-wise ELU, four elements per thread
__global__ __launch_bounds__(256) void elu_vec4_fwd_kernel(const float* __restrict__ x, int64_t n4, int apply_elu, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<const float4*>(x)[i];
  if (apply_elu) {
    v.x = v.x <= 0.f ? expm1f(v.x) : v.x; v.y = v.y <= 0.f ? expm1f(v.y) : v.y;
    v.z = v.z <= 0.f ? expm1f(v.z) : v.z; v.w = v.w <= 0.f ? expm1f(v.w) : v.w;
  }
  reinterpret_cast<float4*>(y)[i] = v;
}
__global__ __launch_bounds__(256) void elu_vec4_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t n4,
                                                           int apply_elu, float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  float4 g = reinterpret_cast<const float4*>(dy)[i];
  if (apply_elu) {
    g.x *= v.x <= 0.f ? expf(v.x) : 1.f; g.y *= v.y <= 0.f ? expf(v.y) : 1.f;
    g.z *= v.z <= 0.f ? expf(v.z) : 1.f; g.w *= v.w <= 0.f ? expf(v.w) : 1.f;
  }
  reinterpret_cast<float4*>(dx)[i] = g;
}
__global__ void elu_heads_fwd_kernel(const float* __restrict__ x, int64_t rows, int H, int Fh, int mean_heads, int apply_elu,
                                     float* __restrict__ y) {
  const int Co = mean_heads ? Fh : H * Fh;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Co) return;
  const int64_t r = i / Co;
  const int c = (int)(i % Co);
  float v;
  if (mean_heads) {
    v = 0.f;
    for (int h = 0; h < H; ++h) v += x[r * (int64_t)(H * Fh) + (int64_t)h * Fh + c];
    v /= (float)H;
  } else v = x[i];
  y[i] = (apply_elu && v <= 0.f) ? expm1f(v) : v;
}
__global__ void elu_heads_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t rows, int H, int Fh,
                                     int mean_heads, int apply_elu, float* __restrict__ dx) {
  const int C = H * Fh;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const int64_t r = i / C;
  const int c = (int)(i % C);
  if (mean_heads) {
    const int f = c % Fh;
    float v = 0.f;
    for (int h = 0; h < H; ++h) v += x[r * (int64_t)C + (int64_t)h * Fh + f];
    v /= (float)H;
    const float g = dy[r * Fh + f] * ((apply_elu && v <= 0.f) ? expf(v) : 1.f);
    dx[i] = g / (float)H;
  } else {
    const float v = x[i];
    dx[i] = dy[i] * ((apply_elu && v <= 0.f) ? expf(v) : 1.f);
  }
}

// ---------------------------------------------------------------- parameters of a layer's heads <-> fused operands
// forward:  W[i, h*Fo + f] = w_h[i, f]   (the [Fin, H*Fo] operand of the projection h = x W, encoders_GAT.py:32)
//           A[s, h, f]     = a_h[s*Fo + f]   (s = 0: the a1 half that meets h_i, s = 1: the a2 half, :34-36)
// backward: gw[h, i, f] = dW[i, h*Fo + f];  ga[h, s*Fo + f] = dA_s[h, f]   — one launch each way instead of cat / permute / copy
// launches per tensor kind (the reference keeps one w and one a per head module: attention_{i}.w / .a)
constexpr int PACK_HMAX = 8;
struct HeadPtrs { const float* w[PACK_HMAX]; const float* a[PACK_HMAX]; };
__global__ __launch_bounds__(256) void pack_heads_kernel(HeadPtrs p, int H, int Fin, int Fo, float* __restrict__ W, float* __restrict__ A) {
  const int64_t nw = (int64_t)Fin * H * Fo, na = (int64_t)2 * H * Fo;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nw) {
    const int c = (int)(i % (H * Fo)), r = (int)(i / (H * Fo));
    const int h = c / Fo, f = c % Fo;
    W[i] = p.w[h][(int64_t)r * Fo + f];
  } else if (i < nw + na) {
    const int64_t j = i - nw;
    const int f = (int)(j % Fo), h = (int)((j / Fo) % H), sidx = (int)(j / ((int64_t)H * Fo));
    A[j] = p.a[h][sidx * Fo + f];
  }
}
__global__ __launch_bounds__(256) void unpack_heads_kernel(const float* __restrict__ dW, const float* __restrict__ dA0,
                                                           const float* __restrict__ dA1, int H, int Fin, int Fo,
                                                           float* __restrict__ gw, float* __restrict__ ga) {
  const int64_t nw = (int64_t)Fin * H * Fo, na = (int64_t)2 * H * Fo;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nw) {                                             // i indexes gw[h][r][f]: contiguous stores
    const int f = (int)(i % Fo), r = (int)((i / Fo) % Fin), h = (int)(i / ((int64_t)Fin * Fo));
    gw[i] = dW ? dW[(int64_t)r * H * Fo + (int64_t)h * Fo + f] : 0.f;
  } else if (i < nw + na) {
    const int64_t j = i - nw;                               // ga[h][s*Fo + f]
    const int f = (int)(j % Fo), sidx = (int)((j / Fo) % 2), h = (int)(j / (2 * (int64_t)Fo));
    const float* src = sidx ? dA1 : dA0;
    ga[j] = src ? src[(int64_t)h * Fo + f] : 0.f;
  }
}

}  // namespace

extern "C" {

int tsgnn_node_scores_f32(const float* x, int64_t ldx, int64_t rows, int H, int Fh, const float* a, int64_t lda, float* s,
                          tsgnn_stream_t stream) {
  if (!x || !a || !s || rows < 0 || H <= 0 || Fh <= 0 || ldx < (int64_t)H * Fh) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  if (node_scores_vec4_ok(x, ldx, H, Fh, a, lda, nullptr, 0))
    launch_node_scores_vec4(Fh / 4, (unsigned)ceil_div64(rows, 4), stream, x, ldx, rows, H, Fh, a, lda, s, (const float*)nullptr, (int64_t)0,
                            (float*)nullptr);
  else
    node_scores_kernel<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(x, ldx, rows, H, Fh, a, lda, s, nullptr, 0, nullptr);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_node_scores2_f32(const float* x, int64_t ldx, int64_t rows, int H, int Fh, const float* a1, int64_t lda1, float* s1,
                           const float* a2, int64_t lda2, float* s2, tsgnn_stream_t stream) {
  if (!x || !a1 || !s1 || !a2 || !s2 || rows < 0 || H <= 0 || Fh <= 0 || ldx < (int64_t)H * Fh) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  if (node_scores_vec4_ok(x, ldx, H, Fh, a1, lda1, a2, lda2))
    launch_node_scores_vec4(Fh / 4, (unsigned)ceil_div64(rows, 4), stream, x, ldx, rows, H, Fh, a1, lda1, s1, a2, lda2, s2);
  else
    node_scores_kernel<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(x, ldx, rows, H, Fh, a1, lda1, s1, a2, lda2, s2);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_edge_softmax_fwd_f32(const int* rowptr, const int* col, int64_t rows, int H, const float* s_grp, const float* s_oth,
                               int mod, float slope, float* alpha, tsgnn_stream_t stream) {
  if (!rowptr || !s_grp || !s_oth || !alpha || rows < 0 || H <= 0 || mod < 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  edge_softmax_fwd_kernel<<<(unsigned)ceil_div64(rows * H * SM_EG, 256), 256, 0, stream>>>(rowptr, col, rows, H, s_grp, s_oth, mod, slope, alpha);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_edge_softmax_bwd_f32(const int* rowptr, const int* col, int64_t rows, int H, const float* s_grp, const float* s_oth,
                               int mod, float slope, const float* alpha, const float* dalpha, float* dt, float* ds_grp,
                               tsgnn_stream_t stream) {
  if (!rowptr || !s_grp || !s_oth || !alpha || !dalpha || !dt || !ds_grp || rows < 0 || H <= 0 || mod < 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  edge_softmax_bwd_kernel<<<(unsigned)ceil_div64(rows * H * SM_EG, 256), 256, 0, stream>>>(rowptr, col, rows, H, s_grp, s_oth, mod, slope,
                                                                                  alpha, dalpha, dt, ds_grp);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_edge_permute_f32(const float* src, const int* perm, int64_t n, int H, int scatter, float* dst, tsgnn_stream_t stream) {
  if (!src || !perm || !dst || n < 0 || H <= 0) return TSGNN_EINVAL;
  if (n == 0) return TSGNN_OK;
  if (scatter) scatter_rows_small<<<(unsigned)ceil_div64(n * H, 256), 256, 0, stream>>>(src, perm, n, H, dst);
  else gather_rows_small<<<(unsigned)ceil_div64(n * H, 256), 256, 0, stream>>>(src, perm, n, H, dst);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_csr_row_sum_f32(const int* rowptr, const float* val, int64_t rows, int H, float* out, tsgnn_stream_t stream) {
  if (!rowptr || !val || !out || rows < 0 || H <= 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  csr_row_sum_kernel<<<(unsigned)ceil_div64(rows * H * SM_EG, 256), 256, 0, stream>>>(rowptr, val, rows, H, out, nullptr);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_csr_row_sum_perm_f32(const int* rowptr, const float* val, const int* eperm, int64_t rows, int H, float* out,
                               tsgnn_stream_t stream) {
  if (!rowptr || !val || !eperm || !out || rows < 0 || H <= 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  csr_row_sum_kernel<<<(unsigned)ceil_div64(rows * H * SM_EG, 256), 256, 0, stream>>>(rowptr, val, rows, H, out, eperm);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

static int spmm_heads_launch(const int* rowptr, const int* col, const float* alpha, int H, int Fh, const float* x, int64_t ldx,
                             int mod, float* y, int64_t ldy, int64_t rows, const HeadsEpi& epi, tsgnn_stream_t stream) {
  if (!rowptr || !alpha || !x || !y || rows < 0 || H <= 0 || Fh <= 0 || mod < 0 || ldx < (int64_t)H * Fh || ldy < (int64_t)H * Fh)
    return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  const int F = H * Fh;
  if ((Fh % 4) == 0 && F <= 256 && (ldx % 4) == 0 && (ldy % 4) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    const int nvec = F / 4;
    if (nvec <= 16) spmm_heads_vec4<16><<<(unsigned)ceil_div64(rows, 16), 256, 0, stream>>>(rowptr, col, alpha, H, Fh, x, ldx, mod, y, ldy, rows, nvec, epi);
    else if (nvec <= 32) spmm_heads_vec4<32><<<(unsigned)ceil_div64(rows, 8), 256, 0, stream>>>(rowptr, col, alpha, H, Fh, x, ldx, mod, y, ldy, rows, nvec, epi);
    else spmm_heads_vec4<64><<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(rowptr, col, alpha, H, Fh, x, ldx, mod, y, ldy, rows, nvec, epi);
    TSGNN_CHECK_LAUNCH();
    return TSGNN_OK;
  }
  const unsigned nblk = (unsigned)ceil_div64(rows, 4);
  spmm_heads_kernel<<<nblk, 256, 0, stream>>>(rowptr, col, alpha, H, Fh, x, ldx, mod, y, ldy, rows, nblk, epi);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_csr_spmm_heads_f32(const int* rowptr, const int* col, const float* alpha, int H, int Fh, const float* x, int64_t ldx,
                             int mod, float* y, int64_t ldy, int64_t rows, tsgnn_stream_t stream) {
  HeadsEpi epi{};
  return spmm_heads_launch(rowptr, col, alpha, H, Fh, x, ldx, mod, y, ldy, rows, epi, stream);
}

int tsgnn_csr_spmm_heads_epi_f32(const int* rowptr, const int* col, const float* alpha, int H, int Fh, const float* x, int64_t ldx,
                                 int mod, float* y, int64_t ldy, int64_t rows, const float* w1, const float* a1, int64_t lda1,
                                 const float* w2, const float* a2, int64_t lda2, const float* u, int64_t ldu, const float* uw,
                                 int rows_per_seg, const int* row_seg, float uscale, const int* eperm, tsgnn_stream_t stream) {
  if ((w1 && !a1) || (w2 && !a2) || (u && rows_per_seg <= 0 && !row_seg)) return TSGNN_EINVAL;
  HeadsEpi epi{w1, a1, lda1, w2, a2, lda2, u, ldu, uw, rows_per_seg, uscale, row_seg, eperm};
  return spmm_heads_launch(rowptr, col, alpha, H, Fh, x, ldx, mod, y, ldy, rows, epi, stream);
}

int tsgnn_csr_sddmm_heads_f32(const int* rowptr, const int* col, int H, int Fh, const float* dy, int64_t lddy, const float* x,
                              int64_t ldx, int mod, float* dalpha, int64_t rows, tsgnn_stream_t stream) {
  if (!rowptr || !dy || !x || !dalpha || rows < 0 || H <= 0 || Fh <= 0 || mod < 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  const int F = H * Fh, lph = Fh / 4;
  if ((Fh % 4) == 0 && F <= 256 && (lph == 1 || lph == 2 || lph == 4 || lph == 8 || lph == 16) && (ldx % 4) == 0 && (lddy % 4) == 0 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
    const int nvec = F / 4;
    // lane groups of G lanes start at multiples of G >= 16, so a head's lanes never straddle a 16-lane DPP row
    if (nvec <= 16) sddmm_heads_vec4<16><<<(unsigned)ceil_div64(rows, 16), 256, 0, stream>>>(rowptr, col, H, Fh, dy, lddy, x, ldx, mod, dalpha, rows, nvec);
    else if (nvec <= 32) sddmm_heads_vec4<32><<<(unsigned)ceil_div64(rows, 8), 256, 0, stream>>>(rowptr, col, H, Fh, dy, lddy, x, ldx, mod, dalpha, rows, nvec);
    else sddmm_heads_vec4<64><<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(rowptr, col, H, Fh, dy, lddy, x, ldx, mod, dalpha, rows, nvec);
    TSGNN_CHECK_LAUNCH();
    return TSGNN_OK;
  }
  sddmm_heads_kernel<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(rowptr, col, H, Fh, dy, lddy, x, ldx, mod, dalpha, rows);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_segment_wsum_f32(const float* x, int64_t ldx, const float* w, int H, int Fh, const int* seg_ptr, int nseg, int64_t rows,
                           int64_t max_seg, float scale, int mean, float* ws, float* out, int64_t ldo, tsgnn_stream_t stream) {
  if (!x || !out || !ws || H <= 0 || Fh <= 0 || nseg <= 0 || rows < 0 || max_seg < 0 || ldx < (int64_t)H * Fh || ldo < (int64_t)H * Fh)
    return TSGNN_EINVAL;
  if (!seg_ptr && nseg != 1) return TSGNN_EINVAL;
  const int C = H * Fh;
  const int64_t longest = seg_ptr ? max_seg : rows;
  const int nchunk = (int)(longest > 0 ? ceil_div64(longest, SEG_CHUNK) : 1);
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)nseg, (unsigned)nchunk);
  segment_wsum_kernel<<<grid, 256, 0, stream>>>(x, ldx, w, H, Fh, seg_ptr, rows, C, ws, nseg, nullptr, nullptr);
  segment_wsum_final<<<dim3((unsigned)((C + 63) / 64), (unsigned)nseg), 1024, 0, stream>>>(ws, nchunk, nseg, C, seg_ptr, rows, scale, mean, out, ldo,
                                                                                           nullptr, nullptr);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* two weighted sums of the same rows in one pass (x read once): out1 = sum w1 x, out2 = sum w2 x; ws: twice the floats of
 * tsgnn_segment_wsum_f32 */
int tsgnn_segment_wsum2_f32(const float* x, int64_t ldx, const float* w1, const float* w2, int H, int Fh, const int* seg_ptr, int nseg,
                            int64_t rows, int64_t max_seg, float scale, float* ws, float* out1, float* out2, int64_t ldo,
                            tsgnn_stream_t stream) {
  if (!x || !w1 || !w2 || !out1 || !out2 || !ws || H <= 0 || Fh <= 0 || nseg <= 0 || rows < 0 || max_seg < 0 || ldx < (int64_t)H * Fh ||
      ldo < (int64_t)H * Fh)
    return TSGNN_EINVAL;
  if (!seg_ptr && nseg != 1) return TSGNN_EINVAL;
  const int C = H * Fh;
  const int64_t longest = seg_ptr ? max_seg : rows;
  const int nchunk = (int)(longest > 0 ? ceil_div64(longest, SEG_CHUNK) : 1);
  float* ws2 = ws + (size_t)nchunk * nseg * C;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)nseg, (unsigned)nchunk);
  segment_wsum_kernel<<<grid, 256, 0, stream>>>(x, ldx, w1, H, Fh, seg_ptr, rows, C, ws, nseg, w2, ws2);
  segment_wsum_final<<<dim3((unsigned)((C + 63) / 64), (unsigned)nseg, 2), 1024, 0, stream>>>(ws, nchunk, nseg, C, seg_ptr, rows, scale, 0, out1,
                                                                                              ldo, ws2, out2);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gather_wsum_f32(const float* x, int64_t ldx, const int* idx, const float* w, const int* seg_ptr, int nseg, int C, float scale,
                          float* out, int64_t ldo, tsgnn_stream_t stream) {
  if (!x || !idx || !w || !seg_ptr || !out || nseg <= 0 || C <= 0 || ldx < C || ldo < C) return TSGNN_EINVAL;
  gather_wsum_kernel<<<(unsigned)nseg, 256, 0, stream>>>(x, ldx, idx, w, seg_ptr, C, scale, out, ldo);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_broadcast_add_f32(float* y, int64_t ldy, int64_t rows, int H, int Fh, const float* w, const float* a, int64_t lda,
                            const float* u, int64_t ldu, int rows_per_seg, float scale, tsgnn_stream_t stream) {
  if (!y || rows < 0 || H <= 0 || Fh <= 0 || ((a == nullptr) == (u == nullptr)) || (u && rows_per_seg <= 0)) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  broadcast_add_kernel<<<(unsigned)ceil_div64(rows * H * Fh, 256), 256, 0, stream>>>(y, ldy, rows, H, Fh, w, a, lda, u, ldu,
                                                                                    rows_per_seg, scale);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_elu_heads_fwd_f32(const float* x, int64_t rows, int H, int Fh, int mean_heads, int apply_elu, float* y, tsgnn_stream_t stream) {
  if (!x || !y || rows < 0 || H <= 0 || Fh <= 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  const int Co = mean_heads ? Fh : H * Fh;
  const int64_t n = rows * (int64_t)H * Fh;
  if (!mean_heads && n % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
    elu_vec4_fwd_kernel<<<(unsigned)ceil_div64(n / 4, 256), 256, 0, stream>>>(x, n / 4, apply_elu, y);
  else
    elu_heads_fwd_kernel<<<(unsigned)ceil_div64(rows * Co, 256), 256, 0, stream>>>(x, rows, H, Fh, mean_heads, apply_elu, y);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_elu_heads_bwd_f32(const float* x, const float* dy, int64_t rows, int H, int Fh, int mean_heads, int apply_elu, float* dx,
                            tsgnn_stream_t stream) {
  if (!x || !dy || !dx || rows < 0 || H <= 0 || Fh <= 0) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  const int64_t n = rows * (int64_t)H * Fh;
  if (!mean_heads && n % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0)
    elu_vec4_bwd_kernel<<<(unsigned)ceil_div64(n / 4, 256), 256, 0, stream>>>(x, dy, n / 4, apply_elu, dx);
  else
    elu_heads_bwd_kernel<<<(unsigned)ceil_div64(rows * H * Fh, 256), 256, 0, stream>>>(x, dy, rows, H, Fh, mean_heads, apply_elu, dx);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_pack_heads_max(void) { return PACK_HMAX; }

int tsgnn_pack_heads_f32(const float* w0, const float* w1, const float* w2, const float* w3, const float* w4, const float* w5,
                         const float* w6, const float* w7, const float* a0, const float* a1, const float* a2, const float* a3,
                         const float* a4, const float* a5, const float* a6, const float* a7, int H, int Fin, int Fo, float* W, float* A,
                         tsgnn_stream_t stream) {
  if (H <= 0 || H > PACK_HMAX || Fin <= 0 || Fo <= 0 || !W || !A) return TSGNN_EINVAL;
  HeadPtrs p{{w0, w1, w2, w3, w4, w5, w6, w7}, {a0, a1, a2, a3, a4, a5, a6, a7}};
  for (int h = 0; h < H; ++h)
    if (!p.w[h] || !p.a[h]) return TSGNN_EINVAL;
  const int64_t n = (int64_t)Fin * H * Fo + (int64_t)2 * H * Fo;
  pack_heads_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(p, H, Fin, Fo, W, A);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_unpack_heads_f32(const float* dW, const float* dA0, const float* dA1, int H, int Fin, int Fo, float* gw, float* ga,
                           tsgnn_stream_t stream) {
  if (H <= 0 || Fin <= 0 || Fo <= 0 || !gw || !ga) return TSGNN_EINVAL;
  const int64_t n = (int64_t)Fin * H * Fo + (int64_t)2 * H * Fo;
  unpack_heads_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(dW, dA0, dA1, H, Fin, Fo, gw, ga);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
