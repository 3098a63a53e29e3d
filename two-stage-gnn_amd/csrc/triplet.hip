// Tail of the 2stg triplet step (tripletnet.py:35-45): the three graphs' embeddings  e_b = W r_b + bias  (map_model, one nn.Linear on
// the concatenated readouts) and the two pairwise distances  ||e_a - e_p + eps||, ||e_a - e_n + eps||  (F.pairwise_distance) in ONE
// launch forward and ONE launch backward.  With torch ops this tail was ~25 launches of a 80-launch step (three 1-row GEMMs of
// hipBLASLt, slices and their zero-filled gradients, sub / add / norm and their backward).
//
// Forward: one workgroup of 16 waves; a wave takes eight output columns at a time: lanes stride over the D inputs with
// 16-byte loads of W's rows (nn.Linear's [out, in] layout, eight rows requested together) and of the three readout rows, three wave
// sums per column; the distances from the finished embeddings in LDS.
// Backward: grid over slices of 32 input columns (no dependency between workgroups): every workgroup re-derives the embeddings'
// gradient de[3, E] (distance terms + the gradient that reaches the embeddings directly, e.g. the norm regularisers of
// train_triplet.py:262-263) and produces its columns of d_r = de W and of dW = de^T r; workgroup 0 also writes db.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

constexpr int TE_MAXE = 512;

// TE_EPW output columns per wave and pass: their W rows are requested together (one round trip for eight rows instead of one per row:
// the first version walked its rows one after the other and spent 12 us on 16 dependent trips)
constexpr int TE_EPW = 8;
__global__ __launch_bounds__(1024) void triplet_embed_fwd_kernel(const float* __restrict__ r, int64_t ldr, const float* __restrict__ w,
                                                                 int64_t ldw, const float* __restrict__ b, int D, int E, float eps,
                                                                 float* __restrict__ embed, float* __restrict__ dist) {
  __shared__ float es[3][TE_MAXE];
  __shared__ float red[2][16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int D4 = D >> 2;
  for (int e0 = wid * TE_EPW; e0 < E; e0 += 16 * TE_EPW) {
    float s0[TE_EPW], s1[TE_EPW], s2[TE_EPW];
#pragma unroll
    for (int j = 0; j < TE_EPW; ++j) { s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
    for (int c = lane; c < D4; c += 64) {
      float4 wv[TE_EPW];
#pragma unroll
      for (int j = 0; j < TE_EPW; ++j)                                  // rows past E: a mapped row, result dropped
        wv[j] = reinterpret_cast<const float4*>(w + (int64_t)min(e0 + j, E - 1) * ldw)[c];
      const float4 a = reinterpret_cast<const float4*>(r)[c];
      const float4 p = reinterpret_cast<const float4*>(r + ldr)[c];
      const float4 n = reinterpret_cast<const float4*>(r + 2 * ldr)[c];
#pragma unroll
      for (int j = 0; j < TE_EPW; ++j) {
        s0[j] = fmaf(wv[j].x, a.x, fmaf(wv[j].y, a.y, fmaf(wv[j].z, a.z, fmaf(wv[j].w, a.w, s0[j]))));
        s1[j] = fmaf(wv[j].x, p.x, fmaf(wv[j].y, p.y, fmaf(wv[j].z, p.z, fmaf(wv[j].w, p.w, s1[j]))));
        s2[j] = fmaf(wv[j].x, n.x, fmaf(wv[j].y, n.y, fmaf(wv[j].z, n.z, fmaf(wv[j].w, n.w, s2[j]))));
      }
    }
#pragma unroll
    for (int j = 0; j < TE_EPW; ++j) {
      const float t0 = wave_sum(s0[j]), t1 = wave_sum(s1[j]), t2 = wave_sum(s2[j]);
      const int e = e0 + j;
      if (lane == 0 && e < E) {
        const float bias = b ? b[e] : 0.f;
        es[0][e] = t0 + bias; es[1][e] = t1 + bias; es[2][e] = t2 + bias;
      }
    }
  }
  __syncthreads();
  float qp = 0.f, qn = 0.f;
  for (int e = threadIdx.x; e < E; e += 1024) {
    embed[e] = es[0][e]; embed[E + e] = es[1][e]; embed[2 * E + e] = es[2][e];
    const float dp = es[0][e] - es[1][e] + eps, dn = es[0][e] - es[2][e] + eps;
    qp = fmaf(dp, dp, qp); qn = fmaf(dn, dn, qn);
  }
  qp = wave_sum(qp); qn = wave_sum(qn);
  if (lane == 0) { red[0][wid] = qp; red[1][wid] = qn; }
  __syncthreads();
  if (threadIdx.x < 2) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[threadIdx.x][k];
    dist[threadIdx.x] = sqrtf(s);
  }
}

// workgroup = 32 input columns [32 bx, 32 bx + 32); thread (j = tid & 31, k = tid >> 5): column j, output rows e = k, k + 8, ...
__global__ __launch_bounds__(256) void triplet_embed_bwd_kernel(const float* __restrict__ r, int64_t ldr, const float* __restrict__ w,
                                                                int64_t ldw, int D, int E, float eps, const float* __restrict__ embed,
                                                                const float* __restrict__ dist, const float* __restrict__ d_dp,
                                                                const float* __restrict__ d_dn, const float* __restrict__ d_ea,
                                                                const float* __restrict__ d_ep, const float* __restrict__ d_en,
                                                                float* __restrict__ d_r, int64_t lddr,
                                                                float* __restrict__ dw, int64_t lddw, float* __restrict__ db) {
  __shared__ float de[3][TE_MAXE];
  __shared__ float part[3][8][32];
  const int tid = threadIdx.x;
  const float gp = d_dp ? d_dp[0] : 0.f, gn = d_dn ? d_dn[0] : 0.f;
  const float ip = dist[0] > 0.f ? gp / dist[0] : 0.f, in_ = dist[1] > 0.f ? gn / dist[1] : 0.f;   // (torch: 0 at a zero distance)
  for (int e = tid; e < E; e += 256) {
    const float a = embed[e], p = embed[E + e], n = embed[2 * E + e];
    const float tp = (a - p + eps) * ip, tn = (a - n + eps) * in_;
    de[0][e] = tp + tn + (d_ea ? d_ea[e] : 0.f);
    de[1][e] = -tp + (d_ep ? d_ep[e] : 0.f);
    de[2][e] = -tn + (d_en ? d_en[e] : 0.f);
  }
  __syncthreads();
  const int j = tid & 31, k = tid >> 5;
  const int d = 32 * (int)blockIdx.x + j;
  const bool ok = d < D;
  const float ra = ok ? r[d] : 0.f, rp = ok ? r[ldr + d] : 0.f, rn = ok ? r[2 * ldr + d] : 0.f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int dc = ok ? d : 0;
  for (int e0 = k; e0 < E; e0 += 64) {                               // eight rows of W per pass, requested together
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = w[(int64_t)min(e0 + 8 * j, E - 1) * ldw + dc];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = e0 + 8 * j;
      if (ok && e < E) {
        const float g0 = de[0][e], g1 = de[1][e], g2 = de[2][e];
        s0 = fmaf(g0, wv[j], s0); s1 = fmaf(g1, wv[j], s1); s2 = fmaf(g2, wv[j], s2);
        dw[(int64_t)e * lddw + d] = fmaf(g0, ra, fmaf(g1, rp, g2 * rn));
      }
    }
  }
  part[0][k][j] = s0; part[1][k][j] = s1; part[2][k][j] = s2;
  __syncthreads();
  if (tid < 96) {
    const int b = tid >> 5, jj = tid & 31, dd = 32 * (int)blockIdx.x + jj;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += part[b][q][jj];
    if (dd < D) d_r[(int64_t)b * lddr + dd] = s;
  }
  if (db && blockIdx.x == 0)
    for (int e = tid; e < E; e += 256) db[e] = (de[0][e] + de[1][e]) + de[2][e];
}

// torch.nn.MarginRankingLoss (train_triplet.py:235,277) on n pairs of distances: loss = reduce(max(0, -t (x1 - x2) + margin)), and
// the per-element coefficient of its gradient (dx1 = g coef, dx2 = -g coef) kept for the backward.  One workgroup: n is the number
// of triplets of a step (1 in the reference's loop); torch's composite runs ~9 element-wise launches forward and ~8 backward.
__global__ __launch_bounds__(256) void margin_rank_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                              const float* __restrict__ t, int64_t n, float margin, int mean,
                                                              float* __restrict__ loss, float* __restrict__ coef) {
  __shared__ float red[4];
  const float scale = mean ? 1.0f / (float)n : 1.0f;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float v = fmaf(-t[i], x1[i] - x2[i], margin);
    const bool on = v > 0.f;
    s += on ? v : 0.f;
    coef[i] = on ? -t[i] * scale : 0.f;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}
__global__ __launch_bounds__(256) void margin_rank_bwd_kernel(const float* __restrict__ g, const float* __restrict__ coef, int64_t n,
                                                              float* __restrict__ dx1, float* __restrict__ dx2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = g[0] * coef[i];
  if (dx1) dx1[i] = v;
  if (dx2) dx2[i] = -v;
}

}  // namespace

extern "C" {

int tsgnn_margin_rank_fwd_f32(const float* x1, const float* x2, const float* target, int64_t n, float margin, int mean, float* loss,
                              float* coef, hipStream_t stream) {
  if (!x1 || !x2 || !target || !loss || !coef || n <= 0) return TSGNN_EINVAL;
  TSGNN_KNAME("margin_rank_fwd_kernel");
  margin_rank_fwd_kernel<<<1, 256, 0, stream>>>(x1, x2, target, n, margin, mean, loss, coef);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_margin_rank_bwd_f32(const float* g, const float* coef, int64_t n, float* dx1, float* dx2, hipStream_t stream) {
  if (!g || !coef || n <= 0) return TSGNN_EINVAL;
  TSGNN_KNAME("margin_rank_bwd_kernel");
  margin_rank_bwd_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(g, coef, n, dx1, dx2);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_triplet_embed_fwd_f32(const float* r, int64_t ldr, const float* w, int64_t ldw, const float* b, int D, int E, float eps,
                                float* embed, float* dist, hipStream_t stream) {
  if (!r || !w || !embed || !dist || D <= 0 || E <= 0 || ldr < D || ldw < D) return TSGNN_EINVAL;
  if (E > TE_MAXE || (D % 4) || (ldr % 4) || (ldw % 4) || ((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(w)) & 15))
    return TSGNN_EUNSUPPORTED;
  TSGNN_KNAME("triplet_embed_fwd_kernel");
  triplet_embed_fwd_kernel<<<1, 1024, 0, stream>>>(r, ldr, w, ldw, b, D, E, eps, embed, dist);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_triplet_embed_bwd_f32(const float* r, int64_t ldr, const float* w, int64_t ldw, int D, int E, float eps, const float* embed,
                                const float* dist, const float* d_dp, const float* d_dn, const float* d_ea, const float* d_ep,
                                const float* d_en, float* d_r, int64_t lddr, float* dw, int64_t lddw, float* db, hipStream_t stream) {
  if (!r || !w || !embed || !dist || !d_r || !dw || D <= 0 || E <= 0 || ldr < D || ldw < D || lddr < D || lddw < D) return TSGNN_EINVAL;
  if (E > TE_MAXE) return TSGNN_EUNSUPPORTED;
  TSGNN_KNAME("triplet_embed_bwd_kernel");
  triplet_embed_bwd_kernel<<<(unsigned)((D + 31) / 32), 256, 0, stream>>>(r, ldr, w, ldw, D, E, eps, embed, dist, d_dp, d_dn, d_ea, d_ep, d_en, d_r,
                                                                         lddr, dw, lddw, db);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
