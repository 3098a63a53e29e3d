// Fixed-order sum of the per-graph (or per-block) partial rows of the SAGPool score-layer gradients: part[nb][F + 4] (columns
// 0..F-1 = dw_s partials, column F = db_s partial).  A device function so that the launch that reduces the conv layer's weight
// gradient slabs can carry it as one extra block (gemm.hip) instead of a launch of its own (sagpool.hip otherwise).
#pragma once
#include "common.h"

namespace {

// 256 threads.  Block `bid` sums rows {q * stride : bid * chunk <= q < min(nb, (bid + 1) * chunk)}; dws == NULL: the sum goes
// back to the block's first row (first stage of the two-stage reduction of a large batch's partials).
__device__ __forceinline__ void du_reduce_body(float* __restrict__ part, int nb, int F, float* __restrict__ dws, float* __restrict__ dbs,
                                               int chunk, int stride, int bid, float4* s_part /* [256] LDS */) {
  const int nvec = F >> 2;
  const int c4 = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int q0 = bid * chunk, q1 = min(nb, q0 + chunk);
  for (int cb = 0; cb < nvec + 1; cb += 32) {                       // column nvec = the db_s partial (lane .x)
    const int c = cb + c4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c <= nvec) {
#pragma unroll 4
      for (int q = q0 + sl; q < q1; q += 8) {
        const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)q * stride * (F + 4) + 4 * c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    __syncthreads();
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && c <= nvec) {
#pragma unroll
      for (int q = 1; q < 8; ++q) {
        const float4 v = s_part[q * 32 + c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      if (dws == nullptr) {
        float* o = part + (int64_t)q0 * stride * (F + 4);
        if (c < nvec) *reinterpret_cast<float4*>(o + 4 * c) = s;
        else o[F] = s.x;
      } else {
        if (c < nvec) *reinterpret_cast<float4*>(dws + 4 * c) = s;
        else dbs[0] = s.x;
      }
    }
  }
}

}  // namespace
