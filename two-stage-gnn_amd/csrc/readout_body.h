// Body of the per-layer max-readout partial kernel (see sage_fused.hip), as a device function so that it can share a
// launch with the next layer's product (layer_fwd.hip).
#pragma once
#include "common.h"

namespace {

struct SlotArgs {
  const int* graph_ptr;
  const int* slot_count;
  int B, nmax;
  int64_t n_real;
  int n_ghost;
};

// ---------------------------------------------------------------------------------------------- readout
__device__ __forceinline__ unsigned long long pack_max(float val, unsigned r) {
  return ((unsigned long long)f32_ordered(val) << 32) | (unsigned long long)(0xFFFFFFFFu - r);
}
// grid (ceil(nslots/64), B); block = (256/G) row lanes x G float4 lanes (G = 32: F <= 128)
template <int G>
__device__ __forceinline__ void readout_partial_body(const SlotArgs& s, const float* __restrict__ x, int64_t ld, int F4,
                                                     unsigned long long* __restrict__ packed, unsigned bx, unsigned by,
                                                     unsigned long long* best_mem /* [256/G][4*G] */) {
  constexpr int RL = 256 / G;
  unsigned long long (*best)[4 * G] = reinterpret_cast<unsigned long long (*)[4 * G]>(best_mem);
  const int b = (int)by;
  const int c4 = threadIdx.x % G, rl = threadIdx.x / G;
  const int g0 = s.graph_ptr[b], sz = s.graph_ptr[b + 1] - g0;
  const int nslots = s.n_ghost ? s.nmax : sz;
  const int n_lo = (int)bx * 64, n_hi = min(nslots, n_lo + 64);
  unsigned long long m0 = 0ull, m1 = 0ull, m2 = 0ull, m3 = 0ull;
  if (c4 < F4) {
    for (int n = n_lo + rl; n < n_hi; n += RL) {
      const int64_t r = n < sz ? (int64_t)g0 + n : s.n_real + n;
      const float4 t = *reinterpret_cast<const float4*>(x + r * ld + 4 * c4);
      const unsigned long long p0 = pack_max(t.x, (unsigned)r), p1 = pack_max(t.y, (unsigned)r), p2 = pack_max(t.z, (unsigned)r),
                               p3 = pack_max(t.w, (unsigned)r);
      m0 = p0 > m0 ? p0 : m0; m1 = p1 > m1 ? p1 : m1; m2 = p2 > m2 ? p2 : m2; m3 = p3 > m3 ? p3 : m3;
    }
  }
  best[rl][4 * c4 + 0] = m0; best[rl][4 * c4 + 1] = m1; best[rl][4 * c4 + 2] = m2; best[rl][4 * c4 + 3] = m3;
  __syncthreads();
  const int F = 4 * F4;
  for (int f = threadIdx.x; f < F; f += 256) {
    unsigned long long m = best[0][f];
#pragma unroll
    for (int w = 1; w < RL; ++w) { const unsigned long long o = best[w][f]; m = o > m ? o : m; }
    if (m) atomicMax(&packed[(int64_t)b * F + f], m);
  }
}

// The same partial on a layer whose slot batch-norm was NOT materialised (rowgemm_body.h, STATS / BNIN): x holds the normalised
// pre-activations v, and  y = (relu(v) - mean[n]) * rstd[n]  is formed on the fly from the layer's integer sums; the blocks of
// graph 0 also leave mean / rstd of their 64 slots in global memory for the backward (slot_post_bwd reads them).
struct BnReadArgs {
  const unsigned long long* sums; const float* ghost; int F;      // F = feature width the statistics run over
  float* mean; float* rstd;
};
template <int G>
__device__ __forceinline__ void readout_partial_bn_body(const SlotArgs& s, const BnReadArgs& bn, const float* __restrict__ x, int64_t ld,
                                                        int F4, unsigned long long* __restrict__ packed, unsigned bx, unsigned by, int ch,
                                                        unsigned long long* best_mem /* [256/G][4*G] + 256 float2 behind it */) {
  // ch = slots per block (64, 128 or 256): the caller picks the smallest one that keeps [row panels + these blocks] within two
  // blocks per compute unit (registers allow two): a launch a little over that limit runs a third round for a handful of blocks
  constexpr int RL = 256 / G;
  unsigned long long (*best)[4 * G] = reinterpret_cast<unsigned long long (*)[4 * G]>(best_mem);
  float2* tab = reinterpret_cast<float2*>(best_mem + RL * 4 * G);
  const int b = (int)by;
  const int c4 = threadIdx.x % G, rl = threadIdx.x / G;
  const int g0 = s.graph_ptr[b], sz = s.graph_ptr[b + 1] - g0;
  const int nslots = s.n_ghost ? s.nmax : sz;
  const int n_lo = (int)bx * ch, n_hi = min(nslots, n_lo + ch);
  if ((int)threadIdx.x < ch) {
    const int n = min(n_lo + (int)threadIdx.x, s.nmax - 1);
    const ulonglong2 sm = *reinterpret_cast<const ulonglong2*>(bn.sums + 2 * n);
    const float2 ms = bn_stats_from_sums(sm.x, sm.y, s.slot_count[n], bn.ghost[0], bn.ghost[1], s.B, 1.0 / ((double)s.B * (double)bn.F), 1e-5f);
    tab[threadIdx.x] = ms;
    if (b == 0 && n_lo + (int)threadIdx.x < s.nmax) { bn.mean[n] = ms.x; bn.rstd[n] = ms.y; }
  }
  __syncthreads();
  unsigned long long m0 = 0ull, m1 = 0ull, m2 = 0ull, m3 = 0ull;
  if (c4 < F4) {
    for (int n = n_lo + rl; n < n_hi; n += RL) {
      const int64_t r = n < sz ? (int64_t)g0 + n : s.n_real + n;
      float4 t = *reinterpret_cast<const float4*>(x + r * ld + 4 * c4);
      const float2 ms = tab[n - n_lo];
      t.x = (fmaxf(t.x, 0.f) - ms.x) * ms.y; t.y = (fmaxf(t.y, 0.f) - ms.x) * ms.y;
      t.z = (fmaxf(t.z, 0.f) - ms.x) * ms.y; t.w = (fmaxf(t.w, 0.f) - ms.x) * ms.y;
      const unsigned long long p0 = pack_max(t.x, (unsigned)r), p1 = pack_max(t.y, (unsigned)r), p2 = pack_max(t.z, (unsigned)r),
                               p3 = pack_max(t.w, (unsigned)r);
      m0 = p0 > m0 ? p0 : m0; m1 = p1 > m1 ? p1 : m1; m2 = p2 > m2 ? p2 : m2; m3 = p3 > m3 ? p3 : m3;
    }
  }
  best[rl][4 * c4 + 0] = m0; best[rl][4 * c4 + 1] = m1; best[rl][4 * c4 + 2] = m2; best[rl][4 * c4 + 3] = m3;
  __syncthreads();
  const int F = 4 * F4;
  for (int f = threadIdx.x; f < F; f += 256) {
    unsigned long long m = best[0][f];
#pragma unroll
    for (int w = 1; w < RL; ++w) { const unsigned long long o = best[w][f]; m = o > m ? o : m; }
    if (m) atomicMax(&packed[(int64_t)b * F + f], m);
  }
}

}  // namespace
