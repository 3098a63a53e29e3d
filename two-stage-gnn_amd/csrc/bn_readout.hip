// Per-node-slot batch norm and node-axis max readout of GcnEncoderGraph.
//
//  * apply_bn (encoders.py:134-138) builds a FRESH BatchNorm1d(Nmax) per call: channel = node slot n,
//    statistics over (graph, feature), biased variance, eps 1e-5, gamma=1, beta=0, always batch stats.
//    Here rows are laid out graph after graph (graph_ptr); slot n of graph b is row graph_ptr[b]+n when
//    n < size_b.  The padded "ghost" rows of the reference (slots n >= size_b) all carry one value per
//    slot; they are represented by ONE row per slot (row n_real + n) with multiplicity
//    ghost_mult[n] = B - slot_count[n]  (DESIGN.md §ghost rows).  ReLU (encoders.py:179) is folded in:
//    the kernels consume relu(v).
//  * readout torch.max(x, dim=1) (encoders.py:183,190,197) INCLUDES ghost rows (trap T5): every graph
//    has exactly nmax candidate slots; ties resolve to the smallest slot (first occurrence).
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

constexpr float BN_EPS = 1e-5f;

struct SlotArgs {
  const int* graph_ptr;      // [B+1]
  const int* slot_count;     // [nmax] graphs that have slot n (real row)
  int B, nmax;
  int64_t n_real;
  int n_ghost;               // 0 or nmax
};

__device__ __forceinline__ float act(float v, int relu) { return relu ? fmaxf(v, 0.f) : v; }

// block reduce of two floats over 256 threads
__device__ __forceinline__ void block_sum2(float& a, float& b, float* lds /*8 floats*/) {
  a = wave_sum(a); b = wave_sum(b);
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { lds[wid] = a; lds[4 + wid] = b; }
  __syncthreads();
  a = lds[0] + lds[1] + lds[2] + lds[3];
  b = lds[4] + lds[5] + lds[6] + lds[7];
}

// one block per slot: two-pass mean / variance over the slot's rows (L2-hot re-read)
__global__ __launch_bounds__(256) void bn_slot_stats(SlotArgs s, const float* __restrict__ v, int64_t ld, int F,
                                                     int relu, float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ float lds[8];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int have = s.slot_count[n];
  const float gm = s.n_ghost ? (float)(s.B - have) : 0.f;      // ghost multiplicity
  const float cnt = ((float)have + gm) * (float)F;
  const float* grow = v + (s.n_real + n) * ld;
  float s1 = 0.f, dummy = 0.f;
  for (int b = wid; b < s.B; b += 4) {
    const int g0 = s.graph_ptr[b];
    if (s.graph_ptr[b + 1] - g0 > n) {
      const float* row = v + (int64_t)(g0 + n) * ld;
      for (int f = lane; f < F; f += 64) s1 += act(row[f], relu);
    }
  }
  if (wid == 0 && gm > 0.f) for (int f = lane; f < F; f += 64) s1 += gm * act(grow[f], relu);
  block_sum2(s1, dummy, lds);
  const float mu = cnt > 0.f ? s1 / cnt : 0.f;
  float s2 = 0.f;
  dummy = 0.f;
  for (int b = wid; b < s.B; b += 4) {
    const int g0 = s.graph_ptr[b];
    if (s.graph_ptr[b + 1] - g0 > n) {
      const float* row = v + (int64_t)(g0 + n) * ld;
      for (int f = lane; f < F; f += 64) { const float d = act(row[f], relu) - mu; s2 = fmaf(d, d, s2); }
    }
  }
  if (wid == 0 && gm > 0.f) for (int f = lane; f < F; f += 64) { const float d = act(grow[f], relu) - mu; s2 = fmaf(gm * d, d, s2); }
  block_sum2(s2, dummy, lds);
  if (threadIdx.x == 0) {
    const float var = cnt > 0.f ? s2 / cnt : 0.f;
    mean[n] = mu;
    rstd[n] = 1.0f / sqrtf(var + BN_EPS);
  }
}

// xhat[r,:] = (act(v[r,:]) - mean[slot]) * rstd[slot]; one wave per row
__global__ __launch_bounds__(256) void bn_slot_apply(const float* __restrict__ v, int64_t ldv, const int* __restrict__ row_slot,
                                                     int64_t n_real, int64_t rows, int F, int relu,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float* __restrict__ y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int n = r < n_real ? row_slot[r] : (int)(r - n_real);
  const float mu = mean ? mean[n] : 0.f, rs = rstd ? rstd[n] : 1.f;
  for (int f = lane; f < F; f += 64) y[r * ldy + f] = (act(v[r * ldv + f], relu) - mu) * rs;
}

// backward stats per slot: m1 = sum dy / cnt, m2 = sum dy*xhat / cnt  (ghost rows: plain sums — their
// dy is already the sum over the copies they stand for)
__global__ __launch_bounds__(256) void bn_slot_bwd_stats(SlotArgs s, const float* __restrict__ v, int64_t ldv,
                                                         const float* __restrict__ dy, int64_t lddy, int F, int relu,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ m1, float* __restrict__ m2) {
  __shared__ float lds[8];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int have = s.slot_count[n];
  const float gm = s.n_ghost ? (float)(s.B - have) : 0.f;
  const float cnt = ((float)have + gm) * (float)F;
  const float mu = mean[n], rs = rstd[n];
  float a = 0.f, c = 0.f;
  for (int b = wid; b < s.B; b += 4) {
    const int g0 = s.graph_ptr[b];
    if (s.graph_ptr[b + 1] - g0 > n) {
      const int64_t r = g0 + n;
      for (int f = lane; f < F; f += 64) {
        const float d = dy[r * lddy + f];
        a += d;
        c = fmaf(d, (act(v[r * ldv + f], relu) - mu) * rs, c);
      }
    }
  }
  if (wid == 0 && s.n_ghost) {
    const int64_t r = s.n_real + n;
    for (int f = lane; f < F; f += 64) {
      const float d = dy[r * lddy + f];
      a += d;
      c = fmaf(d, (act(v[r * ldv + f], relu) - mu) * rs, c);
    }
  }
  block_sum2(a, c, lds);
  if (threadIdx.x == 0) { m1[n] = cnt > 0.f ? a / cnt : 0.f; m2[n] = cnt > 0.f ? c / cnt : 0.f; }
}

// dv[r,:] = relu'(v) * rstd * (dy - w_r (m1 + xhat m2)),  w_r = 1 (real) or ghost multiplicity
__global__ __launch_bounds__(256) void bn_slot_bwd_apply(SlotArgs s, const float* __restrict__ v, int64_t ldv,
                                                         const float* __restrict__ dy, int64_t lddy,
                                                         const int* __restrict__ row_slot, int64_t rows, int F, int relu,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ m1, const float* __restrict__ m2,
                                                         float* __restrict__ dv, int64_t lddv) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bool ghost = r >= s.n_real;
  const int n = ghost ? (int)(r - s.n_real) : row_slot[r];
  float w = 1.f, mu = 0.f, rs = 1.f, a1 = 0.f, a2 = 0.f;
  if (mean) {
    if (ghost) w = (float)(s.B - s.slot_count[n]);
    mu = mean[n]; rs = rstd[n]; a1 = m1[n]; a2 = m2[n];
  }
  for (int f = lane; f < F; f += 64) {
    const float x = v[r * ldv + f];
    const float xh = (act(x, relu) - mu) * rs;
    float g = rs * (dy[r * lddy + f] - w * (a1 + xh * a2));
    if (relu && !(x > 0.f)) g = 0.f;
    dv[r * lddv + f] = g;
  }
}

// ---------------------------------------------------------------- per-graph statistics (B = 1 semantics, batched)
// With ONE graph per forward (the 2stg / 2stg+ triplet settings call the encoder at B = 1, tripletnet.py:36-38) the
// slot batch-norm degenerates to a per-row layer norm over the features (trap T2).  Row mode lets anchor / positive /
// negative (or any number of graphs) share one launch while each keeps its own B = 1 statistics.  One wave per row.
__global__ __launch_bounds__(256) void row_ln_fwd(const float* __restrict__ v, int64_t ldv, int64_t rows, int F, int relu,
                                                  float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s1 = 0.f;
  for (int f = lane; f < F; f += 64) s1 += act(v[r * ldv + f], relu);
  const float mu = wave_sum(s1) / (float)F;
  float s2 = 0.f;
  for (int f = lane; f < F; f += 64) { const float d = act(v[r * ldv + f], relu) - mu; s2 = fmaf(d, d, s2); }
  const float rs = 1.0f / sqrtf(wave_sum(s2) / (float)F + BN_EPS);
  for (int f = lane; f < F; f += 64) y[r * ldy + f] = (act(v[r * ldv + f], relu) - mu) * rs;
  if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}
__global__ __launch_bounds__(256) void row_ln_bwd(const float* __restrict__ v, int64_t ldv, const float* __restrict__ dy, int64_t lddy,
                                                  int64_t rows, int F, int relu, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, float* __restrict__ dv, int64_t lddv) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float mu = mean[r], rs = rstd[r];
  float a = 0.f, c = 0.f;
  for (int f = lane; f < F; f += 64) {
    const float d = dy[r * lddy + f];
    a += d;
    c = fmaf(d, (act(v[r * ldv + f], relu) - mu) * rs, c);
  }
  const float m1 = wave_sum(a) / (float)F, m2 = wave_sum(c) / (float)F;
  for (int f = lane; f < F; f += 64) {
    const float x = v[r * ldv + f];
    float g = rs * (dy[r * lddy + f] - m1 - (act(x, relu) - mu) * rs * m2);
    if (relu && !(x > 0.f)) g = 0.f;
    dv[r * lddv + f] = g;
  }
}

// Backward of [max readout ; per-row layer norm ; ReLU ; L2 normalise] of a hidden GraphConv layer in ONE pass when the batch-norm
// statistics are per graph (B = 1 semantics, row_ln_fwd): the row-local counterpart of slot_post_bwd (sage_fused.hip).
//   dy(r, f) = dxs[r, f]                                   (real rows: what the next layer's input gradient brought; a ghost row has no edges)
//            + sum_b (arg[b, f] == r ? dout[b, f] : 0)     (max-readout winners: a real row can only win in its own graph, a ghost row —
//                                                            the first padded slot of every graph of its size — in any)
//   LN:  dv = rstd (dy - mean(dy) - xhat mean(dy xhat)) ; ReLU mask ;  L2:  du = rinv (dv - v <v, dv>)   (rinv >= 1e12: clamped norm)
// One wave per row, F <= 256 (four values per lane in registers).
__global__ __launch_bounds__(256) void row_post_bwd(const int* __restrict__ row_graph, int B, int64_t n_real, int64_t rows,
                                                    const float* __restrict__ v, int64_t ldv, const float* __restrict__ dxs, int64_t lddxs,
                                                    const float* __restrict__ dout, int64_t ldo, const int* __restrict__ arg, int F,
                                                    int relu, int ln, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ rinv, float* __restrict__ du, int64_t lddu) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bool ghost = r >= n_real;
  const int b0 = ghost ? 0 : row_graph[r], b1 = ghost ? B : b0 + 1;
  const float mu = ln ? mean[r] : 0.f, rs = ln ? rstd[r] : 1.f;
  const float ri = rinv[r];
  float x[4], dy[4];
  float a = 0.f, c = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = lane + 64 * q;
    x[q] = 0.f; dy[q] = 0.f;
    if (f < F) {
      x[q] = v[r * ldv + f];
      float d = (dxs && !ghost) ? dxs[r * lddxs + f] : 0.f;
      if (arg)
        for (int b = b0; b < b1; ++b)
          if ((int64_t)arg[(int64_t)b * F + f] == r) d += dout[(int64_t)b * ldo + f];
      dy[q] = d;
      a += d;
      c = fmaf(d, (act(x[q], relu) - mu) * rs, c);
    }
  }
  float dot = 0.f;
  if (ln) {
    const float m1 = wave_sum(a) / (float)F, m2 = wave_sum(c) / (float)F;
#pragma unroll
    for (int q = 0; q < 4; ++q) dy[q] = rs * (dy[q] - m1 - (act(x[q], relu) - mu) * rs * m2);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (relu && !(x[q] > 0.f)) dy[q] = 0.f;
    if (lane + 64 * q >= F) dy[q] = 0.f;
    dot = fmaf(x[q], dy[q], dot);
  }
  dot = ri >= 0.999e12f ? 0.f : wave_sum(dot);            // |u| < eps: F.normalize's clamp_min passes no norm gradient
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = lane + 64 * q;
    if (f < F) du[r * lddu + f] = ri * (dy[q] - x[q] * dot);
  }
}

// ---------------------------------------------------------------- max readout over node slots
// grid (chunks of 64 slots, B); block = 4 waves x 16 slots, lanes over features.  ONE launch: every block folds the packed
// (value, row) maxima of its chunk into packed[b, :] with device-scope atomicMax (performed at the memory side: coherent across
// the XCDs' L2s without any cache write-back — an agent-scope fence here cost 10 us: it writes back whatever the previous launches
// left dirty in the L2), waits for the atomics' RETURN values, and takes a ticket of its graph; the block that draws the last ticket
// reads the graph's maxima back with atomic loads, decodes them into (out, arg) and leaves packed[b, :] and the ticket counter at
// zero for the next call.  max is order-free, so the result does not depend on which block comes last.
__global__ __launch_bounds__(256) void readout_max_chunks(SlotArgs s, const float* __restrict__ x, int64_t ld, int F, int relu,
                                                          unsigned long long* part, unsigned* count, float* __restrict__ out,
                                                          int64_t ldo, int* __restrict__ arg) {
  extern __shared__ unsigned long long best_lds[];          // [4][FP]
  __shared__ unsigned ticket;
  const int b = blockIdx.y;
  const int nch = gridDim.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g0 = s.graph_ptr[b];
  const int sz = s.graph_ptr[b + 1] - g0;
  const int nslots = s.n_ghost ? s.nmax : sz;
  const int n_lo = blockIdx.x * 64, n_hi = min(nslots, n_lo + 64);
  const int FP = (F + 63) & ~63;
  for (int fb = 0; fb < F; fb += 64) {
    const int f = fb + lane;
    unsigned long long best = 0ull;
    if (f < F) {
      // the wave's (up to) 16 slots of this chunk: all loads in flight at once (one at a time, the loop was a chain of 16
      // dependent round trips per 64 features: 12 us for a 5,000-row batch)
      float val[16];
      int64_t rr[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int n = n_lo + wid + 4 * u;
        rr[u] = n < sz ? (int64_t)g0 + n : s.n_real + n;
        val[u] = n < n_hi ? x[rr[u] * ld + f] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (n_lo + wid + 4 * u < n_hi) {
          const unsigned long long p = ((unsigned long long)f32_ordered(act(val[u], relu)) << 32) |
                                       (unsigned long long)(0xFFFFFFFFu - (unsigned)rr[u]);
          best = p > best ? p : best;
        }
      }
    }
    best_lds[wid * FP + fb + lane] = best;
  }
  __syncthreads();
  unsigned long long* mine = part + (int64_t)b * F;
  unsigned long long seen = 0ull;
  for (int f = threadIdx.x; f < F; f += 256) {
    unsigned long long m = best_lds[f];
#pragma unroll
    for (int w = 1; w < 4; ++w) { const unsigned long long o = best_lds[w * FP + f]; m = o > m ? o : m; }
    if (m) seen |= atomicMax(&mine[f], m);                  // the RETURNING form: once the value is here the atomic has been performed
  }
  asm volatile("" ::"v"((unsigned)(seen >> 32)), "v"((unsigned)seen));   // every return value is in its register before the barrier
  __syncthreads();
  if (threadIdx.x == 0) ticket = atomicAdd(&count[b], 1u);
  __syncthreads();
  if (ticket != (unsigned)(nch - 1)) return;
  if (threadIdx.x == 0) atomicExch(&count[b], 0u);          // nobody else touches it any more in this launch
  for (int f = threadIdx.x; f < F; f += 256) {
    const unsigned long long m = __hip_atomic_exchange(&mine[f], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[(int64_t)b * ldo + f] = m ? ordered_f32((unsigned)(m >> 32)) : 0.f;
    arg[(int64_t)b * F + f] = m ? (int)(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull)) : -1;
  }
}
// nmax <= 64 (the pooled DiffPool levels: 64- and 8-node graphs): one chunk per graph, so neither the zeroed packed buffer,
// nor atomics, nor the decode launch are needed — block b scans graph b's slots, thread per feature, sixteen slots in flight
__global__ __launch_bounds__(256) void readout_max_direct(SlotArgs s, const float* __restrict__ x, int64_t ld, int F, int relu,
                                                          float* __restrict__ out, int64_t ldo, int* __restrict__ arg) {
  const int b = blockIdx.x;
  const int g0 = s.graph_ptr[b];
  const int sz = s.graph_ptr[b + 1] - g0;
  const int nslots = s.n_ghost ? s.nmax : sz;
  for (int f = threadIdx.x; f < F; f += 256) {
    unsigned long long best = 0ull;
    for (int n0 = 0; n0 < nslots; n0 += 16) {
      float val[16];
      int64_t rr[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int n = min(n0 + u, nslots - 1);                 // clamped: the eight requests are unconditional (a load inside its
        rr[u] = n < sz ? (int64_t)g0 + n : s.n_real + n;       // own `n < nslots ?` was waited for before the next was issued)
        val[u] = x[rr[u] * ld + f];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (n0 + u < nslots) {
          const unsigned long long p = ((unsigned long long)f32_ordered(act(val[u], relu)) << 32) |
                                       (unsigned long long)(0xFFFFFFFFu - (unsigned)rr[u]);
          best = p > best ? p : best;
        }
      }
    }
    out[(int64_t)b * ldo + f] = best ? ordered_f32((unsigned)(best >> 32)) : 0.f;
    arg[(int64_t)b * F + f] = best ? (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull)) : -1;
  }
}
// dx[arg[b,f], f] += dout[b,f]   (ghost rows can be chosen by several graphs -> atomic)
__global__ void readout_max_bwd(const float* __restrict__ dout, int64_t ldo, const int* __restrict__ arg, int B, int F,
                                const float* __restrict__ x, int64_t ldx_in, int relu, int64_t n_real,
                                float* __restrict__ dx, int64_t ldx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * F) return;
  const int b = (int)(i / F), f = (int)(i % F);
  const int r = arg[i];
  if (r < 0) return;
  if (relu && !(x[(int64_t)r * ldx_in + f] > 0.f)) return;
  const float g = dout[(int64_t)b * ldo + f];
  if (r >= n_real) atomicAdd(&dx[(int64_t)r * ldx + f], g);
  else dx[(int64_t)r * ldx + f] += g;
}

// the same as a dense pass for batches WITHOUT ghost rows: every element of dx is written (no zero fill beforehand, no atomics):
// dx[r, f] = (arg[graph(r), f] == r ? dout[graph(r), f] : 0) + (add ? add[r, f] : 0);  rows [rows, rows_total) (ghost rows whose
// gradient the caller discards, or none) get add[r, f] or 0.  `add` is the gradient that reaches the same tensor through its
// other consumer (the DiffPool contraction reads the embeddings the readout reads): one pass instead of fill + scatter + add.
__global__ void readout_max_bwd_rows(const float* __restrict__ dout, int64_t ldo, const int* __restrict__ arg, const int* __restrict__ row_graph,
                                     int F4, int64_t rows, int64_t rows_total, const float* __restrict__ add, int64_t ldadd,
                                     float* __restrict__ dx, int64_t ldx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_total * F4) return;
  const int64_t r = i / F4;
  const int c = 4 * (int)(i - r * F4);
  float4 o = add ? *reinterpret_cast<const float4*>(add + r * ldadd + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < rows) {
    const int b = row_graph[r];
    const int4 w = *reinterpret_cast<const int4*>(arg + (int64_t)b * 4 * F4 + c);
    const float4 g = *reinterpret_cast<const float4*>(dout + (int64_t)b * ldo + c);
    const int r32 = (int)r;
    o.x += w.x == r32 ? g.x : 0.f; o.y += w.y == r32 ? g.y : 0.f; o.z += w.z == r32 ? g.z : 0.f; o.w += w.w == r32 ? g.w : 0.f;
  }
  *reinterpret_cast<float4*>(dx + r * ldx + c) = o;
}

// ---------------------------------------------------------------- padded <-> packed rows, ghost masking
__global__ void pack_rows_kernel(const float* __restrict__ src, int nmax, int F, const int* __restrict__ row_graph,
                                 const int* __restrict__ row_slot, int64_t n_real, float* __restrict__ dst, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_real * ld) return;
  const int64_t r = i / ld;
  const int f = (int)(i % ld);
  dst[i] = f < F ? src[((int64_t)row_graph[r] * nmax + row_slot[r]) * F + f] : 0.f;
}
// dst[b,n,:] = n < size_b ? src[graph_ptr[b]+n] : (ghost rows ? src[n_real+n] : fill)
__global__ void unpack_rows_kernel(SlotArgs s, const float* __restrict__ src, int64_t ld, int F, float fill,
                                   float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)s.B * s.nmax * F) return;
  const int f = (int)(i % F);
  const int64_t bn = i / F;
  const int n = (int)(bn % s.nmax), b = (int)(bn / s.nmax);
  const int g0 = s.graph_ptr[b], sz = s.graph_ptr[b + 1] - g0;
  float val = fill;
  if (n < sz) val = src[(int64_t)(g0 + n) * ld + f];
  else if (s.n_ghost) val = src[(s.n_real + n) * ld + f];
  dst[i] = val;
}
// gradient of unpack wrt the ghost rows: dsrc[n_real+n, f] = sum_{b: size_b <= n} ddst[b,n,f]
__global__ void unpack_rows_bwd_ghost(SlotArgs s, const float* __restrict__ ddst, int F, float* __restrict__ dsrc, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)s.nmax * F) return;
  const int f = (int)(i % F), n = (int)(i / F);
  float acc = 0.f;
  for (int b = 0; b < s.B; ++b)
    if (s.graph_ptr[b + 1] - s.graph_ptr[b] <= n) acc += ddst[((int64_t)b * s.nmax + n) * F + f];
  dsrc[(s.n_real + n) * ld + f] = acc;
}

}  // namespace

extern "C" {

int tsgnn_bn_slots_fwd_f32(const int* graph_ptr, const int* slot_count, const int* row_slot, int B, int nmax,
                           int64_t n_real, int n_ghost, const float* v, int64_t ldv, int F, int relu, int bn,
                           float* mean, float* rstd, float* y, int64_t ldy, tsgnn_stream_t stream) {
  if (!graph_ptr || !slot_count || !row_slot || !v || !y || B <= 0 || nmax <= 0 || n_real < 0 || F <= 0 ||
      ldv < F || ldy < F || (n_ghost != 0 && n_ghost != nmax) || (bn && (!mean || !rstd)))
    return TSGNN_EINVAL;
  SlotArgs s{graph_ptr, slot_count, B, nmax, n_real, n_ghost};
  const int64_t rows = n_real + n_ghost;
  if (bn) bn_slot_stats<<<nmax, 256, 0, stream>>>(s, v, ldv, F, relu, mean, rstd);
  if (rows > 0)
    bn_slot_apply<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(v, ldv, row_slot, n_real, rows, F, relu,
                                                                     bn ? mean : nullptr, bn ? rstd : nullptr, y, ldy);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_bn_slots_bwd_f32(const int* graph_ptr, const int* slot_count, const int* row_slot, int B, int nmax,
                           int64_t n_real, int n_ghost, const float* v, int64_t ldv, const float* dy, int64_t lddy, int F,
                           int relu, int bn, const float* mean, const float* rstd, float* m1, float* m2, float* dv,
                           int64_t lddv, tsgnn_stream_t stream) {
  if (!graph_ptr || !slot_count || !row_slot || !v || !dy || !dv || B <= 0 || nmax <= 0 || n_real < 0 || F <= 0 ||
      ldv < F || lddy < F || lddv < F || (n_ghost != 0 && n_ghost != nmax) || (bn && (!mean || !rstd || !m1 || !m2)))
    return TSGNN_EINVAL;
  SlotArgs s{graph_ptr, slot_count, B, nmax, n_real, n_ghost};
  const int64_t rows = n_real + n_ghost;
  if (bn) bn_slot_bwd_stats<<<nmax, 256, 0, stream>>>(s, v, ldv, dy, lddy, F, relu, mean, rstd, m1, m2);
  if (rows > 0)
    bn_slot_bwd_apply<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(
        s, v, ldv, dy, lddy, row_slot, rows, F, relu, bn ? mean : nullptr, rstd, m1, m2, dv, lddv);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_row_ln_fwd_f32(const float* v, int64_t ldv, int64_t rows, int F, int relu, float* mean, float* rstd, float* y, int64_t ldy,
                         tsgnn_stream_t stream) {
  if (!v || !mean || !rstd || !y || rows < 0 || F <= 0 || ldv < F || ldy < F) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  row_ln_fwd<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(v, ldv, rows, F, relu, mean, rstd, y, ldy);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_row_ln_bwd_f32(const float* v, int64_t ldv, const float* dy, int64_t lddy, int64_t rows, int F, int relu, const float* mean,
                         const float* rstd, float* dv, int64_t lddv, tsgnn_stream_t stream) {
  if (!v || !dy || !mean || !rstd || !dv || rows < 0 || F <= 0 || ldv < F || lddy < F || lddv < F) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  row_ln_bwd<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(v, ldv, dy, lddy, rows, F, relu, mean, rstd, dv, lddv);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_row_post_bwd_f32(const int* row_graph, int B, int64_t n_real, int64_t rows, const float* v, int64_t ldv, const float* dxs,
                           int64_t lddxs, const float* dout, int64_t ldo, const int* arg, int F, int relu, int ln, const float* mean,
                           const float* rstd, const float* rinv, float* du, int64_t lddu, tsgnn_stream_t stream) {
  if (!v || !rinv || !du || B <= 0 || n_real < 0 || rows < n_real || F <= 0 || ldv < F || lddu < F || (dxs && lddxs < F) ||
      (n_real > 0 && !row_graph) || ((arg == nullptr) != (dout == nullptr)) || (arg && ldo < F) || (ln && (!mean || !rstd)))
    return TSGNN_EINVAL;
  if (F > 256) return TSGNN_EUNSUPPORTED;
  if (rows == 0) return TSGNN_OK;
  TSGNN_KNAME("row_post_bwd");
  row_post_bwd<<<(unsigned)ceil_div64(rows, 4), 256, 0, stream>>>(row_graph, B, n_real, rows, v, ldv, dxs, lddxs, dout, ldo, arg, F, relu, ln,
                                                                 mean, rstd, rinv, du, lddu);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_readout_max_ws_words(int B, int nmax, int F) {
  if (B <= 0 || nmax <= 0 || F <= 0) return 0;
  if (nmax <= 64) return 1;
  const int64_t w = (int64_t)B * F + (B + 1) / 2;
  return w < (int64_t)1 << 30 ? (int)w : -1;
}

int tsgnn_readout_max_fwd_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                              const float* x, int64_t ldx, int F, int relu, unsigned long long* packed_ws, float* out,
                              int64_t ldo, int* arg, tsgnn_stream_t stream) {
  if (!graph_ptr || !x || !packed_ws || !out || !arg || B <= 0 || nmax <= 0 || F <= 0 || ldx < F || ldo < F ||
      (n_ghost != 0 && n_ghost != nmax))
    return TSGNN_EINVAL;
  SlotArgs s{graph_ptr, slot_count, B, nmax, n_real, n_ghost};
  if (nmax <= 64) {                                        // one chunk per graph: a single launch
    readout_max_direct<<<(unsigned)B, 256, 0, stream>>>(s, x, ldx, F, relu, out, ldo, arg);
    TSGNN_CHECK_LAUNCH();
    return TSGNN_OK;
  }
  const int FP = (F + 63) & ~63;
  const int nch = (nmax + 63) / 64;
  dim3 grid((unsigned)nch, (unsigned)B);
  unsigned* count = reinterpret_cast<unsigned*>(packed_ws + (size_t)B * F);
  readout_max_chunks<<<grid, 256, sizeof(unsigned long long) * 4 * FP, stream>>>(s, x, ldx, F, relu, packed_ws, count, out, ldo, arg);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_readout_max_bwd_f32(const float* dout, int64_t ldo, const int* arg, int B, int F, const float* x, int64_t ldx_in,
                              int relu, int64_t n_real, float* dx, int64_t ldx, tsgnn_stream_t stream) {
  if (!dout || !arg || !dx || B <= 0 || F <= 0 || ldo < F || ldx < F || (relu && !x)) return TSGNN_EINVAL;
  readout_max_bwd<<<(unsigned)ceil_div64((int64_t)B * F, 256), 256, 0, stream>>>(dout, ldo, arg, B, F, x, ldx_in, relu,
                                                                                n_real, dx, ldx);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* backward of the max readout as a dense pass: writes EVERY element of dx[rows_total, F] (no fill, no atomics).  rows: the rows
 * that can hold a maximum (all of them for batches without ghost rows; the real rows when the caller discards the ghost rows'
 * gradient); add (nullable): a second gradient of the same tensor, summed in the same pass. */
int tsgnn_readout_max_bwd_rows_f32(const float* dout, int64_t ldo, const int* arg, const int* row_graph, int F, int64_t rows,
                                   int64_t rows_total, const float* add, int64_t ldadd, float* dx, int64_t ldx, tsgnn_stream_t stream) {
  if (!dout || !arg || !row_graph || !dx || F <= 0 || rows < 0 || rows_total < rows || ldo < F || ldx < F || (add && ldadd < F))
    return TSGNN_EINVAL;
  if ((F % 4) || (ldo % 4) || (ldx % 4) || (add && (ldadd % 4)) ||
      ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(arg) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(add)) & 15))
    return TSGNN_EUNSUPPORTED;
  if (rows_total == 0) return TSGNN_OK;
  readout_max_bwd_rows<<<(unsigned)ceil_div64(rows_total * (F / 4), 256), 256, 0, stream>>>(dout, ldo, arg, row_graph, F / 4, rows, rows_total,
                                                                                             add, ldadd, dx, ldx);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_pack_rows_f32(const float* src, int nmax, int F, const int* row_graph, const int* row_slot, int64_t n_real,
                        float* dst, int64_t ld, tsgnn_stream_t stream) {
  if (!src || !row_graph || !row_slot || !dst || nmax <= 0 || F <= 0 || ld < F || n_real < 0) return TSGNN_EINVAL;
  if (n_real == 0) return TSGNN_OK;
  pack_rows_kernel<<<(unsigned)ceil_div64(n_real * ld, 256), 256, 0, stream>>>(src, nmax, F, row_graph, row_slot, n_real, dst, ld);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_unpack_rows_f32(const int* graph_ptr, int B, int nmax, int64_t n_real, int n_ghost, const float* src, int64_t ld,
                          int F, float fill, float* dst, tsgnn_stream_t stream) {
  if (!graph_ptr || !src || !dst || B <= 0 || nmax <= 0 || F <= 0 || ld < F) return TSGNN_EINVAL;
  SlotArgs s{graph_ptr, nullptr, B, nmax, n_real, n_ghost};
  unpack_rows_kernel<<<(unsigned)ceil_div64((int64_t)B * nmax * F, 256), 256, 0, stream>>>(s, src, ld, F, fill, dst);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_unpack_rows_bwd_ghost_f32(const int* graph_ptr, int B, int nmax, int64_t n_real, const float* ddst, int F,
                                    float* dsrc, int64_t ld, tsgnn_stream_t stream) {
  if (!graph_ptr || !ddst || !dsrc || B <= 0 || nmax <= 0 || F <= 0 || ld < F) return TSGNN_EINVAL;
  SlotArgs s{graph_ptr, nullptr, B, nmax, n_real, nmax};
  unpack_rows_bwd_ghost<<<(unsigned)ceil_div64((int64_t)nmax * F, 256), 256, 0, stream>>>(s, ddst, F, dsrc, ld);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
