// Graph-level prediction head of the encoders (encoders.py:207-217 / :396-406): two chained nn.Linear on the
// concatenated readout [B, P]  ->  vec [B, E]  ->  y [B, C],  forward and backward in 1 + 1 launches instead of the
// ~8 library launches (2 addmm, 4 mm, 2 bias reductions) a B = 32 batch spends most of its time dispatching.
// Shapes are tiny (B <= a few hundred rows): one workgroup per graph row, weights streamed from L2 with 16-byte loads.
#include "common.h"
#include "../../include/tsgnn.h"
#include "readout_body.h"
#include "ingest_rider.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// block b: vec[b,:] = W1 out[b,:] + b1 ; y[b,:] = W2 vec[b,:] + b2        (W1 [E,P], W2 [C,E] row-major = nn.Linear.weight)
constexpr int HW = 16;   // waves per row block (1024 threads): one batch of 8 weight rows per wave covers E = 128

// shared tail: xs[P] (the readout row of graph b) is in LDS; vec[b,:] = W1 xs + b1 ; y[b,:] = W2 vec[b,:] + b2
__device__ __forceinline__ void head2_fwd_tail(const float* xs, float* vs, int b, const float* __restrict__ w1,
                                               const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                               int P, int E, int C, float* __restrict__ vec, float* __restrict__ y) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int P4 = P >> 2;
  // each wave owns rows j = wid, wid+16, ...; eight rows are in flight per iteration (independent 16-byte loads),
  // their dot products are reduced together
  for (int j0 = wid; j0 < E; j0 += 8 * HW) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int k4 = lane; k4 < P4; k4 += 64) {
      const float4 x = *reinterpret_cast<const float4*>(xs + 4 * k4);
      float4 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + HW * u;
        w[u] = ld4(w1 + (int64_t)(j < E ? j : 0) * P + 4 * k4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += (w[u].x * x.x + w[u].y * x.y) + (w[u].z * x.z + w[u].w * x.w);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + HW * u;
      const float r = wave_sum(acc[u]);
      if (lane == 0 && j < E) { const float v = r + (b1 ? b1[j] : 0.f); vs[j] = v; vec[(int64_t)b * E + j] = v; }
    }
  }
  __syncthreads();
  for (int c = wid; c < C; c += HW) {
    float acc = 0.f;
    for (int j = lane; j < E; j += 64) acc = fmaf(w2[(int64_t)c * E + j], vs[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) y[(int64_t)b * C + c] = acc + (b2 ? b2[c] : 0.f);
  }
}

// the max readout of a LAST uniform level (N <= 64 rows per graph: rows b * N + n of z) that fills columns [c0, c0 + F) of the
// head's input inside the head's own launches (DiffPool's last pooled level, encoders.py:383,388-391)
struct RoTail {
  const float* z; int64_t ldz; int N, c0, F;
  int* arg;                         // [B, F] winning rows (forward: written; backward: read)
  float* dz; int64_t lddz;          // backward: gradient of z's rows, every element written
};

__global__ __launch_bounds__(64 * HW) void head2_fwd_kernel(float* __restrict__ out, int64_t ldo, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, int P, int E, int C,
                                                        float* __restrict__ vec, float* __restrict__ y, RoTail rt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                 // [P]
  float* vs = smem + ((P + 3) & ~3);  // [E]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c0 = rt.z ? rt.c0 : P;
  for (int k = tid; k < P; k += 64 * HW) {
    if (k < c0) { xs[k] = out[(int64_t)b * ldo + k]; continue; }
    // column f of the tail: the same packed (value, row) order as readout_max_direct (bn_readout.hip), rows in batches of 16
    const int f = k - c0;
    unsigned long long best = 0ull;
    for (int n0 = 0; n0 < rt.N; n0 += 16) {
      float val[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) val[u] = rt.z[((int64_t)b * rt.N + min(n0 + u, rt.N - 1)) * rt.ldz + f];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (n0 + u < rt.N) {
          const unsigned long long q = ((unsigned long long)f32_ordered(val[u]) << 32) |
                                       (unsigned long long)(0xFFFFFFFFu - (unsigned)(b * rt.N + n0 + u));
          best = q > best ? q : best;
        }
      }
    }
    const float v = best ? ordered_f32((unsigned)(best >> 32)) : 0.f;
    xs[k] = v;
    out[(int64_t)b * ldo + k] = v;
    rt.arg[(int64_t)b * rt.F + f] = best ? (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull)) : -1;
  }
  __syncthreads();
  head2_fwd_tail(xs, vs, b, w1, b1, w2, b2, P, E, C, vec, y);
}

// The GraphSage stack's tail in ONE launch: block b finishes graph b's concatenated max readout and runs the head on it.
//   layers 0 .. L-2 : decode the packed (value, ~row) maxima the per-layer partial kernels accumulated;
//   last layer      : scan graph b's rows of v_last directly (32 row lanes x 32 float4 lanes; first ghost row = all ghost
//                     rows of the un-normalised last layer) - no partial launch, no atomics for that layer;
//   then vec = W1 out + b1, y = W2 vec + b2.   out[b, :] and arg (winning rows, packed-layout order) are kept for backward.
struct ReadoutHeadArgs {
  const unsigned long long* packed; int B, L, Fh, Fl;
  const float* v_last; int64_t ldv;
  const int* graph_ptr; int64_t n_real; int nslots; int n_ghost;
  float* out; int64_t ldo; int* arg;
};
__global__ __launch_bounds__(64 * HW) void readout_head_fwd_kernel(ReadoutHeadArgs a, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, int P, int E, int C,
                                                               float* __restrict__ vec, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                                   // [P]
  float* vs = smem + ((P + 3) & ~3);                                  // [E]
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem + ((P + 3) & ~3) + ((E + 3) & ~3));   // [32][Fl]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int PH = (a.L - 1) * a.Fh;
  for (int k = tid; k < PH; k += 64 * HW) {
    const int l = k / a.Fh, f = k % a.Fh;
    const int64_t i = (int64_t)l * a.B * a.Fh + (int64_t)b * a.Fh + f;
    const unsigned long long p = a.packed[i];
    const float val = p ? ordered_f32((unsigned)(p >> 32)) : 0.f;
    xs[k] = val;
    a.out[(int64_t)b * a.ldo + k] = val;
    a.arg[i] = p ? (int)(0xFFFFFFFFu - (unsigned)(p & 0xFFFFFFFFull)) : -1;
  }
  {
    const int c4 = tid & 31, rl = tid >> 5, F4 = a.Fl >> 2;
    const int g0 = a.graph_ptr[b], sz = a.graph_ptr[b + 1] - g0;
    unsigned long long m0 = 0ull, m1 = 0ull, m2 = 0ull, m3 = 0ull;
    if (c4 < F4) {
      const int nrows = sz + ((a.n_ghost && sz < a.nslots) ? 1 : 0);   // + the first ghost row (the others equal it)
      // four rows in flight per lane (independent 16-byte loads), maxima folded afterwards
      for (int n0 = rl; n0 < nrows; n0 += 8 * HW) {
        float4 t[4];
        int64_t r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = n0 + 2 * HW * u;
          r[u] = n < sz ? (int64_t)g0 + n : a.n_real + n;
          t[u] = n < nrows ? ld4(a.v_last + r[u] * a.ldv + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (n0 + 2 * HW * u < nrows) {
            const unsigned long long p0 = pack_max(t[u].x, (unsigned)r[u]), p1 = pack_max(t[u].y, (unsigned)r[u]),
                                     p2 = pack_max(t[u].z, (unsigned)r[u]), p3 = pack_max(t[u].w, (unsigned)r[u]);
            m0 = p0 > m0 ? p0 : m0; m1 = p1 > m1 ? p1 : m1; m2 = p2 > m2 ? p2 : m2; m3 = p3 > m3 ? p3 : m3;
          }
        }
      }
      best[rl * a.Fl + 4 * c4 + 0] = m0; best[rl * a.Fl + 4 * c4 + 1] = m1; best[rl * a.Fl + 4 * c4 + 2] = m2; best[rl * a.Fl + 4 * c4 + 3] = m3;
    }
    __syncthreads();
    for (int f = tid; f < a.Fl; f += 64 * HW) {
      unsigned long long m = best[f];
      for (int w = 1; w < 2 * HW; ++w) { const unsigned long long o = best[w * a.Fl + f]; m = o > m ? o : m; }
      const float val = m ? ordered_f32((unsigned)(m >> 32)) : 0.f;
      xs[PH + f] = val;
      a.out[(int64_t)b * a.ldo + PH + f] = val;
      a.arg[(int64_t)(a.L - 1) * a.B * a.Fh + (int64_t)b * a.Fl + f] = m ? (int)(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull)) : -1;
    }
  }
  __syncthreads();
  head2_fwd_tail(xs, vs, b, w1, b1, w2, b2, P, E, C, vec, y);
}

// The tail when the last layer's readout is already in `packed` (tsgnn_sage_layer_fwd_ro_f32 folded it into the product's
// epilogue): block b decodes its P packed maxima and runs the head.  Everything the block reads from memory is requested
// before anything is waited for — the packed words (the only operands that depend on the previous launch) and, behind them,
// the wave's W1 rows and W2 — so the kernel is ONE memory round trip, an LDS exchange and two short reductions.
template <int NJ>     // W1 rows per wave held in registers (E <= HW * NJ)
__global__ __launch_bounds__(64 * HW) void packed_head_fwd_kernel(const unsigned long long* __restrict__ packed, int B, int L, int Fh,
                                                              int Fl, float* __restrict__ out, int64_t ldo, int* __restrict__ arg,
                                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2, int P, int E,
                                                              int C, float* __restrict__ vec, float* __restrict__ y, ExpandRider rider,
                                                              unsigned long long* __restrict__ clear, int64_t clear_n, int clear_packed) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= B) {                        // passengers: the next mini-batch's expansion on the CUs this launch leaves idle
    expand_rider_body(rider, blockIdx.x - (unsigned)B, 64 * HW);
    return;
  }
  // housekeeping for the NEXT step (tsgnn_packed_head_fwd_z_f32): the integer sums of the fused slot batch-norms were consumed by
  // the launches before this one; zeroing them here (and this graph's packed maxima after they are decoded, below) takes the
  // clearing launch out of the step
  for (int64_t i2 = (int64_t)blockIdx.x * (64 * HW) + threadIdx.x; i2 < clear_n; i2 += (int64_t)B * (64 * HW)) clear[i2] = 0ull;
  float* xs = smem;                                  // [P]
  float* vs = smem + ((P + 3) & ~3);                 // [E]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int PH = (L - 1) * Fh, P4 = P >> 2;
  // (1) the packed maxima of this graph: thread k < P (P <= 2048: two per thread)
  unsigned long long pk[2];
  int64_t pidx[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = tid + u * 64 * HW;
    int64_t idx = 0;
    if (k < P) {
      if (k < PH) { const int l = k / Fh, f = k - l * Fh; idx = (int64_t)l * B * Fh + (int64_t)b * Fh + f; }
      else idx = (int64_t)(L - 1) * B * Fh + (int64_t)b * Fl + (k - PH);
    }
    pidx[u] = idx;
    pk[u] = packed[idx];                               // unconditional (idx = 0 beyond P), masked below: a load inside a
    if (k >= P) pk[u] = 0ull;                          // predicate is waited for before the next request is issued
  }
  // (2) this wave's rows of W1 (j = wid + HW * u), float4 column `lane` and `lane + 64` of each; W2 for the waves that own a class
  float4 w[NJ][2];
#pragma unroll
  for (int u = 0; u < NJ; ++u) {
    const int j = wid + HW * u;
    const float* row = w1 + (int64_t)(j < E ? j : 0) * P;
    // (every request unconditional from a clamped address, masked afterwards: with `cond ? ld4(..) : 0` the ISA had sixteen
    // load -> s_waitcnt vmcnt(0) pairs here — sixteen dependent round trips in a kernel meant to be one)
    w[u][0] = ld4(row + 4 * (lane < P4 ? lane : 0));
    w[u][1] = ld4(row + 4 * (lane + 64 < P4 ? lane + 64 : 0));
  }
#pragma unroll
  for (int u = 0; u < NJ; ++u) {
    if (lane >= P4) w[u][0] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane + 64 >= P4) w[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float bj[NJ];
#pragma unroll
  for (int u = 0; u < NJ; ++u) { const int j = wid + HW * u; bj[u] = (b1 && j < E) ? b1[j] : 0.f; }
  float w2v[4];                                      // class `wid`: columns lane, lane + 64, ... of W2 (E <= 256 in registers)
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int j = lane + 64 * q; w2v[q] = (wid < C && j < E) ? w2[(int64_t)wid * E + j] : 0.f; }
  const float b2v = (b2 && wid < C) ? b2[wid] : 0.f;
  // decode -> LDS, out, arg
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = tid + u * 64 * HW;
    if (k < P) {
      const unsigned long long p = pk[u];
      const float val = p ? ordered_f32((unsigned)(p >> 32)) : 0.f;
      xs[k] = val;
      out[(int64_t)b * ldo + k] = val;
      arg[pidx[u]] = p ? (int)(0xFFFFFFFFu - (unsigned)(p & 0xFFFFFFFFull)) : -1;
      if (clear_packed) const_cast<unsigned long long*>(packed)[pidx[u]] = 0ull;
    }
  }
  __syncthreads();
  {
    const float4 x0 = lane < P4 ? *reinterpret_cast<const float4*>(xs + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 x1 = lane + 64 < P4 ? *reinterpret_cast<const float4*>(xs + 4 * (lane + 64)) : make_float4(0.f, 0.f, 0.f, 0.f);
    float acc[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u)
      acc[u] = ((w[u][0].x * x0.x + w[u][0].y * x0.y) + (w[u][0].z * x0.z + w[u][0].w * x0.w)) +
               ((w[u][1].x * x1.x + w[u][1].y * x1.y) + (w[u][1].z * x1.z + w[u][1].w * x1.w));
    // columns beyond 128 float4 (P > 512) and rows beyond HW * NJ are streamed (not the shapes this kernel is picked for)
    for (int k4 = lane + 128; k4 < P4; k4 += 64) {
      const float4 x = *reinterpret_cast<const float4*>(xs + 4 * k4);
#pragma unroll
      for (int u = 0; u < NJ; ++u) {
        const int j = wid + HW * u;
        const float4 t = ld4(w1 + (int64_t)(j < E ? j : 0) * P + 4 * k4);
        acc[u] += (t.x * x.x + t.y * x.y) + (t.z * x.z + t.w * x.w);
      }
    }
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
      const int j = wid + HW * u;
      const float r = wave_sum(acc[u]);
      if (lane == 0 && j < E) { const float v = r + bj[u]; vs[j] = v; vec[(int64_t)b * E + j] = v; }
    }
  }
  __syncthreads();
  for (int c = wid; c < C; c += HW) {
    float acc = 0.f;
    if (c == wid && E <= 256) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int j = lane + 64 * q; if (j < E) acc = fmaf(w2v[q], vs[j], acc); }
    } else {
      for (int j = lane; j < E; j += 64) acc = fmaf(w2[(int64_t)c * E + j], vs[j], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) y[(int64_t)b * C + c] = acc + ((c == wid) ? b2v : (b2 ? b2[c] : 0.f));
  }
}

// Backward in ONE launch.  Blocks [0, B): block b computes dvt[b,:] = dvec[b,:] + W2^T dy[b,:] and dout[b,:] = W1^T dvt[b,:].
// Blocks [B, B + ceil(E/4)]: the weight gradients; they rebuild the four dvt columns they need from dy, dvec and W2
// (C fused multiply-adds per value) instead of waiting for the row blocks, so nothing orders the two groups.
__device__ __forceinline__ void head2_bwd_rows(float* smem, int b, const float* __restrict__ dy, const float* __restrict__ dvec,
                                               const float* __restrict__ w1, const float* __restrict__ w2, int P, int E, int C,
                                               float* __restrict__ dout, int64_t ldo) {
  float* ds = smem;                          // [E] dvt row
  float* part = smem + ((E + 3) & ~3);       // [HW][P] partial dout
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  for (int j = tid; j < E; j += 64 * HW) {
    float acc = dvec ? dvec[(int64_t)b * E + j] : 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(dy[(int64_t)b * C + c], w2[(int64_t)c * E + j], acc);
    ds[j] = acc;
  }
  __syncthreads();
  // each wave takes every 4th row j of W1 and accumulates its contribution to all P columns
  const int P4 = P >> 2;
  for (int k4 = lane; k4 < P4 + ((P & 3) ? 1 : 0); k4 += 64) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k4 < P4) {
      for (int j = wid; j < E; j += HW) {
        const float4 w = ld4(w1 + (int64_t)j * P + 4 * k4);
        const float d = ds[j];
        acc.x = fmaf(d, w.x, acc.x); acc.y = fmaf(d, w.y, acc.y); acc.z = fmaf(d, w.z, acc.z); acc.w = fmaf(d, w.w, acc.w);
      }
      *reinterpret_cast<float4*>(part + wid * ((P + 3) & ~3) + 4 * k4) = acc;
    } else {
      for (int k = 4 * P4; k < P; ++k) {
        float a = 0.f;
        for (int j = wid; j < E; j += HW) a = fmaf(ds[j], w1[(int64_t)j * P + k], a);
        part[wid * ((P + 3) & ~3) + k] = a;
      }
    }
  }
  __syncthreads();
  const int PP = (P + 3) & ~3;
  for (int k = tid; k < P; k += 64 * HW) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < HW; ++w) a += part[w * PP + k];
    dout[(int64_t)b * ldo + k] = a;
  }
}

// weights: block jb owns rows j = 4*jb .. 4*jb+3 of dW1 (+ db1); the last block produces dW2, db2
__device__ __forceinline__ void head2_bwd_weights(float* smem /* [B][4] dvt slice */, int jb, const float* __restrict__ out, int64_t ldo,
                                                  const float* __restrict__ vec, const float* __restrict__ dy,
                                                  const float* __restrict__ dvec, const float* __restrict__ w2, int B, int P, int E,
                                                  int C, float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                  float* __restrict__ db2, float* __restrict__ normparts) {
  const int tid = threadIdx.x, NTH = 64 * HW;
  const int nj = (E + 3) / 4;
  float sq = 0.f;                                        // |gradient written by this thread|^2
  if (jb < nj) {
    const int j0 = 4 * jb;
    for (int i = tid; i < B * 4; i += NTH) {
      const int bb = i >> 2, jj = j0 + (i & 3);
      float a = 0.f;
      if (jj < E) {
        a = dvec ? dvec[(int64_t)bb * E + jj] : 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(dy[(int64_t)bb * C + c], w2[(int64_t)c * E + jj], a);   // same order as the row blocks
      }
      smem[i] = a;
    }
    __syncthreads();
    for (int k = tid; k < P; k += NTH) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int bb = 0;
      for (; bb + 8 <= B; bb += 8) {                     // eight rows in flight (independent loads), same summation order
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = out[(int64_t)(bb + u) * ldo + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 d = *reinterpret_cast<const float4*>(smem + 4 * (bb + u));
          a0 = fmaf(d.x, x[u], a0); a1 = fmaf(d.y, x[u], a1); a2 = fmaf(d.z, x[u], a2); a3 = fmaf(d.w, x[u], a3);
        }
      }
      for (; bb < B; ++bb) {
        const float x = out[(int64_t)bb * ldo + k];
        a0 = fmaf(smem[4 * bb + 0], x, a0); a1 = fmaf(smem[4 * bb + 1], x, a1);
        a2 = fmaf(smem[4 * bb + 2], x, a2); a3 = fmaf(smem[4 * bb + 3], x, a3);
      }
      if (j0 + 0 < E) { dw1[(int64_t)(j0 + 0) * P + k] = a0; sq = fmaf(a0, a0, sq); }
      if (j0 + 1 < E) { dw1[(int64_t)(j0 + 1) * P + k] = a1; sq = fmaf(a1, a1, sq); }
      if (j0 + 2 < E) { dw1[(int64_t)(j0 + 2) * P + k] = a2; sq = fmaf(a2, a2, sq); }
      if (j0 + 3 < E) { dw1[(int64_t)(j0 + 3) * P + k] = a3; sq = fmaf(a3, a3, sq); }
    }
    if (db1 && tid < 4 && j0 + tid < E) {
      float a = 0.f;
      for (int bb = 0; bb < B; ++bb) a += smem[4 * bb + tid];
      db1[j0 + tid] = a;
      sq = fmaf(a, a, sq);
    }
  } else {
    for (int i = tid; i < C * E; i += NTH) {
      const int c = i / E, j = i % E;
      float a = 0.f;
      for (int bb = 0; bb < B; ++bb) a = fmaf(dy[(int64_t)bb * C + c], vec[(int64_t)bb * E + j], a);
      dw2[i] = a;
      sq = fmaf(a, a, sq);
    }
    if (db2) for (int c = tid; c < C; c += NTH) {
      float a = 0.f;
      for (int bb = 0; bb < B; ++bb) a += dy[(int64_t)bb * C + c];
      db2[c] = a;
      sq = fmaf(a, a, sq);
    }
  }
  if (normparts) {                                       // block total in a fixed order: waves through LDS
    sq = wave_sum(sq);
    __syncthreads();
    if ((tid & 63) == 0) smem[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < HW; ++w) t += smem[w];
      normparts[jb] = t;
    }
  }
}

// ce_label != NULL: dy is not an input — it is the gradient of mean softmax cross-entropy of the logits ce_y[B, C] w.r.t. them
// (F.cross_entropy, encoders.py:221-224), rebuilt by every block in LDS (B * C values) instead of being produced by a launch
// of its own between the forward and this kernel; block 0 also writes the loss value.
__global__ __launch_bounds__(64 * HW) void head2_bwd_kernel(const float* __restrict__ out, int64_t ldo, const float* __restrict__ vec,
                                                        const float* __restrict__ dy, const float* __restrict__ dvec,
                                                        const float* __restrict__ w1, const float* __restrict__ w2, int B, int P, int E,
                                                        int C, float* __restrict__ dout, int64_t lddo, float* __restrict__ dw1,
                                                        float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                        float* __restrict__ normparts, const float* __restrict__ ce_y,
                                                        const int64_t* __restrict__ ce_label, float* __restrict__ ce_loss) {
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  float* smem = smem_all;
  if (ce_label != nullptr) {
    float* dyl = smem_all;                               // [B * C]
    float* lb = smem_all + ((B * C + 3) & ~3);           // [B] per-graph loss terms
    smem = lb + ((B + 3) & ~3);
    const float invB = 1.f / (float)B;
    // weight blocks (and block 0, which also writes the loss) need every row's gradient, a row block only its own
    const bool all_rows = (int)blockIdx.x >= B || blockIdx.x == 0;
    for (int b = all_rows ? (int)threadIdx.x : (int)blockIdx.x + (int)threadIdx.x * B; b < B; b += 64 * HW) {
      const float* row = ce_y + (int64_t)b * C;
      float m = -INFINITY;
      for (int c = 0; c < C; ++c) m = fmaxf(m, row[c]);
      float d = 0.f;
      for (int c = 0; c < C; ++c) d += expf(row[c] - m);
      const int yb = (int)ce_label[b];
      const float logz = m + logf(d);
      lb[b] = logz - row[yb];
      for (int c = 0; c < C; ++c) dyl[b * C + c] = (expf(row[c] - logz) - (c == yb ? 1.f : 0.f)) * invB;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      float t = 0.f;
      for (int b = 0; b < B; ++b) t += lb[b];            // graph order: reproducible
      ce_loss[0] = t * invB;
    }
    dy = dyl;
  }
  if ((int)blockIdx.x < B) head2_bwd_rows(smem, blockIdx.x, dy, dvec, w1, w2, P, E, C, dout, lddo);
  else head2_bwd_weights(smem, (int)blockIdx.x - B, out, ldo, vec, dy, dvec, w2, B, P, E, C, dw1, db1, dw2, db2, normparts);
}

// ---- second generation of the backward launch: same roles, same arithmetic per output, but every block REQUESTS all of its
// global operands (logits / labels, W2, its W1 rows or its rows of `out`) before it waits for anything, so a block is one
// memory round trip followed by LDS phases instead of four or five dependent trips (measured r1: 11.6 us in the step).
//   row block b   : threads = (row group jg, float4 column k4); group jg owns rows j = jg + G*u of W1; partial sums over the
//                   groups meet in LDS and are added in group order (fixed order: reproducible).
//   weight block  : threads = (graph group bg, float4 column k4) over `out`; the four dvt columns of the block sit in LDS.
constexpr int RB_ROWS = 16;      // W1 rows per thread and batch (row blocks)
constexpr int WB_ROWS = 4;       // rows of `out` per thread and batch (weight blocks)

struct CeArgs { const float* y; const int64_t* label; float* loss; };

// softmax-CE gradient rows into dyl[B*C] (+ per-graph loss terms lb[B]) — rows: all of them, or only row `only` (row blocks > 0)
__device__ __forceinline__ void ce_rows(const CeArgs& ce, int B, int C, float* dyl, float* lb, int only) {
  const float invB = 1.f / (float)B;
  for (int b = only >= 0 ? only + (int)threadIdx.x * B : (int)threadIdx.x; b < B; b += 64 * HW) {
    const float* row = ce.y + (int64_t)b * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, row[c]);
    float d = 0.f;
    for (int c = 0; c < C; ++c) d += expf(row[c] - m);
    const int yb = (int)ce.label[b];
    const float logz = m + logf(d);
    lb[b] = logz - row[yb];
    for (int c = 0; c < C; ++c) dyl[b * C + c] = (expf(row[c] - logz) - (c == yb ? 1.f : 0.f)) * invB;
  }
}

// ---- third role of the backward launch (tsgnn_head2_bwd_du_f32): dU of the stack's LAST GraphConv layer.  That layer has no
// batch-norm, so its dU is a row-wise function of the readout gradient (readout_l2_bwd_rows in sage_fused.hip):
//   du[r] = rinv_r (g_r - v_r <v_r, g_r>),   g_r[f] = dout[b, off + f] if r is the max-readout winner of (graph b, column f), else 0.
// It used to be a launch of its own between this one and the layer's weight-gradient / input-gradient launch.  Here block
// (b, c) owns rows [128c, 128c + 128) of graph b and does not wait for row block b: it rebuilds the 128-wide segment of dout[b, :]
// itself from the same operands in the same order (same bits), with every request — its rows of v, rinv, the winners, its W1
// rows — issued before anything is waited for.  Ghost rows: all ghost rows of this layer are identical, so graph b can only
// have won with its first one (row n_real + size_b); its contribution is written to du[n_real + b] — the rows behind the real
// ones feed the bias gradient only (they aggregate nothing), where only their SUM matters, so B contribution rows stand for
// the ghost rows (the caller passes bias_only_rows = B to the weight-gradient launch).  b == B: the padding rows of a
// capacity-padded batch ([graph_ptr[B], n_real)) are zero-filled.
// DU_CHUNK (64 or 128): rows of a graph per dU block.  Every block rebuilds its graph's dout segment, so fewer, larger blocks do less
// redundant work — but 128 rows per block is four rows per lane group and a slower block; the entry point takes 64 while the launch
// stays near one block per compute unit and 128 above (DD batches whose largest graph has > 448 nodes).
struct DuArgs {
  const int* graph_ptr; int64_t n_real; int n_ghost_rows; int chunks;
  const float* v; int64_t ldv; const float* rinv; const int* arg; int off; int F;
  float* du; int64_t lddu;
  // nullable: the (graph, chunk) of dU block d as graph << 8 | chunk — the host lists the NON-EMPTY chunks of an exact batch
  // (tsgnn_head2_bwd_du_map_f32), so that no workgroup is launched only to return; without it block d is (d / chunks, d % chunks)
  // of the dense (B + 1) x chunks grid, and which blocks come in a second round depends on where the batch's large graphs sit
  // (8.5 us for one DD batch, 11.1 for another)
  const int* map;
  // compact != 0 (B <= 64, no map): block d is the d-th NON-EMPTY (graph, chunk) pair, resolved by every wave itself from the graph
  // pointers (a 64-lane prefix sum of the graphs' chunk counts: no host list — the sizes of a capacity-padded ingest batch live on
  // the device); the blocks behind the last pair (compact = their number) zero-fill the padding rows
  int compact;
};
// (graph, chunk) of dU block d; pad_k >= 0: not a pair but the pad_k-th of pad_n zero-fill blocks
template <int DU_CHUNK>
__device__ __forceinline__ void du_resolve(const DuArgs& a, int d, int B, int& b, int& c, int& pad_k, int& pad_n) {
  pad_k = -1; pad_n = a.chunks;
  if (a.map) { b = a.map[d] >> 8; c = a.map[d] & 255; }
  else if (a.compact) {
    const int lane = threadIdx.x & 63;
    const int g0 = a.graph_ptr[min(lane, B)], g1 = a.graph_ptr[min(lane + 1, B)];
    const int nch = lane < B ? max(1, (g1 - g0 + DU_CHUNK - 1) / DU_CHUNK) : 0;
    int pre = nch;                                               // inclusive prefix over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, o, 64); if (lane >= o) pre += t; }
    const int total = __shfl(pre, 63, 64);
    if (d >= total) { b = B; c = 0; pad_k = d - total; pad_n = (int)gridDim.x - a.compact - total; return; }
    const unsigned long long m = __ballot(pre > d);              // the first lane whose prefix exceeds d owns block d
    b = __ffsll((long long)m) - 1;
    c = d - (__shfl(pre, b, 64) - __shfl(nch, b, 64));
  } else { b = d / a.chunks; c = d - b * a.chunks; }
  if (b >= B) { pad_k = c; b = B; }
}

template <int DU_CHUNK>
__device__ __forceinline__ void head2_du_role(const DuArgs& a, float* smem, const float* dy /* LDS or global [B, C] */, int b, int c,
                                              int pad_k, int pad_n, const float* __restrict__ dvec, const float* __restrict__ w1,
                                              const float* __restrict__ w2, int B, int P, int E, int C, bool dy_ready_needs_sync) {
  const int tid = threadIdx.x, NTH = 64 * HW;
  const int F4 = a.F >> 2, lig = tid & 31, rg = tid >> 5;       // 32 lanes per row (F <= 128), 32 rows per pass
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pad_k >= 0) {                                              // padding rows of a capacity-padded batch: zeros
    const int64_t lo = a.graph_ptr[B];
    for (int64_t r = lo + (int64_t)pad_k * 32 + rg; r < a.n_real; r += (int64_t)max(pad_n, 1) * 32)
      if (lig < F4) *reinterpret_cast<float4*>(a.du + r * a.lddu + 4 * lig) = z4;
    return;
  }
  const int g0 = a.graph_ptr[b], sz = a.graph_ptr[b + 1] - g0;
  const int lo = DU_CHUNK * c, hi = min(sz, lo + DU_CHUNK);
  const bool ghost_job = c == 0;                                 // chunk 0 also writes the graph's ghost contribution row
  if (lo >= hi && !ghost_job) return;
  const int P4 = P >> 2;
  const int G = min(16, NTH / P4);                               // the row block's grouping of the W1 rows: same sums, same order
  float* ds = smem;                                              // [E]
  float* part = smem + ((E + 3) & ~3);                           // [G][F]
  float* dseg = part + G * a.F;                                  // [F]
  // ---- requests
  // (W1 rows in batches of DU_ROWS = 8: sixteen rows in flight plus the row operands below spilled 46 registers of this 1,024-thread
  // kernel into scratch — 17.5 us for the launch instead of 6.2; the FIRST batch is requested up front, the second after its products,
  // accumulated in the row block's order u = 0..15)
  constexpr int DU_ROWS = 8;
  static_assert(RB_ROWS % DU_ROWS == 0, "the row block's batches are walked in pieces");
  const int jg = tid >> 5;                                       // W1 row group (groups >= G idle)
  const bool wact = jg < G && lig < F4;
  // (addresses as a uniform base + a 32-bit lane offset: one register per request instead of a 64-bit pair — this kernel runs at
  // 128 registers per lane, and the pairs were what spilled)
  const unsigned wcol = (unsigned)a.off + 4u * (unsigned)(wact ? lig : 0);
  float4 w[DU_ROWS];
#pragma unroll
  for (int u = 0; u < DU_ROWS; ++u) {
    const int j = jg + G * u;
    w[u] = ld4(w1 + (size_t)((unsigned)((wact && j < E) ? j : 0) * (unsigned)P + wcol));
  }
  float w2r[4];
  float dv0 = 0.f;
  if (tid < E) {
#pragma unroll
    for (int q = 0; q < 4; ++q) w2r[q] = q < C ? w2[(int64_t)q * E + tid] : 0.f;
    if (dvec) dv0 = dvec[(int64_t)b * E + tid];
  }
  if (dy_ready_needs_sync) __syncthreads();
  // ---- dvt row of graph b, then the segment of dout[b, :]   (expressions and order of the row blocks)
  for (int j = tid; j < E; j += NTH) {
    float acc = j == tid ? dv0 : (dvec ? dvec[(int64_t)b * E + j] : 0.f);
    for (int q = 0; q < C; ++q) acc = fmaf(dy[(int64_t)b * C + q], (j == tid && q < 4) ? w2r[q] : w2[(int64_t)q * E + j], acc);
    ds[j] = acc;
  }
  __syncthreads();
  // (E <= G * RB_ROWS, checked by the entry point: the row block's loop over batches of G * RB_ROWS rows has ONE iteration, and
  // no loop here keeps the address arithmetic of the requests out of long live ranges)
  float4 acc = z4;
#pragma unroll
  for (int u0 = 0; u0 < RB_ROWS; u0 += DU_ROWS) {
    if (u0 > 0) {
#pragma unroll
      for (int u = 0; u < DU_ROWS; ++u) {
        const int j = jg + G * (u0 + u);
        w[u] = ld4(w1 + (size_t)((unsigned)((wact && j < E) ? j : 0) * (unsigned)P + wcol));
      }
    }
#pragma unroll
    for (int u = 0; u < DU_ROWS; ++u) {
      const int j = jg + G * (u0 + u);
      const float dd = (wact && j < E) ? ds[j] : 0.f;
      acc.x = fmaf(dd, w[u].x, acc.x); acc.y = fmaf(dd, w[u].y, acc.y); acc.z = fmaf(dd, w[u].z, acc.z); acc.w = fmaf(dd, w[u].w, acc.w);
    }
    __builtin_amdgcn_sched_barrier(0);                           // (the second batch's requests stay behind the first batch's products)
  }
  // the rows' own operands are requested HERE: the W1 registers are dead, and the requests travel under the two barriers and the
  // group sums below (requested up front next to the W1 rows they pushed the kernel over its 128 registers per lane)
  const unsigned ldv = (unsigned)a.ldv, lddu = (unsigned)a.lddu, col = 4u * (unsigned)(lig < F4 ? lig : 0);
  float4 vv[DU_CHUNK / 32];
  float ri[DU_CHUNK / 32];
  int rows[DU_CHUNK / 32];                                       // (rows < 2^31 / ld: checked by the entry point)
#pragma unroll
  for (int u = 0; u < DU_CHUNK / 32; ++u) {
    const int n = lo + rg + 32 * u;
    rows[u] = n < hi ? g0 + n : -1;
    const unsigned rr = rows[u] >= 0 ? (unsigned)rows[u] : 0u;
    vv[u] = ld4(a.v + (size_t)(rr * ldv + col));
    ri[u] = a.rinv[rr];
  }
  const bool has_ghost = ghost_job && sz < a.n_ghost_rows;      // (a graph that fills every slot has no padded row)
  const int grow = (int)a.n_real + (has_ghost ? sz : 0);
  float4 gv = z4;
  float gri = 0.f;
  if (ghost_job && rg == 0 && a.n_ghost_rows > 0) { gv = ld4(a.v + (size_t)((unsigned)grow * ldv + col)); gri = a.rinv[grow]; }
  const int4 win = *reinterpret_cast<const int4*>(a.arg + (size_t)((unsigned)b * (unsigned)a.F + col));
  if (wact) *reinterpret_cast<float4*>(part + jg * a.F + 4 * lig) = acc;
  __syncthreads();
  for (int k = tid; k < a.F; k += NTH) {
    float t = 0.f;
    for (int q = 0; q < G; ++q) t += part[q * a.F + k];
    dseg[k] = t;
  }
  __syncthreads();
  const float4 gd = lig < F4 ? *reinterpret_cast<const float4*>(dseg + 4 * lig) : z4;
  // ---- rows
#pragma unroll
  for (int u = 0; u < DU_CHUNK / 32; ++u) {
    const int r32 = rows[u];
    float4 dyv = z4;
    if (rows[u] >= 0 && lig < F4) {
      if (win.x == r32) dyv.x = gd.x;
      if (win.y == r32) dyv.y = gd.y;
      if (win.z == r32) dyv.z = gd.z;
      if (win.w == r32) dyv.w = gd.w;
    } else {
      vv[u] = z4;
    }
    float dot = (vv[u].x * dyv.x + vv[u].y * dyv.y) + (vv[u].z * dyv.z + vv[u].w * dyv.w);
    dot = group_sum<32>(dot);
    if (rows[u] >= 0 && lig < F4) {
      if (ri[u] >= 0.999e12f) dot = 0.f;
      *reinterpret_cast<float4*>(a.du + (size_t)((unsigned)rows[u] * lddu + col)) =
          make_float4(ri[u] * (dyv.x - vv[u].x * dot), ri[u] * (dyv.y - vv[u].y * dot), ri[u] * (dyv.z - vv[u].z * dot),
                      ri[u] * (dyv.w - vv[u].w * dot));
    }
  }
  if (ghost_job && rg == 0) {                                    // (one whole half-wave: the cross-lane sum below is complete)
    const int r32 = grow;
    float4 dyv = z4;
    if (has_ghost && lig < F4) {
      if (win.x == r32) dyv.x = gd.x;
      if (win.y == r32) dyv.y = gd.y;
      if (win.z == r32) dyv.z = gd.z;
      if (win.w == r32) dyv.w = gd.w;
    } else {
      gv = z4;
    }
    float dot = (gv.x * dyv.x + gv.y * dyv.y) + (gv.z * dyv.z + gv.w * dyv.w);
    dot = group_sum<32>(dot);
    if (gri >= 0.999e12f) dot = 0.f;
    if (lig < F4)
      *reinterpret_cast<float4*>(a.du + (size_t)((unsigned)((int)a.n_real + b) * lddu + col)) =
          make_float4(gri * (dyv.x - gv.x * dot), gri * (dyv.y - gv.y * dot), gri * (dyv.z - gv.z * dot), gri * (dyv.w - gv.w * dot));
  }
}

template <int DU_CHUNK>
__global__ __launch_bounds__(64 * HW) void head2_bwd2_kernel(const float* __restrict__ out, int64_t ldo, const float* __restrict__ vec,
                                                         const float* __restrict__ dy_in, const float* __restrict__ dvec,
                                                         const float* __restrict__ w1, const float* __restrict__ w2, int B, int P, int E,
                                                         int C, float* __restrict__ dout, int64_t lddo, float* __restrict__ dw1,
                                                         float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                         float* __restrict__ normparts, CeArgs ce, DuArgs dua, RoTail rt) {
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  const int tid = threadIdx.x, NTH = 64 * HW;
  const int P4 = P >> 2, PP = P;                         // P % 4 == 0 on this path
  const bool has_ce = ce.label != nullptr;
  float* dyl = smem_all;                                 // [B * C]   (has_ce)
  float* lb = smem_all + (has_ce ? ((B * C + 3) & ~3) : 0);
  float* smem = lb + (has_ce ? ((B + 3) & ~3) : 0);      // role-specific region, 16-byte aligned
  const float* dy = has_ce ? dyl : dy_in;
  const int nj = (E + 3) / 4;
  if ((int)blockIdx.x >= B + nj + 1) {
    // ------------------------------------------------------------------------------------------------ last layer's dU rows
    const int d = (int)blockIdx.x - (B + nj + 1);
    int gb, gc, pad_k, pad_n;
    du_resolve<DU_CHUNK>(dua, d, B, gb, gc, pad_k, pad_n);
    if (has_ce && pad_k < 0) ce_rows(ce, B, C, dyl, lb, gb);       // (thread 0 rebuilds row gb; synchronised inside the role)
    head2_du_role<DU_CHUNK>(dua, smem, dy, gb, gc, pad_k, pad_n, dvec, w1, w2, B, P, E, C, has_ce);
    return;
  }

  if ((int)blockIdx.x < B) {
    // ------------------------------------------------------------------------------------------------ row block
    const int b = blockIdx.x;
    const int G = min(16, NTH / P4);                     // row groups (P4 <= 512: G >= 2)
    const int jg = tid / P4, k4 = tid - jg * P4;
    const bool active = jg < G;
    float* ds = smem;                                    // [E]
    float* part = smem + ((E + 3) & ~3);                 // [G][PP]
    // requests: W2 column(s) of thread j, dvec, then the first batch of W1 rows
    float w2r[4];
    float dv0 = 0.f;
    if (tid < E) {
#pragma unroll
      for (int c = 0; c < 4; ++c) w2r[c] = c < C ? w2[(int64_t)c * E + tid] : 0.f;
      if (dvec) dv0 = dvec[(int64_t)b * E + tid];
    }
    float4 w[RB_ROWS];
#pragma unroll
    for (int u = 0; u < RB_ROWS; ++u) {
      const int j = jg + G * u;
      // (unconditional from a clamped address; rows / groups beyond the matrix are multiplied by ds = 0 below: a load inside a
      // predicate is waited for before the next one is issued — sixteen dependent round trips here)
      w[u] = ld4(w1 + (int64_t)((active && j < E) ? j : 0) * P + 4 * (active ? k4 : 0));
    }
    if (has_ce) {
      ce_rows(ce, B, C, dyl, lb, b == 0 ? -1 : b);
      __syncthreads();
      if (b == 0) {
        // the loss value, summed exactly as tsgnn_softmax_ce_f32 sums it (thread t: rows t, t + 256, ...; DPP wave sums;
        // (w0 + w1) + (w2 + w3); / B) so that the deferred and the ordinary loss agree to the bit
        float* red = lb + ((B + 3) & ~3);                // start of the role region: not in use yet
        if (tid < 256) {
          float part = 0.f;
          for (int q = tid; q < B; q += 256) part += lb[q];
          part = wave_sum(part);
          if ((tid & 63) == 0) red[tid >> 6] = part;
        }
        __syncthreads();
        if (tid == 0) ce.loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
        __syncthreads();
      }
    }
    for (int j = tid; j < E; j += NTH) {                 // dvt row (same expression and order as the weight blocks)
      float acc = j == tid ? dv0 : (dvec ? dvec[(int64_t)b * E + j] : 0.f);
      for (int c = 0; c < C; ++c) acc = fmaf(dy[(int64_t)b * C + c], (j == tid && c < 4) ? w2r[c] : w2[(int64_t)c * E + j], acc);
      ds[j] = acc;
    }
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = 0; j0 < E; j0 += G * RB_ROWS) {
      if (j0 > 0) {
#pragma unroll
        for (int u = 0; u < RB_ROWS; ++u) {
          const int j = j0 + jg + G * u;
          w[u] = ld4(w1 + (int64_t)((active && j < E) ? j : 0) * P + 4 * (active ? k4 : 0));
        }
      }
#pragma unroll
      for (int u = 0; u < RB_ROWS; ++u) {
        const int j = j0 + jg + G * u;
        const float d = (active && j < E) ? ds[j] : 0.f;
        acc.x = fmaf(d, w[u].x, acc.x); acc.y = fmaf(d, w[u].y, acc.y); acc.z = fmaf(d, w[u].z, acc.z); acc.w = fmaf(d, w[u].w, acc.w);
      }
    }
    if (active) *reinterpret_cast<float4*>(part + jg * PP + 4 * k4) = acc;
    __syncthreads();
    for (int k = tid; k < P; k += NTH) {
      float a = 0.f;
      for (int q = 0; q < G; ++q) a += part[q * PP + k];
      dout[(int64_t)b * lddo + k] = a;
      if (rt.dz && k >= rt.c0 && k < rt.c0 + rt.F) {     // the tail readout's backward: graph b's rows of dz, every element
        const int f = k - rt.c0;
        const int win = rt.arg[(int64_t)b * rt.F + f];
        for (int n = 0; n < rt.N; ++n) rt.dz[((int64_t)b * rt.N + n) * rt.lddz + f] = (b * rt.N + n) == win ? a : 0.f;
      }
    }
    return;
  }
  // -------------------------------------------------------------------------------------------------- weight blocks
  const int jb = (int)blockIdx.x - B;
  float sq = 0.f;                                        // |gradient written by this thread|^2
  if (jb < nj) {
    const int j0 = 4 * jb;
    const int G = max(1, min(8, 768 / P4));              // graph groups: G * 4 * P floats of partial sums (<= 48 KB)
    const int bg = tid / P4, k4 = tid - bg * P4;
    const bool active = bg < G;
    float* dvs = smem;                                   // [B][4] dvt slice
    float* part = smem + 4 * ((B + 3) & ~3);             // [G][4][PP]
    // requests: W2 / dvec of the thread's (graph, column) of the slice, the first batch of `out` rows
    float w2r[4];
    float dv0 = 0.f;
    const int sb = tid >> 2, sj = j0 + (tid & 3);
    if (tid < 4 * B && sj < E) {
#pragma unroll
      for (int c = 0; c < 4; ++c) w2r[c] = c < C ? w2[(int64_t)c * E + sj] : 0.f;
      if (dvec) dv0 = dvec[(int64_t)sb * E + sj];
    }
    float4 x[WB_ROWS];
#pragma unroll
    for (int u = 0; u < WB_ROWS; ++u) {
      const int bb = bg + G * u;
      x[u] = ld4(out + (int64_t)((active && bb < B) ? bb : 0) * ldo + 4 * (active ? k4 : 0));      // (unconditional, see the row blocks;
    }                                                                                                 //  used under `active && bb < B` only)
    if (has_ce) { ce_rows(ce, B, C, dyl, lb, -1); __syncthreads(); }
    for (int i = tid; i < 4 * B; i += NTH) {
      const int bb = i >> 2, jj = j0 + (i & 3);
      float a = 0.f;
      if (jj < E) {
        a = i == tid ? dv0 : (dvec ? dvec[(int64_t)bb * E + jj] : 0.f);
        for (int c = 0; c < C; ++c) a = fmaf(dy[(int64_t)bb * C + c], (i == tid && c < 4) ? w2r[c] : w2[(int64_t)c * E + jj], a);
      }
      dvs[i] = a;
    }
    __syncthreads();
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    for (int b0 = 0; b0 < B; b0 += G * WB_ROWS) {
      if (b0 > 0) {
#pragma unroll
        for (int u = 0; u < WB_ROWS; ++u) {
          const int bb = b0 + bg + G * u;
          x[u] = ld4(out + (int64_t)((active && bb < B) ? bb : 0) * ldo + 4 * (active ? k4 : 0));
        }
      }
#pragma unroll
      for (int u = 0; u < WB_ROWS; ++u) {
        const int bb = b0 + bg + G * u;
        if (active && bb < B) {
          const float4 d = *reinterpret_cast<const float4*>(dvs + 4 * bb);
          a0.x = fmaf(d.x, x[u].x, a0.x); a0.y = fmaf(d.x, x[u].y, a0.y); a0.z = fmaf(d.x, x[u].z, a0.z); a0.w = fmaf(d.x, x[u].w, a0.w);
          a1.x = fmaf(d.y, x[u].x, a1.x); a1.y = fmaf(d.y, x[u].y, a1.y); a1.z = fmaf(d.y, x[u].z, a1.z); a1.w = fmaf(d.y, x[u].w, a1.w);
          a2.x = fmaf(d.z, x[u].x, a2.x); a2.y = fmaf(d.z, x[u].y, a2.y); a2.z = fmaf(d.z, x[u].z, a2.z); a2.w = fmaf(d.z, x[u].w, a2.w);
          a3.x = fmaf(d.w, x[u].x, a3.x); a3.y = fmaf(d.w, x[u].y, a3.y); a3.z = fmaf(d.w, x[u].z, a3.z); a3.w = fmaf(d.w, x[u].w, a3.w);
        }
      }
    }
    if (active) {
      float* pp = part + (int64_t)bg * 4 * PP + 4 * k4;
      *reinterpret_cast<float4*>(pp) = a0; *reinterpret_cast<float4*>(pp + PP) = a1;
      *reinterpret_cast<float4*>(pp + 2 * PP) = a2; *reinterpret_cast<float4*>(pp + 3 * PP) = a3;
    }
    __syncthreads();
    for (int i = tid; i < 4 * P; i += NTH) {
      const int jj = i / P, k = i - jj * P;
      if (j0 + jj < E) {
        float a = 0.f;
        for (int q = 0; q < G; ++q) a += part[(q * 4 + jj) * PP + k];      // group order: reproducible
        dw1[(int64_t)(j0 + jj) * P + k] = a;
        sq = fmaf(a, a, sq);
      }
    }
    if (db1 && tid < 4 && j0 + tid < E) {
      float a = 0.f;
      for (int bb = 0; bb < B; ++bb) a += dvs[4 * bb + tid];
      db1[j0 + tid] = a;
      sq = fmaf(a, a, sq);
    }
  } else {
    if (has_ce) { ce_rows(ce, B, C, dyl, lb, -1); __syncthreads(); }
    for (int i = tid; i < C * E; i += NTH) {
      const int c = i / E, j = i - c * E;
      float a = 0.f;
      int bb = 0;
      for (; bb + 8 <= B; bb += 8) {                     // eight rows of vec in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = vec[(int64_t)(bb + u) * E + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) a = fmaf(dy[(int64_t)(bb + u) * C + c], v[u], a);
      }
      for (; bb < B; ++bb) a = fmaf(dy[(int64_t)bb * C + c], vec[(int64_t)bb * E + j], a);
      dw2[i] = a;
      sq = fmaf(a, a, sq);
    }
    if (db2) for (int c = tid; c < C; c += NTH) {
      float a = 0.f;
      for (int bb = 0; bb < B; ++bb) a += dy[(int64_t)bb * C + c];
      db2[c] = a;
      sq = fmaf(a, a, sq);
    }
  }
  if (normparts) {                                       // block total in a fixed order: waves through LDS
    sq = wave_sum(sq);
    __syncthreads();
    if ((tid & 63) == 0) smem[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < HW; ++w) t += smem[w];
      normparts[jb] = t;
    }
  }
}

}  // namespace

extern "C" {

int tsgnn_head2_fwd_f32(const float* out, int64_t ldo, const float* w1, const float* b1, const float* w2, const float* b2, int B, int P,
                        int E, int C, float* vec, float* y, tsgnn_stream_t stream) {
  if (!out || !w1 || !w2 || !vec || !y || B <= 0 || P <= 0 || E <= 0 || C <= 0 || ldo < P) return TSGNN_EINVAL;
  if ((P % 4) || P > 4096 || E > 4096 || (reinterpret_cast<uintptr_t>(w1) & 15)) return TSGNN_EUNSUPPORTED;
  const size_t lds = sizeof(float) * (size_t)(((P + 3) & ~3) + E);
  TSGNN_KNAME("head2_fwd_kernel");
  head2_fwd_kernel<<<B, 64 * HW, lds, stream>>>(const_cast<float*>(out), ldo, w1, b1, w2, b2, P, E, C, vec, y, RoTail{});
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* tsgnn_head2_fwd_f32 whose launch first FILLS columns [c0, c0 + F) of its input with the max readout of a uniform level —
 * out[b, c0 + f] = max_n z[b * N + n, f], arg[b, f] = the winning row (what tsgnn_readout_max_fwd_f32 returns for the uniform batch
 * (B, N), N <= 64) — DiffPool's last pooled level (encoders.py:383,388-391) without a readout launch */
int tsgnn_head2_fwd_ro_f32(float* out, int64_t ldo, const float* w1, const float* b1, const float* w2, const float* b2, int B, int P,
                           int E, int C, float* vec, float* y, const float* z, int64_t ldz, int N, int c0, int F, int* arg,
                           tsgnn_stream_t stream) {
  if (!out || !w1 || !w2 || !vec || !y || !z || !arg || B <= 0 || P <= 0 || E <= 0 || C <= 0 || ldo < P || N <= 0 || F <= 0 || c0 < 0 ||
      c0 + F != P || ldz < F)
    return TSGNN_EINVAL;
  if ((P % 4) || P > 4096 || E > 4096 || N > 64 || (reinterpret_cast<uintptr_t>(w1) & 15)) return TSGNN_EUNSUPPORTED;
  const size_t lds = sizeof(float) * (size_t)(((P + 3) & ~3) + E);
  TSGNN_KNAME("head2_fwd_kernel(ro)");
  head2_fwd_kernel<<<B, 64 * HW, lds, stream>>>(out, ldo, w1, b1, w2, b2, P, E, C, vec, y, RoTail{z, ldz, N, c0, F, arg, nullptr, 0});
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_readout_head_fwd_f32(const unsigned long long* packed, int B, int L, int Fh, int Fl, const float* v_last, int64_t ldv,
                               const int* graph_ptr, int64_t n_real, int nslots, int n_ghost, float* out, int64_t ldo, int* arg,
                               const float* w1, const float* b1, const float* w2, const float* b2, int E, int C, float* vec, float* y,
                               tsgnn_stream_t stream) {
  if (!packed || !v_last || !graph_ptr || !out || !arg || !w1 || !w2 || !vec || !y || B <= 0 || L <= 0 || Fh <= 0 || Fl <= 0 ||
      E <= 0 || C <= 0 || nslots <= 0 || (n_ghost != 0 && n_ghost != nslots))
    return TSGNN_EINVAL;
  const int P = (L - 1) * Fh + Fl;
  if ((P % 4) || P > 4096 || E > 4096 || (Fl % 4) || Fl > 128 || (ldv % 4) || ldo < P || (reinterpret_cast<uintptr_t>(w1) & 15) ||
      (reinterpret_cast<uintptr_t>(v_last) & 15))
    return TSGNN_EUNSUPPORTED;
  ReadoutHeadArgs a{packed, B, L, Fh, Fl, v_last, ldv, graph_ptr, n_real, nslots, n_ghost, out, ldo, arg};
  const size_t lds = sizeof(float) * (size_t)(((P + 3) & ~3) + ((E + 3) & ~3)) + sizeof(unsigned long long) * (size_t)(2 * HW) * Fl;
  TSGNN_KNAME("readout_head_fwd_kernel");
  readout_head_fwd_kernel<<<B, 64 * HW, lds, stream>>>(a, w1, b1, w2, b2, P, E, C, vec, y);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_packed_head_fwd_f32(const unsigned long long* packed, int B, int L, int Fh, int Fl, float* out, int64_t ldo, int* arg,
                              const float* w1, const float* b1, const float* w2, const float* b2, int E, int C, float* vec, float* y,
                              tsgnn_stream_t stream) {
  if (!packed || !out || !arg || !w1 || !w2 || !vec || !y || B <= 0 || L <= 0 || Fh <= 0 || Fl <= 0 || E <= 0 || C <= 0) return TSGNN_EINVAL;
  const int P = (L - 1) * Fh + Fl;
  if ((P % 4) || P > 2048 || E > 8 * HW || ldo < P || (reinterpret_cast<uintptr_t>(w1) & 15)) return TSGNN_EUNSUPPORTED;
  const size_t lds = sizeof(float) * (size_t)(((P + 3) & ~3) + ((E + 3) & ~3));
  const ExpandRider rider = take_expand_rider();       // (no workgroups unless tsgnn_ingest_arm_expand_rider armed one on this thread)
  TSGNN_KNAME("packed_head_fwd_kernel<8>");
  packed_head_fwd_kernel<8><<<(unsigned)B + rider.blocks, 64 * HW, lds, stream>>>(packed, B, L, Fh, Fl, out, ldo, arg, w1, b1, w2, b2, P, E, C,
                                                                               vec, y, rider, nullptr, 0, 0);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* tsgnn_packed_head_fwd_f32 that also does the step's housekeeping: every decoded entry of packed is zeroed (ready for the next
 * step's atomicMax) and clear[0 .. clear_n) 64-bit words are zeroed (the integer sums of the fused slot batch-norms,
 * tsgnn_gather_rowgemm_st_f32 / tsgnn_sage_layer_fwd_bn_f32: all consumed by the launches before this one). */
int tsgnn_packed_head_fwd_z_f32(unsigned long long* packed, int B, int L, int Fh, int Fl, float* out, int64_t ldo, int* arg,
                                const float* w1, const float* b1, const float* w2, const float* b2, int E, int C, float* vec, float* y,
                                unsigned long long* clear, int64_t clear_n, tsgnn_stream_t stream) {
  if (!packed || !out || !arg || !w1 || !w2 || !vec || !y || B <= 0 || L <= 0 || Fh <= 0 || Fl <= 0 || E <= 0 || C <= 0 || clear_n < 0 ||
      (clear_n > 0 && !clear))
    return TSGNN_EINVAL;
  const int P = (L - 1) * Fh + Fl;
  if ((P % 4) || P > 2048 || E > 8 * HW || ldo < P || (reinterpret_cast<uintptr_t>(w1) & 15)) return TSGNN_EUNSUPPORTED;
  const size_t lds = sizeof(float) * (size_t)(((P + 3) & ~3) + ((E + 3) & ~3));
  const ExpandRider rider = take_expand_rider();
  TSGNN_KNAME("packed_head_fwd_kernel<8>");
  packed_head_fwd_kernel<8><<<(unsigned)B + rider.blocks, 64 * HW, lds, stream>>>(packed, B, L, Fh, Fl, out, ldo, arg, w1, b1, w2, b2, P, E, C,
                                                                               vec, y, rider, clear, clear_n, 1);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

static int head2_bwd_launch(const float* out, int64_t ldo, const float* vec, const float* dy, const float* dvec, const float* w1,
                            const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo, float* dw1, float* db1,
                            float* dw2, float* db2, float* normparts, const float* ce_y, const int64_t* ce_label, float* ce_loss,
                            tsgnn_stream_t stream, const DuArgs* du = nullptr, const RoTail* rtail = nullptr, int du_map_n = 0,
                            int du_map_chunk = 64, int du_compact = 0) {
  if (!out || !vec || (!dy && !ce_label) || !w1 || !w2 || !dout || !dw1 || !dw2 || B <= 0 || P <= 0 || E <= 0 || C <= 0) return TSGNN_EINVAL;
  if (ce_label && (!ce_y || !ce_loss)) return TSGNN_EINVAL;
  if ((P % 4) || P > 2048 || E > 4096 || B > 1024 || (reinterpret_cast<uintptr_t>(w1) & 15)) return TSGNN_EUNSUPPORTED;
  if ((ldo % 4) == 0 && (lddo % 4) == 0 && !((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout)) & 15) && (P / 4) <= 512) {
    // second-generation kernel: all operands requested up front (16-byte rows of `out` required)
    const int P4 = P / 4;
    const int Gr = 64 * HW / P4 < 16 ? 64 * HW / P4 : 16;
    const int Gw = 768 / P4 < 1 ? 1 : (768 / P4 > 8 ? 8 : 768 / P4);
    size_t role = (size_t)((E + 3) & ~3) + (size_t)Gr * P;
    const size_t wrole = 4 * (size_t)((B + 3) & ~3) + (size_t)Gw * 4 * P;
    if (wrole > role) role = wrole;
    if (role < HW) role = HW;
    size_t lds2 = sizeof(float) * role;
    DuArgs dua{};
    const RoTail rt = rtail ? *rtail : RoTail{};
    unsigned du_blocks = 0;
    int du_chunk = 64;
    if (du) {
      dua = *du;                                         // (du->chunks carries the largest graph's bound: see the entry point)
      static int ncu = 0;
      if (ncu == 0) {
        int dev = 0, v = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
      }
      const int nodes = dua.chunks;                      // bound on the largest graph
      const int other = B + (E + 3) / 4 + 1;
      unsigned compact_blocks = 0;
      if (dua.map) du_chunk = du_map_chunk;              // (the host listed the non-empty chunks of this size)
      else if (du_compact && B <= 64) {
        // the pairs resolved in the kernel: at most ceil(rows / chunk) + B of them, + a few blocks for the padding rows
        constexpr unsigned PADB = 4;
        unsigned n64 = (unsigned)((dua.n_real + 63) / 64) + (unsigned)B;
        if (other + (int)(n64 + PADB) > ncu) { du_chunk = 128; n64 = (unsigned)((dua.n_real + 127) / 128) + (unsigned)B; }
        compact_blocks = n64 + PADB;
        dua.compact = other;                             // (the role subtracts the launch's other blocks from gridDim.x)
      }
      else if ((B + 1) * ((nodes + 63) / 64) + other > ncu + ncu / 5) du_chunk = 128;
      dua.chunks = (nodes + du_chunk - 1) / du_chunk;
      const int G = 64 * HW / P4 < 16 ? 64 * HW / P4 : 16;
      const size_t drole = (size_t)((E + 3) & ~3) + (size_t)(G + 1) * dua.F;
      if (drole > role) role = drole;
      du_blocks = dua.map ? (unsigned)du_map_n : (compact_blocks ? compact_blocks : (unsigned)(B + 1) * (unsigned)dua.chunks);
    }
    lds2 = sizeof(float) * role;
    if (ce_label) lds2 += sizeof(float) * (size_t)(((B * C + 3) & ~3) + ((B + 3) & ~3));
    if (lds2 <= 160 * 1024 - 1024) {
      static size_t attr_set = 0;
      if (lds2 > 64 * 1024 && lds2 > attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(head2_bwd2_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(head2_bwd2_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        attr_set = lds2;
      }
      TSGNN_KNAME(du_chunk == 128 ? "head2_bwd2_kernel<128>" : "head2_bwd2_kernel<64>");
      if (du_chunk == 128)
        head2_bwd2_kernel<128><<<B + (E + 3) / 4 + 1 + du_blocks, 64 * HW, lds2, stream>>>(out, ldo, vec, dy, dvec, w1, w2, B, P, E, C, dout, lddo,
                                                                                        dw1, db1, dw2, db2, normparts,
                                                                                        CeArgs{ce_y, ce_label, ce_loss}, dua, rt);
      else
        head2_bwd2_kernel<64><<<B + (E + 3) / 4 + 1 + du_blocks, 64 * HW, lds2, stream>>>(out, ldo, vec, dy, dvec, w1, w2, B, P, E, C, dout, lddo,
                                                                                       dw1, db1, dw2, db2, normparts,
                                                                                       CeArgs{ce_y, ce_label, ce_loss}, dua, rt);
      TSGNN_CHECK_LAUNCH();
      return TSGNN_OK;
    }
  }
  if (du || rtail) return TSGNN_EUNSUPPORTED;          // the dU role / the tail readout ride in the second-generation kernel only
  size_t lds = sizeof(float) * (size_t)(((E + 3) & ~3) + HW * ((P + 3) & ~3));
  if (lds < sizeof(float) * (4 * (size_t)B + HW)) lds = sizeof(float) * (4 * (size_t)B + HW);
  if (ce_label) lds += sizeof(float) * (size_t)(((B * C + 3) & ~3) + ((B + 3) & ~3));
  TSGNN_KNAME("head2_bwd_kernel");
  head2_bwd_kernel<<<B + (E + 3) / 4 + 1, 64 * HW, lds, stream>>>(out, ldo, vec, dy, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2,
                                                                db2, normparts, ce_y, ce_label, ce_loss);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_head2_bwd_f32(const float* out, int64_t ldo, const float* vec, const float* dy, const float* dvec, const float* w1,
                        const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo, float* dw1, float* db1,
                        float* dw2, float* db2, float* normparts, tsgnn_stream_t stream) {
  return head2_bwd_launch(out, ldo, vec, dy, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2, db2, normparts, nullptr, nullptr,
                          nullptr, stream);
}

/* tsgnn_head2_bwd_f32 for a head whose input columns [c0, c0 + F) were filled by tsgnn_head2_fwd_ro_f32: the row block of graph b
 * also writes the gradient of that level's rows, dz[b * N + n, f] = (arg[b, f] == b * N + n) ? dout[b, c0 + f] : 0, every element
 * (the pass tsgnn_readout_max_bwd_rows_f32 would make).  dy NULL: the loss is folded in as in tsgnn_head2_bwd_ce_f32 (ce_y = the
 * logits, ce_label, ce_loss).  TSGNN_EUNSUPPORTED: shapes the second-generation kernel does not take. */
int tsgnn_head2_bwd_ro_f32(const float* out, int64_t ldo, const float* vec, const float* dy, const float* dvec, const float* w1,
                           const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo, float* dw1, float* db1,
                           float* dw2, float* db2, float* normparts, const int* arg, int c0, int F, int N, float* dz, int64_t lddz,
                           const float* ce_y, const int64_t* ce_label, float* ce_loss, tsgnn_stream_t stream) {
  if (!arg || !dz || N <= 0 || F <= 0 || c0 < 0 || c0 + F > P || lddz < F) return TSGNN_EINVAL;
  if ((dy == nullptr) == (ce_label == nullptr)) return TSGNN_EINVAL;         // dy given, or the loss folded in (ce_y, ce_label, ce_loss)
  const RoTail rt{nullptr, 0, N, c0, F, const_cast<int*>(arg), dz, lddz};
  return head2_bwd_launch(out, ldo, vec, dy, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2, db2, normparts, ce_y, ce_label,
                          ce_loss, stream, nullptr, &rt);
}

/* the same with the loss folded in: dy = d mean-softmax-cross-entropy(y, label) / dy is rebuilt inside the kernel and the loss
 * value written to loss[0] (F.cross_entropy of encoders.py:221-224 without a launch of its own) */
int tsgnn_head2_bwd_ce_f32(const float* out, int64_t ldo, const float* vec, const float* y, const int64_t* label, float* loss,
                           const float* dvec, const float* w1, const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo,
                           float* dw1, float* db1, float* dw2, float* db2, float* normparts, tsgnn_stream_t stream) {
  if (!y || !label || !loss) return TSGNN_EINVAL;
  return head2_bwd_launch(out, ldo, vec, nullptr, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2, db2, normparts, y, label, loss,
                          stream);
}

/* tsgnn_head2_bwd_ce_f32 / tsgnn_head2_bwd_f32 (y == NULL: dy given) whose launch ALSO produces dU of the stack's last GraphConv
 * layer (no batch-norm: dU is a row-wise function of the readout gradient — what tsgnn_readout_l2_bwd_f32 computes in a launch of
 * its own): extra workgroups (graph b, 128-row chunk) rebuild the F-wide segment [seg_off, seg_off + F) of dout[b, :] themselves.
 * v / rinv: the layer's output rows and 1/norm; arg [B, F]: its max-readout winners; du rows [0, n_real) are written, and
 * du[n_real + b] = graph b's ghost-row contribution (all ghost rows of this layer are identical; only the SUM of the rows behind the
 * real ones is ever used — the bias gradient — so pass bias_only_rows = B downstream).  n_ghost_rows: ghost rows that exist
 * (a graph with size >= n_ghost_rows has none); max_nodes = a bound on the largest graph.  TSGNN_EUNSUPPORTED: shapes the
 * second-generation backward kernel does not take (nothing launched: fall back to the two launches). */
int tsgnn_head2_bwd_du_f32(const float* out, int64_t ldo, const float* vec, const float* y, const int64_t* label, float* loss,
                           const float* dy, const float* dvec, const float* w1, const float* w2, int B, int P, int E, int C, float* dout,
                           int64_t lddo, float* dw1, float* db1, float* dw2, float* db2, float* normparts, const int* graph_ptr,
                           int64_t n_real, int n_ghost_rows, int max_nodes, const float* v, int64_t ldv, const float* rinv, const int* arg,
                           int seg_off, int F, float* du, int64_t lddu, tsgnn_stream_t stream) {
  return tsgnn_head2_bwd_du_map_f32(out, ldo, vec, y, label, loss, dy, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2, db2, normparts,
                                    graph_ptr, n_real, n_ghost_rows, max_nodes, v, ldv, rinv, arg, seg_off, F, du, lddu, nullptr, 0, 64, stream);
}

/* the same with the dU workgroups LISTED by the host: du_map[n_map] = graph << 8 | chunk for every chunk of du_chunk (64 or 128) rows
 * that holds rows (chunk 0 of EVERY graph: it also writes the graph's ghost contribution row; graph == B, chunks 0 .. k-1: the
 * padding rows of a capacity-padded batch, zero-filled by k workgroups) — an exact batch launches no workgroup that only returns.
 * du_map NULL: the dense (B + 1) x chunks grid of tsgnn_head2_bwd_du_f32. */
int tsgnn_head2_bwd_du_map_f32(const float* out, int64_t ldo, const float* vec, const float* y, const int64_t* label, float* loss,
                               const float* dy, const float* dvec, const float* w1, const float* w2, int B, int P, int E, int C,
                               float* dout, int64_t lddo, float* dw1, float* db1, float* dw2, float* db2, float* normparts,
                               const int* graph_ptr, int64_t n_real, int n_ghost_rows, int max_nodes, const float* v, int64_t ldv,
                               const float* rinv, const int* arg, int seg_off, int F, float* du, int64_t lddu, const int* du_map,
                               int n_map, int du_chunk, tsgnn_stream_t stream) {
  if (du_map && (n_map <= 0 || (du_chunk != 64 && du_chunk != 128) || max_nodes > 255 * du_chunk || B > (1 << 22))) return TSGNN_EINVAL;
  const int compact = (!du_map && n_map < 0) ? 1 : 0;           // n_map < 0 without a list: the pairs are resolved inside the kernel
  const int chunks = max_nodes;
  if (!graph_ptr || !v || !rinv || !arg || !du || n_real < 0 || n_ghost_rows < 0 || chunks <= 0 || F <= 0 || seg_off < 0) return TSGNN_EINVAL;
  if ((y == nullptr) == (dy == nullptr)) return TSGNN_EINVAL;
  if (y && (!label || !loss)) return TSGNN_EINVAL;
  if ((F % 4) || F > 128 || (seg_off % 4) || seg_off + F > P || (ldv % 4) || (lddu % 4) || ldv < F || lddu < F || chunks > (1 << 20) ||
      ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(arg)) & 15))
    return TSGNN_EUNSUPPORTED;
  if ((n_real + 1024 + B) * (ldv > lddu ? ldv : lddu) >= (int64_t)1 << 30 || (int64_t)E * P >= (int64_t)1 << 30) return TSGNN_EUNSUPPORTED;   // 32-bit offsets
  {
    const int P4 = P / 4, G = P4 > 0 ? (64 * HW / P4 < 16 ? 64 * HW / P4 : 16) : 0;
    if ((P % 4) || G < 1 || E > G * RB_ROWS || G * 32 > 64 * HW) return TSGNN_EUNSUPPORTED;     // one batch of W1 rows per row group (head2_du_role)
  }
  const DuArgs a{graph_ptr, n_real, n_ghost_rows, chunks, v, ldv, rinv, arg, seg_off, F, du, lddu, du_map, 0};
  return head2_bwd_launch(out, ldo, vec, dy, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2, db2, normparts, y, label, loss, stream, &a,
                          nullptr, n_map, du_chunk, compact);
}

}  // extern "C"
