// Fused per-node-slot kernels of the GraphSage-style encoder stack (encoders.py:177-205):
//
//   slot_bn_fwd : ReLU + slot batch-norm (statistics AND normalisation) in ONE pass — a workgroup owns slot n and
//                 keeps the slot's candidate rows of all graphs in registers (B <= 256/TPR graphs), so the
//                 statistics are exact two-pass (mean, then centred variance) without re-reading memory.
//   slot_post_bwd: backward of [max-readout scatter] + [slot batch-norm] + [ReLU] + [row L2-normalise] in one
//                 pass: dU for the `.W` GEMMs comes out directly.
//   readout_partial4 / readout_decode_layers: max readout with 16-byte loads; one decode for all layers.
//
// "Candidate (b, n)" = the real row graph_ptr[b]+n when n < size_b, else the ghost row n_real+n that stands for
// the reference's padded row.  Each ghost-using graph is one copy of the ghost row, so the
// reference's multiplicities fall out of the per-graph loop with no special weights.
#include "common.h"
#include <cstdlib>
#include "../../include/tsgnn.h"
#include "readout_body.h"

namespace {

constexpr float BN_EPS = 1e-5f;


template <int NW>                                        // NW waves per block (4, 8 or 16)
__device__ __forceinline__ void block_sum2(float& a, float& b, float* lds /* 2*NW */) {
  a = wave_sum(a); b = wave_sum(b);
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { lds[wid] = a; lds[NW + wid] = b; }
  __syncthreads();
  float ta = 0.f, tb = 0.f;
#pragma unroll
  for (int w = 0; w < NW; w += 4) {                      // fixed order
    ta += (lds[w] + lds[w + 1]) + (lds[w + 2] + lds[w + 3]);
    tb += (lds[NW + w] + lds[NW + w + 1]) + (lds[NW + w + 2] + lds[NW + w + 3]);
  }
  a = ta; b = tb;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------------------------- forward
// BT threads per block = TPR threads per graph x up to BT/TPR graphs (B <= 32: 256, <= 64: 512, <= 128: 1024 threads, so that a
// thread never owns more than NV <= 4 float4 of a row: short dependency chains, <= 80 registers)
// second problem of a PAIRED launch (grid.y = 2: two stacks on the same graph and shapes, sage_stack._SageStackPair): the
// operands that differ; blockIdx.y == 1 works on these
struct SlotFwdAlt { const float* v; float* mean; float* rstd; float* y; };
struct SlotBwdAlt { const float* v; const float* dxs; const float* dxs2; const float* mean; const float* rstd; const float* rinv; float* du; };

template <int TPR, int NV, int BT>
__global__ __launch_bounds__(BT) void slot_bn_fwd(SlotArgs s, const float* __restrict__ v, int64_t ldv, int F4, int relu,
                                                   float* __restrict__ mean, float* __restrict__ rstd,
                                                   float* __restrict__ y, int64_t ldy,
                                                   unsigned long long* __restrict__ zero_ptr, int64_t zero_n, SlotFwdAlt alt) {
  if (blockIdx.y) { v = alt.v; mean = alt.mean; rstd = alt.rstd; y = alt.y; zero_n = 0; }
  constexpr int NW = BT / 64;
  __shared__ float red[2 * NW];
  __shared__ int first_ghost;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = tid / TPR, c = tid % TPR;
  // optional: clear the packed max-readout buffer of the whole stack (consumed only by later launches)
  for (int64_t i = (int64_t)blockIdx.x * BT + tid; i < zero_n; i += (int64_t)gridDim.x * BT) zero_ptr[i] = 0ull;
  if (tid == 0) first_ghost = 0x7fffffff;
  __syncthreads();
  int64_t row = -1;
  bool ghost = false;
  if (b < s.B) {
    const int g0 = s.graph_ptr[b], sz = s.graph_ptr[b + 1] - g0;
    if (n < sz) row = (int64_t)g0 + n;
    else if (s.n_ghost) { row = s.n_real + n; ghost = true; }
  }
  if (ghost && c == 0) atomicMin(&first_ghost, b);
  float4 x[NV];
  float s1 = 0.f, dummy = 0.f;
  // all NV requests first, unconditionally and from clamped (always mapped) addresses, masked afterwards: a load inside its own
  // `if` is waited for before the next one is issued (the ISA had four load -> s_waitcnt vmcnt(0) pairs here: four dependent
  // round trips in a kernel that is one trip of useful memory time)
  {
    const float* vr = v + (row >= 0 ? row : 0) * ldv;
#pragma unroll
    for (int q = 0; q < NV; ++q) { const int c4 = c + TPR * q; x[q] = ld4(vr + 4 * (c4 < F4 ? c4 : 0)); }
  }
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c4 = c + TPR * q;
    if (row >= 0 && c4 < F4) {
      if (relu) { x[q].x = fmaxf(x[q].x, 0.f); x[q].y = fmaxf(x[q].y, 0.f); x[q].z = fmaxf(x[q].z, 0.f); x[q].w = fmaxf(x[q].w, 0.f); }
      s1 += (x[q].x + x[q].y) + (x[q].z + x[q].w);
    } else {
      x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const int have = s.slot_count[n];
  const float cnt = (float)(s.n_ghost ? s.B : have) * (float)(4 * F4);
  block_sum2<NW>(s1, dummy, red);
  const float mu = cnt > 0.f ? s1 / cnt : 0.f;
  float s2 = 0.f;
  dummy = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c4 = c + TPR * q;
    if (row >= 0 && c4 < F4) {
      const float a = x[q].x - mu, bb = x[q].y - mu, cc = x[q].z - mu, d = x[q].w - mu;
      s2 += (a * a + bb * bb) + (cc * cc + d * d);
    }
  }
  block_sum2<NW>(s2, dummy, red);
  const float rs = 1.0f / sqrtf((cnt > 0.f ? s2 / cnt : 0.f) + BN_EPS);
  if (tid == 0) { mean[n] = mu; rstd[n] = rs; }
  const bool writer = row >= 0 && (!ghost || b == first_ghost);      // first_ghost is final: block_sum2 synchronised
  if (writer) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int c4 = c + TPR * q;
      if (c4 < F4) st4(y + row * ldy + 4 * c4, make_float4((x[q].x - mu) * rs, (x[q].y - mu) * rs, (x[q].z - mu) * rs, (x[q].w - mu) * rs));
    }
  }
  // a ghost row that no graph uses (every graph has this slot) is never read for its value, but keep it defined
  if (s.n_ghost && have == s.B && b == 0) {
    const int64_t gr = s.n_real + n;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int c4 = c + TPR * q;
      if (c4 < F4) st4(y + gr * ldy + 4 * c4, make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward
// dy(b,n,f) = dxs[row] + dxs2[row] (real rows; a ghost row has no edges, so no gradient reaches it through the aggregation)
//           + (arg[b,f] == row ? dout[b,f] : 0)                       [max-readout winner of graph b]
// BN:   dv = rstd (dy - m1 - xhat m2) ; ReLU mask ; ghost copies summed in graph order ;
// L2:   du = rinv (dv - v <v,dv>)   (rinv = 1e12 marks the clamped norm: du = rinv dv)
// slot n of the launch.  TO_LDS: the rows of dU do not go to memory but into the tile u_lds[B <= BT / TPR][F] (zeroed by the caller;
// candidate b's row at u_lds + b * F — ghost copies other than the first stay zero, the first carries their sum), for a caller that
// consumes them at once (slot_post_wgrad_kernel: layer 0's weight gradient, whose dU nobody else reads).
template <int TPR, int NV, int BT, bool TO_LDS>
__device__ __forceinline__ void slot_post_body(const SlotArgs& s, const int n, float* smem, const float* __restrict__ v, int64_t ldv,
                                               const float* __restrict__ dxs, int64_t lddxs, const float* __restrict__ dxs2,
                                               int64_t lddxs2, const float* __restrict__ dout, int64_t ldo, const int* __restrict__ arg,
                                               int F4, int relu, int bn, const float* __restrict__ mean,
                                               const float* __restrict__ rstd, const float* __restrict__ rinv,
                                               float* __restrict__ du, int64_t lddu, float* u_lds) {
  constexpr int NW = BT / 64;
  const int F = 4 * F4;
  float* gacc = smem;                                                // [NW waves][F] ghost-copy sums (only with ghost rows)
  float* red = smem + (s.n_ghost ? NW * F : 0);                      // 2 * NW floats
  int& first_ghost = *reinterpret_cast<int*>(red + 2 * NW);
  const int tid = threadIdx.x;
  const int b = tid / TPR, c = tid % TPR;
  TR(0);
  if (tid == 0) first_ghost = 0x7fffffff;
  if constexpr (!TO_LDS) {
    // capacity-padded batches (ingest.hip): rows [graph_ptr[B], n_real) belong to no graph; nothing below writes their du,
    // and the weight / bias gradients sum du over all n_real rows: they are zeroed here (no-op for exact batches)
    const int64_t pad_lo = s.graph_ptr[s.B], npad = s.n_real - pad_lo;
    for (int64_t i = (int64_t)blockIdx.x * BT + tid; i < npad * F4; i += (int64_t)gridDim.x * BT) {
      const int64_t r = pad_lo + i / F4;
      st4(du + r * lddu + 4 * (i % F4), make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
  // the readout winners and their gradients depend on (graph, column) only: issued first, they fly while the row is resolved
  int4 wq[NV];
  float4 gq[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c4 = c + TPR * q;
    wq[q] = make_int4(-1, -1, -1, -1);
    gq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (arg && b < s.B && c4 < F4) {                        // arg == NULL: this layer has no readout
      wq[q] = *reinterpret_cast<const int4*>(arg + (int64_t)b * F + 4 * c4);
      gq[q] = ld4(dout + (int64_t)b * ldo + 4 * c4);
    }
  }
  // the candidate row (one trip to graph_ptr), then EVERY request of the row at once — before the two barriers of the first-ghost
  // protocol, which used to sit between the two trips
  int64_t row = -1;
  bool ghost = false;
  if (b < s.B) {
    const int g0 = s.graph_ptr[b], sz = s.graph_ptr[b + 1] - g0;
    if (n < sz) row = (int64_t)g0 + n;
    else if (s.n_ghost) { row = s.n_real + n; ghost = true; }
  }
  TR(1);
  const float mu = bn ? mean[n] : 0.f, rs = bn ? rstd[n] : 1.f;
  float4 vv[NV], dy[NV], d2[NV];
  float a1 = 0.f, a2 = 0.f;
  // every request of the row first (v, dxs, dxs2: up to 3 NV loads), unconditionally from clamped addresses, masked below: loads
  // inside their own `if` were NV dependent round trips
  float ri_pre;                                                   // 1 / norm of the row: requested with the row (a load at its use, behind
  {                                                               // the block reductions, was one more dependent round trip at the end)
    const int64_t rv = row >= 0 ? row : 0, rd = (row >= 0 && !ghost) ? row : 0;
    ri_pre = rinv[rv];
    // (dxs / dxs2 are uniform over the grid: ONE branch on them, then straight-line request blocks — a null test per load puts a
    // scalar branch and a full wait between the requests of consecutive q)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dxs && dxs2) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c4 = c + TPR * q, cc = 4 * (c4 < F4 ? c4 : 0);
        vv[q] = ld4(v + rv * ldv + cc); dy[q] = ld4(dxs + rd * lddxs + cc); d2[q] = ld4(dxs2 + rd * lddxs2 + cc);
      }
    } else if (dxs) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c4 = c + TPR * q, cc = 4 * (c4 < F4 ? c4 : 0);
        vv[q] = ld4(v + rv * ldv + cc); dy[q] = ld4(dxs + rd * lddxs + cc); d2[q] = z4;
      }
    } else if (dxs2) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c4 = c + TPR * q, cc = 4 * (c4 < F4 ? c4 : 0);
        vv[q] = ld4(v + rv * ldv + cc); dy[q] = z4; d2[q] = ld4(dxs2 + rd * lddxs2 + cc);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c4 = c + TPR * q, cc = 4 * (c4 < F4 ? c4 : 0);
        vv[q] = ld4(v + rv * ldv + cc); dy[q] = z4; d2[q] = z4;
      }
    }
  }
  __syncthreads();                                                // first_ghost is initialised
  if (ghost && c == 0) atomicMin(&first_ghost, b);
  __syncthreads();
  const bool fg = ghost && b == first_ghost;
  const bool any_ghost = first_ghost != 0x7fffffff;               // uniform
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c4 = c + TPR * q;
    if (!(row >= 0 && c4 < F4)) {
      vv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      dy[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      if (!(dxs && !ghost)) dy[q] = make_float4(0.f, 0.f, 0.f, 0.f);   // nothing aggregates from a ghost row: its dxs is 0
      if (dxs2 && !ghost) {                               // gradient that reaches this layer's output directly (node-level outputs;
        const float4 t = d2[q];                               // ghost rows are masked out of those)
        dy[q].x += t.x; dy[q].y += t.y; dy[q].z += t.z; dy[q].w += t.w;
      }
      const int4 w = wq[q];
      const float4 g = gq[q];
      const int r32 = (int)row;
      if (w.x == r32) dy[q].x += g.x;
      if (w.y == r32) dy[q].y += g.y;
      if (w.z == r32) dy[q].z += g.z;
      if (w.w == r32) dy[q].w += g.w;
      if (bn) {
        const float xa = ((relu ? fmaxf(vv[q].x, 0.f) : vv[q].x) - mu) * rs, xb = ((relu ? fmaxf(vv[q].y, 0.f) : vv[q].y) - mu) * rs;
        const float xc = ((relu ? fmaxf(vv[q].z, 0.f) : vv[q].z) - mu) * rs, xd = ((relu ? fmaxf(vv[q].w, 0.f) : vv[q].w) - mu) * rs;
        a1 += (dy[q].x + dy[q].y) + (dy[q].z + dy[q].w);
        a2 += (dy[q].x * xa + dy[q].y * xb) + (dy[q].z * xc + dy[q].w * xd);
      }
    }
  }
  TR(2);
  float m1 = 0.f, m2 = 0.f;
  if (bn) {
    const int have = s.slot_count[n];
    const float cnt = (float)(s.n_ghost ? s.B : have) * (float)F;
    block_sum2<NW>(a1, a2, red);
    m1 = cnt > 0.f ? a1 / cnt : 0.f;
    m2 = cnt > 0.f ? a2 / cnt : 0.f;
  }
  TR(3);
  // dv per candidate
  float4 dv[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const float e[4] = {vv[q].x, vv[q].y, vv[q].z, vv[q].w};
    const float d[4] = {dy[q].x, dy[q].y, dy[q].z, dy[q].w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float g = d[k];
      if (bn) {
        const float xh = ((relu ? fmaxf(e[k], 0.f) : e[k]) - mu) * rs;
        g = rs * (d[k] - m1 - xh * m2);
      }
      if (relu && !(e[k] > 0.f)) g = 0.f;
      o[k] = (row >= 0) ? g : 0.f;
    }
    dv[q] = make_float4(o[0], o[1], o[2], o[3]);
  }
  TR(4);
  // ghost copies of this slot are summed: across the graphs of a wave with lane exchanges (lanes TPR apart hold the same
  // columns of consecutive graphs), across the four waves through 4 x F floats of LDS; the first ghost copy takes the total
  if (s.n_ghost && any_ghost) {
    float4 gs[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) gs[q] = ghost ? dv[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int d = TPR; d < 64; d <<= 1) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        gs[q].x += __shfl_xor(gs[q].x, d, 64); gs[q].y += __shfl_xor(gs[q].y, d, 64);
        gs[q].z += __shfl_xor(gs[q].z, d, 64); gs[q].w += __shfl_xor(gs[q].w, d, 64);
      }
    }
    const int lane = tid & 63, wid = tid >> 6;
    if (lane < TPR) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c4 = c + TPR * q;
        if (c4 < F4) st4(gacc + wid * F + 4 * c4, gs[q]);
      }
    }
    __syncthreads();
    if (fg) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c4 = c + TPR * q;
        if (c4 < F4) {
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int w = 0; w < NW; w += 4) {                            // fixed order
            const float4 t0 = ld4(gacc + (w + 0) * F + 4 * c4), t1 = ld4(gacc + (w + 1) * F + 4 * c4),
                         t2 = ld4(gacc + (w + 2) * F + 4 * c4), t3 = ld4(gacc + (w + 3) * F + 4 * c4);
            t.x += (t0.x + t1.x) + (t2.x + t3.x); t.y += (t0.y + t1.y) + (t2.y + t3.y);
            t.z += (t0.z + t1.z) + (t2.z + t3.z); t.w += (t0.w + t1.w) + (t2.w + t3.w);
          }
          dv[q] = t;
        }
      }
    }
  }
  TR(5);
  // row L2-normalise backward (real rows, and the ghost row by its first copy)
  const bool writer = row >= 0 && (!ghost || fg);
  float dot = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) dot += (vv[q].x * dv[q].x + vv[q].y * dv[q].y) + (vv[q].z * dv[q].z + vv[q].w * dv[q].w);
  dot = group_sum<TPR>(dot);
  if (writer) {
    const float ri = ri_pre;
    if (ri >= 0.999e12f) dot = 0.f;
    float* dst = TO_LDS ? u_lds + b * F : du + row * lddu;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int c4 = c + TPR * q;
      if (c4 < F4)
        st4(dst + 4 * c4, make_float4(ri * (dv[q].x - vv[q].x * dot), ri * (dv[q].y - vv[q].y * dot),
                                      ri * (dv[q].z - vv[q].z * dot), ri * (dv[q].w - vv[q].w * dot)));
    }
  }
  if constexpr (!TO_LDS) {
    if (s.n_ghost && s.slot_count[n] == s.B && b == 0) {         // unused ghost row: zero gradient
      const int64_t gr = s.n_real + n;
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c4 = c + TPR * q;
        if (c4 < F4) st4(du + gr * lddu + 4 * c4, make_float4(0.f, 0.f, 0.f, 0.f));
      }
    }
  }
  TR(6);
}

template <int TPR, int NV, int BT>
__global__ __launch_bounds__(BT) void slot_post_bwd(SlotArgs s, const float* __restrict__ v, int64_t ldv,
                                                     const float* __restrict__ dxs, int64_t lddxs,
                                                     const float* __restrict__ dxs2, int64_t lddxs2,
                                                     const float* __restrict__ dout, int64_t ldo, const int* __restrict__ arg,
                                                     int F4, int relu, int bn, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ rinv,
                                                     float* __restrict__ du, int64_t lddu, SlotBwdAlt alt) {
  if (blockIdx.y) { v = alt.v; dxs = alt.dxs; dxs2 = alt.dxs2; mean = alt.mean; rstd = alt.rstd; rinv = alt.rinv; du = alt.du; }
  // all LDS in ONE dynamic array (16-byte aligned base for the float4 ghost accumulators, Guideline 17)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  slot_post_body<TPR, NV, BT, false>(s, (int)blockIdx.x, smem, v, ldv, dxs, lddxs, dxs2, lddxs2, dout, ldo, arg, F4, relu, bn, mean, rstd, rinv, du,
                                     lddu, nullptr);
  TR_END();
}

// ---------------------------------------------------------------------------------------------- layer 0: dU and its only consumer
// The FIRST layer's dU has one consumer — its weight / bias gradient (the input features need no gradient) — so the rows never
// go to memory: a persistent workgroup walks slots n = blockIdx.x, + gridDim.x, ...; slot_post_body leaves the slot's <= 32 rows of dU
// in an LDS tile, the rows of z (= A x0, kept by the forward) of the same candidates are staged beside it, and the chunk of 32 rows
// goes through the slab body's MFMA step (tn_rows_body.h: A[m][k] = Z[row k][m], B[k][j] = dU[row k][j], wave w owns column tile w)
// into accumulators that live across the slots.  One slab per workgroup at the end, summed by tsgnn_wgrad_reduce_multi_f32 like any
// other.  Replaces slot_post_bwd + gemm_tn_rows_kernel of layer 0 (two launches, 4.2 MB of dU written and read back).  B <= 32, F = 128.
typedef float sp_f32x16 __attribute__((ext_vector_type(16)));
constexpr int SPW_AUX = 528;                             // gacc [4][128] + red [8] + first_ghost, rounded to 16 bytes
template <int MT>
__global__ __launch_bounds__(256) void slot_post_wgrad_kernel(SlotArgs s, const float* __restrict__ v, int64_t ldv,
                                                              const float* __restrict__ dxs, int64_t lddxs,
                                                              const float* __restrict__ dout, int64_t ldo, const int* __restrict__ arg,
                                                              int relu, int bn, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ rinv,
                                                              const float* __restrict__ z, int64_t ldz, int K_in,
                                                              float* __restrict__ slabs) {
  constexpr int F = 128, F4 = 32, TPR = 8, KP = 32 * MT, KQ = (KP / 4 + TPR - 1) / TPR;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* aux = smem;
  float* Us = smem + SPW_AUX;                            // [2 stages][32][F]
  float* Zs = Us + 2 * 32 * F;                           // [2 stages][32][KP]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, h = lane >> 5;
  const int b = tid / TPR, c = tid % TPR;
  sp_f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbacc = 0.f;
  int it = 0;
  for (int n = blockIdx.x; n < s.nmax; n += gridDim.x, ++it) {
    float* U = Us + (it & 1) * 32 * F;
    float* Z = Zs + (it & 1) * 32 * KP;
    // the z rows of the slot's REAL candidates (a ghost row aggregates nothing: z = 0), requested first
    int64_t zrow = -1;
    if (b < s.B) {
      const int g0 = s.graph_ptr[b], sz = s.graph_ptr[b + 1] - g0;
      if (n < sz) zrow = (int64_t)g0 + n;
    }
    float4 zq[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int c4 = c + TPR * q;
      zq[q] = ld4(z + (zrow >= 0 ? zrow : 0) * ldz + 4 * ((4 * c4 < K_in) ? c4 : 0));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(U + 4 * (tid + 256 * q), make_float4(0.f, 0.f, 0.f, 0.f));     // rows nobody writes stay zero
    slot_post_body<TPR, 4, 256, true>(s, n, aux, v, ldv, dxs, lddxs, nullptr, 0, dout, ldo, arg, F4, relu, bn, mean, rstd, rinv, nullptr, 0, U);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int c4 = c + TPR * q;
      if (c4 < KP / 4) {
        float4 t = (zrow >= 0 && 4 * c4 < K_in) ? zq[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int nv = K_in - 4 * c4;                    // (K_in % 4 may be non-zero: the row padding of z contributes nothing)
        if (nv < 4) t.w = 0.f;
        if (nv < 3) t.z = 0.f;
        if (nv < 2) t.y = 0.f;
        st4(Z + b * KP + 4 * c4, t);
      }
    }
    __syncthreads();
    // the chunk's MFMA step: 32 rows = 16 k-steps
    float bfr[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) bfr[s2] = U[(2 * s2 + h) * F + wid * 32 + i];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float afr[16];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) afr[s2] = Z[(2 * s2 + h) * KP + t * 32 + i];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[s2], bfr[s2], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) dbacc += bfr[s2];
    // (no barrier here: the next slot writes the OTHER stage, and its slot_post_body synchronises before anybody gets back to this one)
  }
  float* slab = slabs + (int64_t)blockIdx.x * (K_in + 1) * F;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int cn = wid * 32 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cm = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (cm < K_in) slab[(int64_t)cm * F + cn] = acc[t][r];
    }
  }
  {
    const float vsum = dbacc + __shfl_xor(dbacc, 32, 64);      // rows of parity h = 0 and h = 1
    if (lane < 32) slab[(int64_t)K_in * F + wid * 32 + i] = vsum;
  }
  TR_END();
}

// Last layer of the stack (no batch-norm and no ReLU follow it, no layer above): dU from the max-readout gradient alone — a
// row-parallel pass instead of the slot-structured one (6.9 -> ~3 us on the DD batch):
//   real row r of graph b      dy[f] = (arg[b, f] == r) ? dout[b, f] : 0 ;  du = rinv (dy - v <v, dy>)   (clamped norm: du = rinv dy)
//   ghost row n_real + n       only graphs with exactly n nodes can have it as their winner (their first ghost row): the same
//                              formula on the sum of those graphs' dy, in graph order
//   padding rows (row_graph >= B, capacity-padded batches) and ghost rows nobody can win: 0
// 32 lanes (one float4 each) per row, 8 rows per block.
__global__ __launch_bounds__(256) void readout_l2_bwd_rows(SlotArgs s, const int* __restrict__ row_graph, const float* __restrict__ v,
                                                           int64_t ldv, const float* __restrict__ dout, int64_t ldo,
                                                           const int* __restrict__ arg, int F4, const float* __restrict__ rinv,
                                                           float* __restrict__ du, int64_t lddu, int n_ghost_rows) {
  const int lig = threadIdx.x & 31;
  const int64_t r = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int64_t total = s.n_real + n_ghost_rows;
  const bool rok = r < total;                             // whole waves stay active (cross-lane sums below)
  const bool live = rok && lig < F4;
  const int F = 4 * F4;
  float4 dy = make_float4(0.f, 0.f, 0.f, 0.f), vv = dy;
  float ri = 0.f;
  if (live) {
    vv = ld4(v + r * ldv + 4 * lig);
    ri = rinv[r];
    const int r32 = (int)r;
    if (r < s.n_real) {
      const int b = row_graph[r];
      if (b < s.B) {
        const int4 w = *reinterpret_cast<const int4*>(arg + (int64_t)b * F + 4 * lig);
        const float4 g = ld4(dout + (int64_t)b * ldo + 4 * lig);
        if (w.x == r32) dy.x = g.x;
        if (w.y == r32) dy.y = g.y;
        if (w.z == r32) dy.z = g.z;
        if (w.w == r32) dy.w = g.w;
      }
    } else {
      const int n = (int)(r - s.n_real);
      for (int b = 0; b < s.B; ++b) {                     // graphs with exactly n nodes (few), graph order
        if (s.graph_ptr[b + 1] - s.graph_ptr[b] != n) continue;
        const int4 w = *reinterpret_cast<const int4*>(arg + (int64_t)b * F + 4 * lig);
        const float4 g = ld4(dout + (int64_t)b * ldo + 4 * lig);
        if (w.x == r32) dy.x += g.x;
        if (w.y == r32) dy.y += g.y;
        if (w.z == r32) dy.z += g.z;
        if (w.w == r32) dy.w += g.w;
      }
    }
  }
  float dot = (vv.x * dy.x + vv.y * dy.y) + (vv.z * dy.z + vv.w * dy.w);
  dot = group_sum<32>(dot);
  if (live) {
    if (ri >= 0.999e12f) dot = 0.f;
    st4(du + r * lddu + 4 * lig, make_float4(ri * (dy.x - vv.x * dot), ri * (dy.y - vv.y * dot), ri * (dy.z - vv.z * dot),
                                             ri * (dy.w - vv.w * dot)));
  }
}

// ---------------------------------------------------------------------------------------------- readout
// grid (ceil(nslots/64), B); block = (256/G) row lanes x G float4 lanes (G = 32: F <= 128); body in readout_body.h
template <int G>
__global__ __launch_bounds__(256) void readout_partial4(SlotArgs s, const float* __restrict__ x, int64_t ld, int F4,
                                                        unsigned long long* __restrict__ packed) {
  __shared__ unsigned long long best[256 / G][4 * G];
  readout_partial_body<G>(s, x, ld, F4, packed, blockIdx.x, blockIdx.y, &best[0][0]);
}
// packed: layers 0..L-2 hold B*Fh entries each, the last layer B*Fl; out[b, off_l + f]; arg in the packed layout
__global__ void readout_decode_layers(const unsigned long long* __restrict__ packed, int B, int L, int Fh, int Fl,
                                      float* __restrict__ out, int64_t ldo, int* __restrict__ arg) {
  const int64_t total = (int64_t)B * ((int64_t)(L - 1) * Fh + Fl);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t per_h = (int64_t)B * Fh;
  int l;
  int64_t j;
  if (i < per_h * (L - 1)) { l = (int)(i / per_h); j = i % per_h; }
  else { l = L - 1; j = i - per_h * (L - 1); }
  const int Fw = (l == L - 1) ? Fl : Fh;
  const int b = (int)(j / Fw), f = (int)(j % Fw);
  const unsigned long long p = packed[i];
  out[(int64_t)b * ldo + (int64_t)l * Fh + f] = p ? ordered_f32((unsigned)(p >> 32)) : 0.f;
  arg[i] = p ? (int)(0xFFFFFFFFu - (unsigned)(p & 0xFFFFFFFFull)) : -1;
}

}  // namespace

extern "C" {

/* 0 if the fused slot kernels do not cover (B, F): callers use tsgnn_bn_slots_* / tsgnn_readout_max_* instead */
int tsgnn_slot_fused_supported(int B, int F) {
  if (B <= 0 || F <= 0 || (F % 4) || F > 128) return 0;
  return B <= 128;
}

// (TPR threads per graph row, NV float4 per thread, block threads): F/4 = TPR * NV lanes-worth of columns per row
// 64-wide rows of up to 16 graphs: 16 threads per row (one float4 each) fill the 256-thread workgroup that 8 threads x two float4 left half
// empty (DiffPool b16: five slot launches per step, 430-433 -> 428 us; TSGNN_SLOT_WIDE16=0: the table's entry)
static inline bool slot_wide16() {
  static const bool on = [] { const char* e = getenv("TSGNN_SLOT_WIDE16"); return e ? atoi(e) != 0 : true; }();
  return on;
}
#define TSGNN_SLOT_DISPATCH(KERNEL, GRID, LDS, ...)                                                                     \
  do {                                                                                                                   \
    const int F4_ = F / 4;                                                                                               \
    TSGNN_KNAME("%s<8,%d,%d>", #KERNEL, F4_ <= 8 ? 1 : (F4_ <= 16 ? 2 : 4), B <= 32 ? 256 : (B <= 64 ? 512 : 1024));     \
    if (B <= 32) {                                                                                                       \
      if (F4_ <= 8) KERNEL<8, 1, 256><<<GRID, 256, LDS, stream>>> __VA_ARGS__;                                            \
      else if (F4_ <= 16 && B <= 16 && slot_wide16()) KERNEL<16, 1, 256><<<GRID, 256, LDS, stream>>> __VA_ARGS__;         \
      else if (F4_ <= 16) KERNEL<8, 2, 256><<<GRID, 256, LDS, stream>>> __VA_ARGS__;                                      \
      else KERNEL<8, 4, 256><<<GRID, 256, LDS, stream>>> __VA_ARGS__;                                                     \
    } else if (B <= 64) {                                                                                                \
      if (F4_ <= 8) KERNEL<8, 1, 512><<<GRID, 512, LDS, stream>>> __VA_ARGS__;                                            \
      else if (F4_ <= 16) KERNEL<8, 2, 512><<<GRID, 512, LDS, stream>>> __VA_ARGS__;                                      \
      else KERNEL<8, 4, 512><<<GRID, 512, LDS, stream>>> __VA_ARGS__;                                                     \
    } else {                                                                                                             \
      if (F4_ <= 8) KERNEL<8, 1, 1024><<<GRID, 1024, LDS, stream>>> __VA_ARGS__;                                          \
      else if (F4_ <= 16) KERNEL<8, 2, 1024><<<GRID, 1024, LDS, stream>>> __VA_ARGS__;                                    \
      else KERNEL<8, 4, 1024><<<GRID, 1024, LDS, stream>>> __VA_ARGS__;                                                   \
    }                                                                                                                    \
  } while (0)

int tsgnn_slot_bn_fwd_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                          const float* v, int64_t ldv, int F, int relu, float* mean, float* rstd, float* y, int64_t ldy,
                          unsigned long long* zero_ptr, int64_t zero_n, tsgnn_stream_t stream) {
  if (!graph_ptr || !slot_count || !v || !mean || !rstd || !y || nmax <= 0 || (n_ghost != 0 && n_ghost != nmax) || ldv < F || ldy < F ||
      (ldv % 4) || (ldy % 4))
    return TSGNN_EINVAL;
  if (!tsgnn_slot_fused_supported(B, F)) return TSGNN_EUNSUPPORTED;
  SlotArgs s{graph_ptr, slot_count, B, nmax, n_real, n_ghost};
  TSGNN_SLOT_DISPATCH(slot_bn_fwd, nmax, 0, (s, v, ldv, F / 4, relu, mean, rstd, y, ldy, zero_ptr, zero_ptr ? zero_n : 0, SlotFwdAlt{}));
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* tsgnn_slot_bn_fwd_f32 for TWO feature matrices on the same batch and shapes in one launch (grid.y = 2) */
int tsgnn_slot_bn_fwd_pair_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                               const float* v0, const float* v1, int64_t ldv, int F, int relu, float* mean0, float* mean1, float* rstd0,
                               float* rstd1, float* y0, float* y1, int64_t ldy, tsgnn_stream_t stream) {
  if (!graph_ptr || !slot_count || !v0 || !v1 || !mean0 || !mean1 || !rstd0 || !rstd1 || !y0 || !y1 || nmax <= 0 ||
      (n_ghost != 0 && n_ghost != nmax) || ldv < F || ldy < F || (ldv % 4) || (ldy % 4))
    return TSGNN_EINVAL;
  if (!tsgnn_slot_fused_supported(B, F)) return TSGNN_EUNSUPPORTED;
  SlotArgs s{graph_ptr, slot_count, B, nmax, n_real, n_ghost};
  const unsigned long long* none = nullptr;
  TSGNN_SLOT_DISPATCH(slot_bn_fwd, dim3((unsigned)nmax, 2), 0,
                      (s, v0, ldv, F / 4, relu, mean0, rstd0, y0, ldy, const_cast<unsigned long long*>(none), 0, SlotFwdAlt{v1, mean1, rstd1, y1}));
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_slot_post_bwd_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                            const float* v, int64_t ldv, const float* dxs, int64_t lddxs, const float* dxs2, int64_t lddxs2,
                            const float* dout, int64_t ldo, const int* arg, int F, int relu, int bn, const float* mean,
                            const float* rstd, const float* rinv, float* du, int64_t lddu, tsgnn_stream_t stream) {
  if (!graph_ptr || !slot_count || !v || ((dout == nullptr) != (arg == nullptr)) || !rinv || !du || nmax <= 0 ||
      (n_ghost != 0 && n_ghost != nmax) || (bn && (!mean || !rstd)) || ldv < F || lddu < F || (ldv % 4) || (lddu % 4) ||
      (arg && (ldo % 4)) || (dxs && (lddxs % 4)) || (dxs2 && (lddxs2 % 4)))
    return TSGNN_EINVAL;
  if (!tsgnn_slot_fused_supported(B, F)) return TSGNN_EUNSUPPORTED;
  SlotArgs s{graph_ptr, slot_count, B, nmax, n_real, n_ghost};
  // 128-wide rows of up to 32 graphs: 16 threads per row (two float4 each) in 512-thread workgroups rather than the dispatch table's
  // 8 x four float4 in 256 — half the dependent chain per thread, twice the requests in flight per slot: the headline step 0.1284 ->
  // 0.1275 ms (two launches; A/B on one box).  32 threads x one float4 in 1,024-thread workgroups: 0.1324 (slower).
  // Up to 64 graphs (1,024-thread workgroups) the same: PROTEINS b64 116.6 -> 115.4 us.
  if (F == 128 && B <= 64) {
    const int nww = B <= 32 ? 8 : 16;
    const size_t ldsw = sizeof(float) * ((n_ghost ? (size_t)nww * F : 0) + 2 * nww + 4);
    if (B <= 32) {
      TSGNN_KNAME("slot_post_bwd<16,2,512>");
      slot_post_bwd<16, 2, 512><<<nmax, 512, ldsw, stream>>>(s, v, ldv, dxs, lddxs, dxs2, lddxs2, dout, ldo, arg, F / 4, relu, bn, mean, rstd, rinv, du, lddu, SlotBwdAlt{});
    } else {
      TSGNN_KNAME("slot_post_bwd<16,2,1024>");
      slot_post_bwd<16, 2, 1024><<<nmax, 1024, ldsw, stream>>>(s, v, ldv, dxs, lddxs, dxs2, lddxs2, dout, ldo, arg, F / 4, relu, bn, mean, rstd, rinv, du, lddu, SlotBwdAlt{});
    }
    TSGNN_CHECK_LAUNCH();
    return TSGNN_OK;
  }
  const int nw = B <= 32 ? 4 : (B <= 64 ? 8 : 16);
  const size_t lds = sizeof(float) * ((n_ghost ? (size_t)nw * F : 0) + 2 * nw + 4);
  TSGNN_SLOT_DISPATCH(slot_post_bwd, nmax, lds, (s, v, ldv, dxs, lddxs, dxs2, lddxs2, dout, ldo, arg, F / 4, relu, bn, mean, rstd, rinv, du, lddu, SlotBwdAlt{}));
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* Layer 0 of a stack: tsgnn_slot_post_bwd_f32 (relu = bn = 1, no dxs2) AND the weight / bias gradient slabs of tsgnn_linear_wgrad_f32 in
 * ONE launch, for the layer whose dU has no other consumer (its input needs no gradient): the rows of dU stay in LDS.  z [rows, K_in]:
 * the layer's aggregated input kept by the forward.  ws: nblocks slabs of (K_in + 1) * 128 floats (nblocks <= nslots workgroups walk the
 * slots; sum them with tsgnn_wgrad_reduce_multi_f32, nslab = nblocks).  B <= 32, F = 128, K_in <= 128, n_ghost = nmax. */
int tsgnn_slot_post_wgrad_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                              const float* v, int64_t ldv, const float* dxs, int64_t lddxs, const float* dout, int64_t ldo,
                              const int* arg, int F, int relu, int bn, const float* mean, const float* rstd, const float* rinv,
                              const float* z, int64_t ldz, int K_in, float* ws, int nblocks, tsgnn_stream_t stream) {
  if (!graph_ptr || !slot_count || !v || ((dout == nullptr) != (arg == nullptr)) || !rinv || !z || !ws || nmax <= 0 || nblocks <= 0 ||
      (bn && (!mean || !rstd)) || ldv < F || (ldv % 4) || (arg && (ldo % 4)) || (dxs && (lddxs % 4)) || K_in <= 0 || ldz < K_in)
    return TSGNN_EINVAL;
  if (B > 32 || F != 128 || K_in > 128 || (ldz % 4) || n_ghost != nmax || nblocks > nmax ||
      ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(ws)) & 15))
    return TSGNN_EUNSUPPORTED;
  SlotArgs s{graph_ptr, slot_count, B, nmax, n_real, n_ghost};
  const int mt = (K_in + 31) / 32;
  const size_t lds = sizeof(float) * (size_t)(SPW_AUX + 2 * 32 * 128 + 2 * 32 * 32 * mt);
#define TSGNN_SPW(M_) do { \
    static bool attr_##M_ = false; \
    if (!attr_##M_ && lds > 64 * 1024) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(slot_post_wgrad_kernel<M_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_##M_ = true; } \
    slot_post_wgrad_kernel<M_><<<(unsigned)nblocks, 256, lds, stream>>>(s, v, ldv, dxs, lddxs, dout, ldo, arg, relu, bn, mean, rstd, rinv, z, ldz, K_in, ws); } while (0)
  TSGNN_KNAME("slot_post_wgrad_kernel<%d>", mt);
  switch (mt) { case 1: TSGNN_SPW(1); break; case 2: TSGNN_SPW(2); break; case 3: TSGNN_SPW(3); break; default: TSGNN_SPW(4); break; }
#undef TSGNN_SPW
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* tsgnn_slot_post_bwd_f32 (without a readout gradient) for TWO stacks on the same batch and shapes in one launch (grid.y = 2);
 * dxs / dxs2 are given for both problems or for neither */
int tsgnn_slot_post_bwd_pair_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                                 const float* v0, const float* v1, int64_t ldv, const float* dxs0, const float* dxs1, int64_t lddxs,
                                 const float* dxs2_0, const float* dxs2_1, int64_t lddxs2, int F, int relu, int bn, const float* mean0,
                                 const float* mean1, const float* rstd0, const float* rstd1, const float* rinv0, const float* rinv1,
                                 float* du0, float* du1, int64_t lddu, tsgnn_stream_t stream) {
  if (!graph_ptr || !slot_count || !v0 || !v1 || !rinv0 || !rinv1 || !du0 || !du1 || nmax <= 0 || (n_ghost != 0 && n_ghost != nmax) ||
      (bn && (!mean0 || !mean1 || !rstd0 || !rstd1)) || ldv < F || lddu < F || (ldv % 4) || (lddu % 4) ||
      ((dxs0 == nullptr) != (dxs1 == nullptr)) || ((dxs2_0 == nullptr) != (dxs2_1 == nullptr)) || (dxs0 && (lddxs % 4)) ||
      (dxs2_0 && (lddxs2 % 4)))
    return TSGNN_EINVAL;
  if (!tsgnn_slot_fused_supported(B, F)) return TSGNN_EUNSUPPORTED;
  SlotArgs s{graph_ptr, slot_count, B, nmax, n_real, n_ghost};
  const int nw = B <= 32 ? 4 : (B <= 64 ? 8 : 16);
  const size_t lds = sizeof(float) * ((n_ghost ? (size_t)nw * F : 0) + 2 * nw + 4);
  const float* nof = nullptr;
  const int* noi = nullptr;
  TSGNN_SLOT_DISPATCH(slot_post_bwd, dim3((unsigned)nmax, 2), lds,
                      (s, v0, ldv, dxs0, lddxs, dxs2_0, lddxs2, nof, 0, noi, F / 4, relu, bn, mean0, rstd0, rinv0, du0, lddu,
                       SlotBwdAlt{v1, dxs1, dxs2_1, mean1, rstd1, rinv1, du1}));
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* tsgnn_slot_post_bwd_f32 for the LAST layer of a stack (relu = bn = 0, no dxs): du from the max-readout gradient, row-parallel.
 * row_graph[n_real] = graph of every real row (>= B: a padding row of a capacity-padded batch -> du = 0); rows
 * [0, n_real + n_ghost_rows) of du are written. */
int tsgnn_readout_l2_bwd_f32(const int* graph_ptr, const int* row_graph, int B, int64_t n_real, int n_ghost_rows, const float* v,
                             int64_t ldv, const float* dout, int64_t ldo, const int* arg, int F, const float* rinv, float* du,
                             int64_t lddu, tsgnn_stream_t stream) {
  if (!graph_ptr || !row_graph || !v || !dout || !arg || !rinv || !du || B <= 0 || n_real < 0 || n_ghost_rows < 0 || F <= 0 || ldv < F ||
      lddu < F)
    return TSGNN_EINVAL;
  if ((F % 4) || F > 128 || (ldv % 4) || (lddu % 4) || (ldo % 4) ||
      ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(arg)) & 15))
    return TSGNN_EUNSUPPORTED;
  const int64_t total = n_real + n_ghost_rows;
  if (total == 0) return TSGNN_OK;
  SlotArgs s{graph_ptr, nullptr, B, 0, n_real, n_ghost_rows};
  TSGNN_KNAME("readout_l2_bwd_rows");
  readout_l2_bwd_rows<<<(unsigned)ceil_div64(total, 8), 256, 0, stream>>>(s, row_graph, v, ldv, dout, ldo, arg, F / 4, rinv, du, lddu,
                                                                        n_ghost_rows);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* partial max of one layer into packed[B*F] (zeroed by the caller once per step for all layers) */
int tsgnn_readout_partial_f32(const int* graph_ptr, int B, int nmax, int64_t n_real, int n_ghost, const float* x, int64_t ldx, int F,
                              unsigned long long* packed, tsgnn_stream_t stream) {
  if (!graph_ptr || !x || !packed || B <= 0 || nmax <= 0 || F <= 0 || (F % 4) || F > 128 || ldx < F || (ldx % 4) ||
      (n_ghost != 0 && n_ghost != nmax))
    return TSGNN_EINVAL;
  SlotArgs s{graph_ptr, nullptr, B, nmax, n_real, n_ghost};
  dim3 grid((unsigned)((nmax + 63) / 64), (unsigned)B);
  TSGNN_KNAME("readout_partial4<32>");
  readout_partial4<32><<<grid, 256, 0, stream>>>(s, x, ldx, F / 4, packed);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_readout_decode_layers_f32(const unsigned long long* packed, int B, int L, int Fh, int Fl, float* out, int64_t ldo, int* arg,
                                    tsgnn_stream_t stream) {
  if (!packed || !out || !arg || B <= 0 || L <= 0 || Fh <= 0 || Fl <= 0) return TSGNN_EINVAL;
  const int64_t total = (int64_t)B * ((int64_t)(L - 1) * Fh + Fl);
  TSGNN_KNAME("readout_decode_layers");
  readout_decode_layers<<<(unsigned)ceil_div64(total, 256), 256, 0, stream>>>(packed, B, L, Fh, Fl, out, ldo, arg);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
