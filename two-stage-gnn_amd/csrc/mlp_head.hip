// Graph-level head of the SAGPool model (Code/sag/network.py:25-27,48-53):
//     x = relu(lin1(x)); x = dropout(x); x = relu(lin2(x)); x = log_softmax(lin3(x))
// forward and backward in 1 + 2 launches instead of ~25 library launches (3 addmm + 6 mm + bias reductions + the
// element-wise passes between them), for the small row counts of a graph-level head (B = graphs per batch).
// The dropout keep-mask (0 / 1 per element, scaled by keep_scale = 1/(1-p) here) is an input: the random stream stays the
// framework's own.
//   forward : one workgroup per graph row; weights streamed from L2 with 16-byte loads, a wave owns eight output
//             rows at a time and reduces their dot products with DPP.
//   backward: a rows kernel (dlogits -> dz2 -> dz1 -> dX once per row) and a weights kernel ([dW1 tile ‖ dW2 tile ‖ dW3] blocks
//             summing over the rows with 16-byte loads) — tsgnn_mlp3_bwd2_f32, 14 us; the single-launch variant
//             (tsgnn_mlp3_bwd_f32: four block kinds, each recomputing dlogits -> dz2 in LDS) is kept for comparison, 21 us.
//             Weight gradients are plain sums over the B rows in a fixed order (no atomics, reproducible).
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

constexpr int MH_WAVES = 16;   // forward: 1024 threads per row

// vs[j] = act(W[j,:] . xs + bias[j]) for j < E   (W [E, P] row-major, P % 4 == 0, xs in LDS)
template <bool RELU>
__device__ __forceinline__ void dense_row(const float* xs, int P, const float* __restrict__ w, const float* __restrict__ bias, int E,
                                          float* vs) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int P4 = P >> 2;
  for (int j0 = wid; j0 < E; j0 += 8 * MH_WAVES) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int k4 = lane; k4 < P4; k4 += 64) {
      const float4 x = *reinterpret_cast<const float4*>(xs + 4 * k4);
      float4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + MH_WAVES * u;
        wv[u] = ld4(w + (int64_t)(j < E ? j : 0) * P + 4 * k4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += (wv[u].x * x.x + wv[u].y * x.y) + (wv[u].z * x.z + wv[u].w * x.w);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + MH_WAVES * u;
      const float r = wave_sum(acc[u]);
      if (lane == 0 && j < E) {
        const float v = r + (bias ? bias[j] : 0.f);
        vs[j] = RELU ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

// Dropout mask made inside the launch (tsgnn_mlp3_fwd_drop_f32): element (b, j) of the hidden layer is kept when its 32 Philox bits
// are >= thresh (= p * 2^32), keyed on (seed + the DEVICE counter state[0]; b, j).  The counter advances once per launch — every
// block reads it first and draws a ticket (state[1]) when it is done; the block with the last ticket increments it — so a training
// step replayed from a hipGraph draws a new mask at every replay without a host-side generator (torch's graph-safe generator costs
// a bernoulli launch plus two fill launches per replay).  used[0] = the counter value of this launch (tests regenerate the mask).
struct Mlp3Drop {
  unsigned thresh; unsigned seed_lo, seed_hi;
  unsigned long long* state;             // [0] launches so far, [1] tickets of the running launch (zero between launches)
  unsigned long long* used;
};
__device__ __forceinline__ float mlp3_keep(unsigned thresh, unsigned klo, unsigned khi, unsigned b, unsigned j) {
  const uint4 r = philox4x32_10(make_uint4(b, j >> 2, 0u, 0x4D4C5033u), make_uint2(klo, khi));
  const unsigned v = (j & 3) == 0 ? r.x : (j & 3) == 1 ? r.y : (j & 3) == 2 ? r.z : r.w;
  return v < thresh ? 0.f : 1.f;
}

__global__ __launch_bounds__(64 * MH_WAVES) void mlp3_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w1,
                                                                 const float* __restrict__ b1, const float* __restrict__ keep,
                                                                 float keep_scale, const float* __restrict__ w2, const float* __restrict__ b2,
                                                                 const float* __restrict__ w3, const float* __restrict__ b3, int D0,
                                                                 int D1, int D2, int C, float* __restrict__ a1, float* __restrict__ a2,
                                                                 float* __restrict__ logp, Mlp3Drop drop) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                              // [D0]
  float* v1 = xs + D0;                           // [D1]
  float* v2 = v1 + ((D1 + 3) & ~3);              // [D2]
  float* lg = v2 + ((D2 + 3) & ~3);              // [C]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  unsigned long long ctr = 0ull;
  if (drop.state) ctr = __hip_atomic_load(drop.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read before anything else)
  for (int k = tid; k < D0; k += 64 * MH_WAVES) xs[k] = x[(int64_t)b * ldx + k];
  __syncthreads();
  dense_row<true>(xs, D0, w1, b1, D1, v1);
  __syncthreads();
  const unsigned klo = drop.seed_lo + (unsigned)(ctr & 0xffffffffull), khi = drop.seed_hi + (unsigned)(ctr >> 32);
  for (int j = tid; j < D1; j += 64 * MH_WAVES) {
    float m = 1.f;                                                                    // dropout after the ReLU (network.py:48-49)
    if (drop.state) m = mlp3_keep(drop.thresh, klo, khi, (unsigned)b, (unsigned)j) * keep_scale;
    else if (keep) m = keep[(int64_t)b * D1 + j] * keep_scale;
    const float v = v1[j] * m;
    v1[j] = v;
    a1[(int64_t)b * D1 + j] = v;
  }
  __syncthreads();
  dense_row<true>(v1, D1, w2, b2, D2, v2);
  __syncthreads();
  for (int j = tid; j < D2; j += 64 * MH_WAVES) a2[(int64_t)b * D2 + j] = v2[j];
  for (int c = wid; c < C; c += MH_WAVES) {
    float acc = 0.f;
    for (int j = lane; j < D2; j += 64) acc = fmaf(w3[(int64_t)c * D2 + j], v2[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) lg[c] = acc + (b3 ? b3[c] : 0.f);
  }
  __syncthreads();
  if (tid < C) {
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, lg[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(lg[c] - m);
    logp[(int64_t)b * C + tid] = lg[tid] - m - logf(s);
  }
  if (drop.state && tid == 0) {
    if (b == 0) drop.used[0] = ctr;
    // every block has read the counter before it draws its ticket: the last one advances it for the next launch
    asm volatile("" ::"v"((unsigned)ctr));
    const unsigned long long t = atomicAdd(drop.state + 1, 1ull);
    if (t == (unsigned long long)gridDim.x - 1ull) {
      atomicExch(drop.state + 1, 0ull);
      atomicAdd(drop.state, 1ull);
    }
  }
}

// the mask of a launch, regenerated (tests; [B, D1] of 0 / 1)
__global__ void mlp3_mask_kernel(unsigned thresh, unsigned klo, unsigned khi, int B, int D1, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * D1) return;
  out[i] = mlp3_keep(thresh, klo, khi, (unsigned)(i / D1), (unsigned)(i % D1));
}

struct Mlp3Bwd {
  const float* x; int64_t ldx;
  const float* w1; const float* w2; const float* w3;
  const float* a1; const float* a2; const float* logp; const float* dlogp;
  float keep_scale;                      // 1/(1-p) (1 without dropout): d a1 / d z1 = keep_scale * [a1 > 0]
  int B, D0, D1, D2, C;
  float* dw1; float* db1; float* dw2; float* db2; float* dw3; float* db3; float* dx; int64_t lddx;
  int nW1, nW2;                          // block kinds: [0, nW1) dW1 tiles, [nW1, nW1+nW2) dW2 tiles, one dW3 block, then dX rows
};

constexpr int MB_TILE = 2;               // weight rows per tile block
constexpr int MB_ROWS = 4;               // batch rows per dX block

// dz2[r, k] = [a2 > 0] * sum_c dlogits[r, c] W3[c, k],  dlogits = dlogp - softmax * sum_c dlogp   for rows [r0, r1)
__device__ __forceinline__ void mlp3_dz2(const Mlp3Bwd& p, int r0, int r1, float* dlg /*[rows][C]*/, float* w3s /*[C][D2]*/,
                                         float* dz2 /*[rows][D2+1]*/) {
  const int tid = threadIdx.x, nr = r1 - r0;
  for (int idx = tid; idx < p.C * p.D2; idx += 256) w3s[idx] = p.w3[idx];     // W3 once into LDS (it was re-read per element)
  for (int r = tid; r < nr; r += 256) {
    float s = 0.f;
    for (int c = 0; c < p.C; ++c) s += p.dlogp[(int64_t)(r0 + r) * p.C + c];
    for (int c = 0; c < p.C; ++c)
      dlg[r * p.C + c] = p.dlogp[(int64_t)(r0 + r) * p.C + c] - expf(p.logp[(int64_t)(r0 + r) * p.C + c]) * s;
  }
  __syncthreads();
  // the a2 loads of different elements are independent: issue them eight at a time (one load per trip was a chain of 32
  // L2 round trips in every weight-tile block)
  const int total = nr * p.D2;
  for (int base = tid; base < total; base += 256 * 8) {
    float av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 256 * u;
      av[u] = idx < total ? p.a2[(int64_t)r0 * p.D2 + idx] : 0.f;          // a2 rows are dense: [r0 + r][k] = r0 * D2 + idx
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 256 * u;
      if (idx < total) {
        const int r = idx / p.D2, k = idx - r * p.D2;
        float v = 0.f;
        for (int c = 0; c < p.C; ++c) v = fmaf(dlg[r * p.C + c], w3s[c * p.D2 + k], v);
        dz2[r * (p.D2 + 1) + k] = av[u] > 0.f ? v : 0.f;
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void mlp3_bwd_kernel(Mlp3Bwd p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  const int B = p.B, D0 = p.D0, D1 = p.D1, D2 = p.D2, C = p.C;
  if (bid < p.nW1 + p.nW2 + 1) {
    float* dlg = smem;                                   // [B][C]
    float* w3s = dlg + ((B * C + 3) & ~3);               // [C][D2]
    float* dz2 = w3s + ((C * D2 + 3) & ~3);              // [B][D2+1]
    float* t8 = dz2 + B * (D2 + 1);                      // [B][MB_TILE]   (dW1 blocks: the dz1 tile)
    mlp3_dz2(p, 0, B, dlg, w3s, dz2);
    if (bid < p.nW1) {
      // ---- dW1[o, :] for o in the tile: dz1[b, o] = gate * sum_k dz2[b, k] W2[k, o]
      const int o0 = bid * MB_TILE;
      float* w2t = t8 + B * MB_TILE;                     // [D2][MB_TILE]: this tile's columns of W2 (one cooperative fetch; the
      for (int idx = tid; idx < D2 * MB_TILE; idx += 256) {   // dependent per-k global loads of a first version cost 30 us)
        const int k = idx / MB_TILE, u = idx - k * MB_TILE;
        w2t[idx] = o0 + u < D1 ? p.w2[(int64_t)k * D1 + o0 + u] : 0.f;
      }
      __syncthreads();
      for (int idx = tid; idx < B * MB_TILE; idx += 256) {
        const int b = idx / MB_TILE, u = idx - b * MB_TILE;
        const int o = o0 + u;
        float v = 0.f;
        if (o < D1) {
#pragma unroll 8
          for (int k = 0; k < D2; ++k) v = fmaf(dz2[b * (D2 + 1) + k], w2t[k * MB_TILE + u], v);
          v = p.a1[(int64_t)b * D1 + o] > 0.f ? v * p.keep_scale : 0.f;
        }
        t8[b * MB_TILE + u] = v;
      }
      __syncthreads();
      for (int i = tid; i < D0; i += 256) {
        float acc[MB_TILE];
#pragma unroll
        for (int u = 0; u < MB_TILE; ++u) acc[u] = 0.f;
#pragma unroll 32
        for (int b = 0; b < B; ++b) {                      // independent coalesced loads, 32 in flight
          const float xv = p.x[(int64_t)b * p.ldx + i];
#pragma unroll
          for (int u = 0; u < MB_TILE; ++u) acc[u] = fmaf(t8[b * MB_TILE + u], xv, acc[u]);
        }
#pragma unroll
        for (int u = 0; u < MB_TILE; ++u)
          if (o0 + u < D1) p.dw1[(int64_t)(o0 + u) * D0 + i] = acc[u];
      }
      if (tid < MB_TILE && o0 + tid < D1) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += t8[b * MB_TILE + tid];
        p.db1[o0 + tid] = s;
      }
    } else if (bid < p.nW1 + p.nW2) {
      // ---- dW2[k, :] for k in the tile: sum_b dz2[b, k] a1[b, :]
      const int k0 = (bid - p.nW1) * MB_TILE;
      for (int o = tid; o < D1; o += 256) {
        float acc[MB_TILE];
#pragma unroll
        for (int u = 0; u < MB_TILE; ++u) acc[u] = 0.f;
#pragma unroll 32
        for (int b = 0; b < B; ++b) {
          const float av = p.a1[(int64_t)b * D1 + o];
#pragma unroll
          for (int u = 0; u < MB_TILE; ++u) acc[u] = fmaf(k0 + u < D2 ? dz2[b * (D2 + 1) + k0 + u] : 0.f, av, acc[u]);
        }
#pragma unroll
        for (int u = 0; u < MB_TILE; ++u)
          if (k0 + u < D2) p.dw2[(int64_t)(k0 + u) * D1 + o] = acc[u];
      }
      if (tid < MB_TILE && k0 + tid < D2) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dz2[b * (D2 + 1) + k0 + tid];
        p.db2[k0 + tid] = s;
      }
    } else {
      // ---- dW3[c, k] = sum_b dlogits[b, c] a2[b, k],  db3[c] = sum_b dlogits[b, c]
      for (int idx = tid; idx < C * D2; idx += 256) {
        const int c = idx / D2, k = idx - c * D2;
        float s = 0.f;
#pragma unroll 16
        for (int b = 0; b < B; ++b) s = fmaf(dlg[b * C + c], p.a2[(int64_t)b * D2 + k], s);
        p.dw3[idx] = s;
      }
      if (tid < C) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dlg[b * C + tid];
        p.db3[tid] = s;
      }
    }
    return;
  }
  // ---- dX rows: dz2 -> dz1 (all D1 columns) -> dx = dz1 W1
  if (p.dx == nullptr) return;
  const int r0 = (bid - (p.nW1 + p.nW2 + 1)) * MB_ROWS, r1 = min(B, r0 + MB_ROWS), nr = r1 - r0;
  float* dlg = smem;                                     // [4][C]
  float* w3s = dlg + ((MB_ROWS * C + 3) & ~3);           // [C][D2]
  float* dz2 = w3s + ((C * D2 + 3) & ~3);                // [4][D2+1]
  float* dz1 = dz2 + MB_ROWS * (D2 + 1);                 // [4][D1]
  mlp3_dz2(p, r0, r1, dlg, w3s, dz2);
  for (int idx = tid; idx < nr * D1; idx += 256) {
    const int r = idx / D1, o = idx - r * D1;
    float v = 0.f;
#pragma unroll 16
    for (int k = 0; k < D2; ++k) v = fmaf(dz2[r * (D2 + 1) + k], p.w2[(int64_t)k * D1 + o], v);       // coalesced over o
    dz1[r * D1 + o] = p.a1[(int64_t)(r0 + r) * D1 + o] > 0.f ? v * p.keep_scale : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < D0; i += 256) {
    float acc[MB_ROWS] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
    for (int o = 0; o < D1; ++o) {
      const float wv = p.w1[(int64_t)o * D0 + i];
#pragma unroll
      for (int r = 0; r < MB_ROWS; ++r) acc[r] = fmaf(r < nr ? dz1[r * D1 + o] : 0.f, wv, acc[r]);
    }
    for (int r = 0; r < nr; ++r) p.dx[(int64_t)(r0 + r) * p.lddx + i] = acc[r];
  }
}

// ---------------------------------------------------------------- two-launch backward
// rows kernel: block b runs its row's chain once — dlogits -> dz2 -> dz1 -> dx — and leaves dlogits / dz2 / dz1 in global
// memory; weights kernel: [dW1 tiles | dW2 tiles | dW3] blocks that only stage their operand tile and sum over the rows.
// The single launch above makes every weight tile recompute dz2 for all rows behind a chain of dependent phases at one wave per
// SIMD; two short kernels with the chain done once are faster (measured 21 -> ~9 us for B = 128, 256 -> 128 -> 64 -> 2).
constexpr int MR_T = 256;
// nll_label != NULL: dlogp is not an input — it is the gradient of F.nll_loss(logp, label) (mean; Code/sag/train.py) and
// dlogits = (softmax - onehot) / B is formed directly; the weights kernel's last block then also writes the loss value.
__global__ __launch_bounds__(MR_T) void mlp3_bwd_rows_kernel(Mlp3Bwd p, float* __restrict__ dlgg, float* __restrict__ dz2g,
                                                           float* __restrict__ dz1g, const int64_t* __restrict__ nll_label) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int D0 = p.D0, D1 = p.D1, D2 = p.D2, C = p.C;
  float* dlg = smem;                         // [C]
  float* dz2 = dlg + ((C + 3) & ~3);         // [D2]
  float* dz1 = dz2 + ((D2 + 3) & ~3);        // [D1]
  if (tid < C) {
    float v;
    if (nll_label != nullptr) {
      v = (expf(p.logp[(int64_t)b * C + tid]) - (tid == (int)nll_label[b] ? 1.f : 0.f)) / (float)p.B;
    } else {
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += p.dlogp[(int64_t)b * C + c];
      v = p.dlogp[(int64_t)b * C + tid] - expf(p.logp[(int64_t)b * C + tid]) * s;
    }
    dlg[tid] = v;
    dlgg[(int64_t)b * C + tid] = v;
  }
  __syncthreads();
  for (int k = tid; k < D2; k += MR_T) {
    float v = 0.f;
    for (int c = 0; c < C; ++c) v = fmaf(dlg[c], p.w3[(int64_t)c * D2 + k], v);
    v = p.a2[(int64_t)b * D2 + k] > 0.f ? v : 0.f;
    dz2[k] = v;
    dz2g[(int64_t)b * D2 + k] = v;
  }
  __syncthreads();
  for (int o = tid; o < D1; o += MR_T) {
    float v = 0.f;
#pragma unroll 16
    for (int k = 0; k < D2; ++k) v = fmaf(dz2[k], p.w2[(int64_t)k * D1 + o], v);          // coalesced over o
    v = p.a1[(int64_t)b * D1 + o] > 0.f ? v * p.keep_scale : 0.f;
    dz1[o] = v;
    dz1g[(int64_t)b * D1 + o] = v;
  }
  if (p.dx == nullptr) return;
  __syncthreads();
  for (int i = tid; i < D0; i += MR_T) {
    float acc = 0.f;
#pragma unroll 16
    for (int o = 0; o < D1; ++o) acc = fmaf(dz1[o], p.w1[(int64_t)o * D0 + i], acc);     // coalesced over i
    p.dx[(int64_t)b * p.lddx + i] = acc;
  }
}

constexpr int MW_TILE = 4;                   // weight rows per block
#ifndef TSGNN_MW_UNROLL
#define TSGNN_MW_UNROLL 8
#endif
#define MW_UNROLL TSGNN_MW_UNROLL
__global__ __launch_bounds__(256) void mlp3_bwd_weights_kernel(Mlp3Bwd p, const float* __restrict__ dlgg, const float* __restrict__ dz2g,
                                                              const float* __restrict__ dz1g, int nW1, int nW2,
                                                              const int64_t* __restrict__ nll_label, float* __restrict__ nll_loss,
                                                              float* __restrict__ normparts) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [B][MW_TILE] operand tile
  __shared__ float nred[4];
  const int tid = threadIdx.x, bid = blockIdx.x;
  float sq = 0.f;                                                    // this thread's share of the squares of what the block writes
  // normparts (nullable): block k leaves the sum of squares of the gradient entries it wrote (dW tile rows + their bias entries; the
  // last block: dW3 + db3) — the shares tsgnn_adam_from_partials_f32 adds up instead of a norm pass over the gradient
  auto leave_share = [&]() {
    if (!normparts) return;
    sq = wave_sum(sq);
    if ((tid & 63) == 0) nred[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) normparts[bid] = (nred[0] + nred[1]) + (nred[2] + nred[3]);
  };
  const int B = p.B, D0 = p.D0, D1 = p.D1, D2 = p.D2, C = p.C;
  if (bid < nW1 + nW2) {
    const bool first = bid < nW1;
    const int r0 = (first ? bid : bid - nW1) * MW_TILE;               // first row of the tile (o for dW1, k for dW2)
    const int Dr = first ? D1 : D2, Dc = first ? D0 : D1;
    const float* dz = first ? dz1g : dz2g;                             // [B, Dr]
    const float* act = first ? p.x : p.a1;                             // [B, Dc] (row stride ldx / D1)
    const int64_t lda = first ? p.ldx : (int64_t)D1;
    for (int idx = tid; idx < B * MW_TILE; idx += 256) {
      const int b = idx / MW_TILE, u = idx - b * MW_TILE;
      smem[idx] = r0 + u < Dr ? dz[(int64_t)b * Dr + r0 + u] : 0.f;
    }
    __syncthreads();
    float* dw = first ? p.dw1 : p.dw2;
    // 64 float4 columns x 4 row lanes: 16-byte loads of the activation rows, B / 4 of them per thread and all independent
    // (one thread per column with 4-byte loads issued 128 dependent-latency-bound loads per thread: 11 us), partial sums of
    // the four row lanes through LDS, added in lane order
    float* part = smem + ((B * MW_TILE + 3) & ~3);                     // [4][MW_TILE][Dc]
    const int c4 = tid & 63, bl = tid >> 6, Dc4 = Dc >> 2;
    for (int cb = 0; cb < Dc4; cb += 64) {
      const int cc = cb + c4;
      float4 acc[MW_TILE];
#pragma unroll
      for (int u = 0; u < MW_TILE; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cc < Dc4) {
#pragma unroll MW_UNROLL
        for (int b = bl; b < B; b += 4) {                               // B / 4 independent 16-byte loads, MW_UNROLL in flight
          const float4 v = ld4(act + (int64_t)b * lda + 4 * cc);
#pragma unroll
          for (int u = 0; u < MW_TILE; ++u) {
            const float d = smem[b * MW_TILE + u];
            acc[u].x = fmaf(d, v.x, acc[u].x); acc[u].y = fmaf(d, v.y, acc[u].y);
            acc[u].z = fmaf(d, v.z, acc[u].z); acc[u].w = fmaf(d, v.w, acc[u].w);
          }
        }
#pragma unroll
        for (int u = 0; u < MW_TILE; ++u) *reinterpret_cast<float4*>(part + ((bl * MW_TILE + u) * Dc) + 4 * cc) = acc[u];
      }
      __syncthreads();
      for (int idx = tid; idx < MW_TILE * min(256, Dc - 4 * cb); idx += 256) {
        const int u = idx / min(256, Dc - 4 * cb), c = 4 * cb + idx % min(256, Dc - 4 * cb);
        const float v = (part[(0 * MW_TILE + u) * Dc + c] + part[(1 * MW_TILE + u) * Dc + c]) +
                        (part[(2 * MW_TILE + u) * Dc + c] + part[(3 * MW_TILE + u) * Dc + c]);
        if (r0 + u < Dr) { dw[(int64_t)(r0 + u) * Dc + c] = v; sq = fmaf(v, v, sq); }
      }
      __syncthreads();
    }
    if (tid < MW_TILE && r0 + tid < Dr) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += smem[b * MW_TILE + tid];
      (first ? p.db1 : p.db2)[r0 + tid] = s;
      sq = fmaf(s, s, sq);
    }
    leave_share();
    return;
  }
  // dW3[c, k] = sum_b dlogits[b, c] a2[b, k]: the row loop is spread over the block (one thread per output walked all B rows
  // in a chain of dependent loads and was the longest block of the launch)
  if ((D2 & 3) == 0 && (D2 >> 2) <= 256) {
    const int nc4 = D2 >> 2, nbl = 256 / nc4;                          // float4 columns x row lanes
    const int k4 = tid % nc4, bl = tid / nc4;
    float* part = smem;                                                // [nbl][C][D2]  (the operand tile region is unused here)
    float4 acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bl < nbl) {
#pragma unroll 4
      for (int b = bl; b < B; b += nbl) {
        const float4 v = ld4(p.a2 + (int64_t)b * D2 + 4 * k4);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (c < C) {
            const float d = dlgg[(int64_t)b * C + c];
            acc[c].x = fmaf(d, v.x, acc[c].x); acc[c].y = fmaf(d, v.y, acc[c].y);
            acc[c].z = fmaf(d, v.z, acc[c].z); acc[c].w = fmaf(d, v.w, acc[c].w);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < C) *reinterpret_cast<float4*>(part + ((size_t)bl * C + c) * D2 + 4 * k4) = acc[c];
    }
    __syncthreads();
    for (int idx = tid; idx < C * D2; idx += 256) {
      float sum = 0.f;
      for (int q = 0; q < nbl; ++q) sum += part[(size_t)q * C * D2 + idx];
      p.dw3[idx] = sum;
      sq = fmaf(sum, sum, sq);
    }
  } else {
    for (int idx = tid; idx < C * D2; idx += 256) {
      const int c = idx / D2, k = idx - c * D2;
      float sum = 0.f;
#pragma unroll 16
      for (int b = 0; b < B; ++b) sum = fmaf(dlgg[(int64_t)b * C + c], p.a2[(int64_t)b * D2 + k], sum);
      p.dw3[idx] = sum;
      sq = fmaf(sum, sum, sq);
    }
  }
  // db3: 16 row lanes per class through LDS (one thread per class walking all rows from global memory was a chain of B loads)
  __syncthreads();
  {
    float* pr = smem;                                                  // [16][C]
    const int c = tid % 16, bl = tid / 16;
    float s = 0.f;
    if (c < C)
      for (int b = bl; b < B; b += 16) s += dlgg[(int64_t)b * C + c];
    if (c < C) pr[bl * C + c] = s;
    __syncthreads();
    if (tid < C) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += pr[q * C + tid];
      p.db3[tid] = t;
      sq = fmaf(t, t, sq);
    }
  }
  leave_share();
  if (nll_label != nullptr) {                                          // loss = -mean_b logp[b, label_b], 256 row lanes then in order
    __syncthreads();
    float s = 0.f;
    for (int b = tid; b < B; b += 256) s -= p.logp[(int64_t)b * C + (int)nll_label[b]];
    smem[tid] = s;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int q = 0; q < 256; ++q) t += smem[q];
      nll_loss[0] = t / (float)B;
    }
  }
}

inline size_t bwd_lds_bytes(int B, int D1, int D2, int C) {
  const size_t w3 = (size_t)((C * D2 + 3) & ~3);
  const size_t wblk = w3 + (size_t)((B * C + 3) & ~3) + (size_t)B * (D2 + 1) + (size_t)B * MB_TILE + (size_t)D2 * MB_TILE;
  const size_t rblk = w3 + (size_t)((MB_ROWS * C + 3) & ~3) + (size_t)MB_ROWS * (D2 + 1) + (size_t)MB_ROWS * D1;
  return sizeof(float) * (wblk > rblk ? wblk : rblk);
}

}  // namespace

extern "C" {

/* 1 when the fused head accepts the shape: D0 % 4 == 0 (16-byte weight rows), D1 % 4 == 0, C <= 16, and the backward's
 * per-block copy of dz2[B, D2] fits LDS */
int tsgnn_mlp3_supported(int B, int D0, int D1, int D2, int C) {
  if (B <= 0 || D0 <= 0 || D1 <= 0 || D2 <= 0 || C <= 0 || C > 16 || D0 % 4 || D1 % 4 || D0 > 4096 || D1 > 2048 || D2 > 1024) return 0;
  // the two-launch backward stages a [B][4] operand tile per weight block (the single-launch variant needs dz2[B, D2] in LDS and
  // reports TSGNN_EUNSUPPORTED itself beyond that)
  return (size_t)B * 16 + 16 * (size_t)(D0 > D1 ? D0 : D1) * 4 + 4096 * (size_t)C <= 128 * 1024 ? 1 : 0;
}

int tsgnn_mlp3_fwd_f32(const float* x, int64_t ldx, const float* w1, const float* b1, const float* keep, float keep_scale, const float* w2,
                       const float* b2, const float* w3, const float* b3, int B, int D0, int D1, int D2, int C, float* a1, float* a2,
                       float* logp, tsgnn_stream_t stream) {
  if (!x || !w1 || !w2 || !w3 || !a1 || !a2 || !logp || ldx < D0) return TSGNN_EINVAL;
  if (!tsgnn_mlp3_supported(B, D0, D1, D2, C) || (reinterpret_cast<uintptr_t>(w1) & 15) || (reinterpret_cast<uintptr_t>(w2) & 15))
    return TSGNN_EUNSUPPORTED;
  const size_t lds = sizeof(float) * ((size_t)D0 + ((D1 + 3) & ~3) + ((D2 + 3) & ~3) + C);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp3_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  mlp3_fwd_kernel<<<(unsigned)B, 64 * MH_WAVES, lds, stream>>>(x, ldx, w1, b1, keep, keep_scale, w2, b2, w3, b3, D0, D1, D2, C, a1, a2, logp,
                                                                Mlp3Drop{});
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

static unsigned mlp3_thresh(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}

/* tsgnn_mlp3_fwd_f32 with the dropout mask (probability p of dropping, survivors scaled by 1 / (1 - p)) made INSIDE the launch:
 * Philox4x32-10 keyed on (seed + the device counter state[0]; graph, hidden unit).  state: two uint64 words, zero-initialised once:
 * [0] counts the launches (advanced by the launch itself: a step replayed from a hipGraph draws a new mask every replay), [1] is
 * its ticket word (zero between launches).  used[0] (nullable) receives the counter value this launch keyed its mask with. */
int tsgnn_mlp3_fwd_drop_f32(const float* x, int64_t ldx, const float* w1, const float* b1, float p, uint64_t seed,
                            unsigned long long* state, unsigned long long* used, const float* w2, const float* b2, const float* w3,
                            const float* b3, int B, int D0, int D1, int D2, int C, float* a1, float* a2, float* logp,
                            tsgnn_stream_t stream) {
  if (!x || !w1 || !w2 || !w3 || !a1 || !a2 || !logp || !state || !used || ldx < D0 || !(p >= 0.f) || !(p < 1.f)) return TSGNN_EINVAL;
  if (!tsgnn_mlp3_supported(B, D0, D1, D2, C) || (reinterpret_cast<uintptr_t>(w1) & 15) || (reinterpret_cast<uintptr_t>(w2) & 15))
    return TSGNN_EUNSUPPORTED;
  const size_t lds = sizeof(float) * ((size_t)D0 + ((D1 + 3) & ~3) + ((D2 + 3) & ~3) + C);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp3_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const Mlp3Drop d{mlp3_thresh(p), (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), state, used};
  mlp3_fwd_kernel<<<(unsigned)B, 64 * MH_WAVES, lds, stream>>>(x, ldx, w1, b1, nullptr, 1.0f / (1.0f - p), w2, b2, w3, b3, D0, D1, D2, C, a1, a2,
                                                                logp, d);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* the 0 / 1 keep mask [B, D1] that tsgnn_mlp3_fwd_drop_f32 used in the launch whose counter value was `counter` (tests, oracles) */
int tsgnn_mlp3_dropout_mask_f32(float p, uint64_t seed, uint64_t counter, int B, int D1, float* out, tsgnn_stream_t stream) {
  if (!out || B <= 0 || D1 <= 0 || !(p >= 0.f) || !(p < 1.f)) return TSGNN_EINVAL;
  const unsigned klo = (unsigned)(seed & 0xffffffffull) + (unsigned)(counter & 0xffffffffull);
  const unsigned khi = (unsigned)(seed >> 32) + (unsigned)(counter >> 32);
  mlp3_mask_kernel<<<(unsigned)ceil_div64((int64_t)B * D1, 256), 256, 0, stream>>>(mlp3_thresh(p), klo, khi, B, D1, out);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_mlp3_bwd_f32(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                       const float* a2, const float* logp, const float* dlogp, float keep_scale, int B, int D0, int D1, int D2, int C,
                       float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx, int64_t lddx,
                       tsgnn_stream_t stream) {
  if (!x || !w1 || !w2 || !w3 || !a1 || !a2 || !logp || !dlogp || !dw1 || !db1 || !dw2 || !db2 || !dw3 || !db3 || ldx < D0 ||
      (dx && lddx < D0))
    return TSGNN_EINVAL;
  if (!tsgnn_mlp3_supported(B, D0, D1, D2, C) || bwd_lds_bytes(B, D1, D2, C) > 96 * 1024) return TSGNN_EUNSUPPORTED;
  Mlp3Bwd p{x, ldx, w1, w2, w3, a1, a2, logp, dlogp, keep_scale, B, D0, D1, D2, C, dw1, db1, dw2, db2, dw3, db3, dx, lddx,
            (D1 + MB_TILE - 1) / MB_TILE, (D2 + MB_TILE - 1) / MB_TILE};
  const unsigned nblk = (unsigned)(p.nW1 + p.nW2 + 1 + (dx ? (B + MB_ROWS - 1) / MB_ROWS : 0));
  const size_t lds = bwd_lds_bytes(B, D1, D2, C);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp3_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  mlp3_bwd_kernel<<<nblk, 256, lds, stream>>>(p);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* the same backward as two short launches (rows: the chain once per row; weights: tiles summing over the rows);
 * ws: B * (C + D2 + D1) floats */
static int mlp3_bwd2_launch(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                            const float* a2, const float* logp, const float* dlogp, float keep_scale, int B, int D0, int D1, int D2, int C,
                            float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx, int64_t lddx, float* ws,
                            const int64_t* nll_label, float* nll_loss, float* normparts, tsgnn_stream_t stream) {
  if (!x || !w1 || !w2 || !w3 || !a1 || !a2 || !logp || (!dlogp && !nll_label) || !dw1 || !db1 || !dw2 || !db2 || !dw3 || !db3 || !ws ||
      ldx < D0 || (dx && lddx < D0) || (nll_label && !nll_loss))
    return TSGNN_EINVAL;
  if (!tsgnn_mlp3_supported(B, D0, D1, D2, C)) return TSGNN_EUNSUPPORTED;
  Mlp3Bwd p{x, ldx, w1, w2, w3, a1, a2, logp, dlogp, keep_scale, B, D0, D1, D2, C, dw1, db1, dw2, db2, dw3, db3, dx, lddx, 0, 0};
  float* dlgg = ws;
  float* dz2g = dlgg + (size_t)B * C;
  float* dz1g = dz2g + (size_t)B * D2;
  const size_t lds_r = sizeof(float) * (size_t)(((C + 3) & ~3) + ((D2 + 3) & ~3) + D1);
  mlp3_bwd_rows_kernel<<<(unsigned)B, MR_T, lds_r, stream>>>(p, dlgg, dz2g, dz1g, nll_label);
  const int nW1 = (D1 + MW_TILE - 1) / MW_TILE, nW2 = (D2 + MW_TILE - 1) / MW_TILE;
  size_t lds_w = sizeof(float) * ((size_t)((B * MW_TILE + 3) & ~3) + 4 * (size_t)MW_TILE * (D0 > D1 ? D0 : D1));
  if (lds_w < sizeof(float) * 1024 * (size_t)C) lds_w = sizeof(float) * 1024 * (size_t)C;       // dW3 block: [256 / (D2/4)][C][D2] partials
  if (lds_w > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp3_bwd_weights_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
  mlp3_bwd_weights_kernel<<<(unsigned)(nW1 + nW2 + 1), 256, lds_w, stream>>>(p, dlgg, dz2g, dz1g, nW1, nW2, nll_label, nll_loss, normparts);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_mlp3_bwd2_f32(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                        const float* a2, const float* logp, const float* dlogp, float keep_scale, int B, int D0, int D1, int D2, int C,
                        float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx, int64_t lddx, float* ws,
                        tsgnn_stream_t stream) {
  return mlp3_bwd2_launch(x, ldx, w1, w2, w3, a1, a2, logp, dlogp, keep_scale, B, D0, D1, D2, C, dw1, db1, dw2, db2, dw3, db3, dx, lddx,
                          ws, nullptr, nullptr, nullptr, stream);
}

/* the same with F.nll_loss(logp, label) (mean) folded in: its gradient is formed inside the rows kernel and the loss value is
 * written to loss[0] by the weights kernel */
int tsgnn_mlp3_bwd2_nll_f32(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                            const float* a2, const float* logp, const int64_t* label, float* loss, float keep_scale, int B, int D0,
                            int D1, int D2, int C, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx,
                            int64_t lddx, float* ws, tsgnn_stream_t stream) {
  if (!label || !loss) return TSGNN_EINVAL;
  return mlp3_bwd2_launch(x, ldx, w1, w2, w3, a1, a2, logp, nullptr, keep_scale, B, D0, D1, D2, C, dw1, db1, dw2, db2, dw3, db3, dx, lddx,
                          ws, label, loss, nullptr, stream);
}

/* blocks of the weights launch of tsgnn_mlp3_bwd2*_f32 = entries of normparts */
int tsgnn_mlp3_bwd2_norm_blocks(int D1, int D2) { return (D1 + MW_TILE - 1) / MW_TILE + (D2 + MW_TILE - 1) / MW_TILE + 1; }

/* tsgnn_mlp3_bwd2_f32 / _nll_f32 (label != NULL: the nll form, dlogp ignored) that also leaves, per block of its weights launch, the
 * sum of squares of the gradient entries that block wrote (normparts[tsgnn_mlp3_bwd2_norm_blocks(D1, D2)]): the head's shares of
 * |grad|^2 for tsgnn_adam_from_partials_f32 when all six gradients are written straight into the optimiser's flat bucket */
int tsgnn_mlp3_bwd2_np_f32(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                           const float* a2, const float* logp, const float* dlogp, const int64_t* label, float* loss, float keep_scale, int B,
                           int D0, int D1, int D2, int C, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx,
                           int64_t lddx, float* ws, float* normparts, tsgnn_stream_t stream) {
  if (!normparts || (label && !loss)) return TSGNN_EINVAL;
  return mlp3_bwd2_launch(x, ldx, w1, w2, w3, a1, a2, logp, label ? nullptr : dlogp, keep_scale, B, D0, D1, D2, C, dw1, db1, dw2, db2, dw3, db3,
                          dx, lddx, ws, label, loss, normparts, stream);
}

}  // extern "C"
