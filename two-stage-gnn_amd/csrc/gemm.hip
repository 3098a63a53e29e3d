// fp32 MFMA GEMM building block (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// Used for: the `.W` transform of GraphConv (encoders.py:36) and its backward GEMMs,
// the GAT projection h = xW (encoders_GAT.py:32), DiffPool's S^T.Z / S^T.(A.S) contractions
// (encoders.py:374-375) as strided / ragged batched products, and their backward.
//
//   C[z] (+)= alpha * op(A[z]) . op(B[z])      element (m,k) of op(A) at A + m*sam + k*sak
//
// so transposes are just strides.  Batches are either strided (stride_*) or ragged: with seg_ptr,
// batch z covers rows [seg_ptr[z], seg_ptr[z+1]) of the ragged dimension (K: both operands shift
// along k; M: A and C shift along m).  Split-K (grid.y * ksplit) writes partial slabs that
// tsgnn_splitk_reduce_f32 sums in a fixed order (bitwise reproducible; no float atomics).
//
// Tile: 64x64 per 256-thread block (4 waves, 2x2, one 32x32 MFMA tile each), K chunk 32, operands
// staged through LDS with +1 padding (conflict-free ds_read_b32 for both fragment shapes).
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include "../../include/tsgnn.h"

#include "tn_rows_body.h"
#include "wgrad_blocks_body.h"
#include "du_reduce_body.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDA_S = BK + 1;   // As[m][k]
constexpr int LDB_S = BN + 1;   // Bs[k][n]

struct GemmArgs {
  const float* A; const float* B; float* C;
  int64_t sam, sak, sbk, sbn, scm, scn;
  int64_t stride_a, stride_b, stride_c;   // per batch (elements)
  int M, N, K;
  const int* seg_ptr;                     // nullable
  int ragged;                             // 0 none, 1 = K ragged, 2 = M ragged
  int ksplit;                             // >= 1
  int64_t slab_stride;                    // elements between split-K slabs (C = slab base when ksplit > 1)
  float alpha;
  int accumulate;                         // C += (only when ksplit == 1)
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float As[BM * LDA_S];
  __shared__ float Bs[BK * LDB_S];
  const int z = blockIdx.z;
  const int tile_n = blockIdx.x, tile_m = blockIdx.y / g.ksplit, split = blockIdx.y % g.ksplit;
  const float* A = g.A + (int64_t)z * g.stride_a;
  const float* B = g.B + (int64_t)z * g.stride_b;
  float* C = g.C + (int64_t)z * g.stride_c + (int64_t)split * g.slab_stride;
  int M = g.M, K = g.K;
  if (g.seg_ptr) {
    const int s0 = g.seg_ptr[z], s1 = g.seg_ptr[z + 1];
    if (g.ragged == 1) { A += (int64_t)s0 * g.sak; B += (int64_t)s0 * g.sbk; K = s1 - s0; }
    else { A += (int64_t)s0 * g.sam; C += (int64_t)s0 * g.scm; M = s1 - s0; }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= M) return;                                   // ragged-M batches shorter than the grid
  // K range of this split (multiples of BK)
  const int kchunks = (K + BK - 1) / BK;
  const int per = (kchunks + g.ksplit - 1) / g.ksplit;
  const int kbeg = split * per * BK;
  const int kend = min(K, (split + 1) * per * BK);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;                 // wave tile (wr*32.., wc*32..)
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const bool a_k_fast = (g.sak == 1);                    // which index is contiguous in memory
  const bool b_n_fast = (g.sbn == 1);
  // register-staged software pipeline: the next K chunk's global loads are issued before the MFMAs of the current one
  constexpr int AV = (BM * BK) / 256, BV = (BK * BN) / 256;
  float ra[AV], rb[BV];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < AV; ++it) {
      const int idx = it * 256 + tid;
      int m, k;
      if (a_k_fast) { m = idx / BK; k = idx % BK; } else { k = idx / BM; m = idx % BM; }
      const int gm = m0 + m, gk = k0 + k;
      ra[it] = (gm < M && gk < kend) ? A[(int64_t)gm * g.sam + (int64_t)gk * g.sak] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < BV; ++it) {
      const int idx = it * 256 + tid;
      int k, n;
      if (b_n_fast) { k = idx / BN; n = idx % BN; } else { n = idx / BK; k = idx % BK; }
      const int gk = k0 + k, gn = n0 + n;
      rb[it] = (gk < kend && gn < g.N) ? B[(int64_t)gk * g.sbk + (int64_t)gn * g.sbn] : 0.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < AV; ++it) {
      const int idx = it * 256 + tid;
      int m, k;
      if (a_k_fast) { m = idx / BK; k = idx % BK; } else { k = idx / BM; m = idx % BM; }
      As[m * LDA_S + k] = ra[it];
    }
#pragma unroll
    for (int it = 0; it < BV; ++it) {
      const int idx = it * 256 + tid;
      int k, n;
      if (b_n_fast) { k = idx / BN; n = idx % BN; } else { n = idx / BK; k = idx % BK; }
      Bs[k * LDB_S + n] = rb[it];
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    commit();
    __syncthreads();
    const int i = lane & 31, h = lane >> 5;
    const float* ap = As + (wr * 32 + i) * LDA_S + h;
    const float* bp = Bs + h * LDB_S + wc * 32 + i;
    float af[BK / 2], bf[BK / 2];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) { af[kk / 2] = ap[kk]; bf[kk / 2] = bp[kk * LDB_S]; }
    __builtin_amdgcn_sched_barrier(0);                   // fragment reads stay ahead of the MFMA chain
    if (k0 + BK < kend) fetch(k0 + BK);                  // in flight under the MFMAs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < BK / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc, 0, 0, 0);
    __syncthreads();
  }
  // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int cn = n0 + wc * 32 + (lane & 31);
  if (cn < g.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cm = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (cm < M) {
        float* cp = C + (int64_t)cm * g.scm + (int64_t)cn * g.scn;
        const float v = g.alpha * acc[r];
        *cp = (g.accumulate && g.ksplit == 1) ? (*cp + v) : v;
      }
    }
  }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int nsplit, int64_t slab_stride, int64_t n,
                                     float* __restrict__ out, int accumulate) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;
  int k = 0;
  for (; k + 8 <= nsplit; k += 8) {                      // eight slab loads in flight (one at a time, 64 slabs were a 16 us chain),
    float t[8];                                          // added in a fixed order
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = slabs[(int64_t)(k + u) * slab_stride + i];
    s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  for (; k < nsplit; ++k) s += slabs[(int64_t)k * slab_stride + i];
  out[i] = s;
}

// column sums of a row-major matrix (bias gradients): two-level, deterministic.
// grid (ceil(F/64), nchunk); block 256 = 4 row-lanes x 64 features
__global__ __launch_bounds__(256) void colsum_partial(const float* __restrict__ x, int64_t ld, int64_t rows, int F,
                                                      int64_t rows_per_chunk, float* __restrict__ part) {
  __shared__ float lds[4][64];
  const int f = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (f < F) {
    int64_t r = r0 + rl;
    for (; r + 28 < r1; r += 32) {                       // eight independent loads in flight, fixed summation order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = x[(r + 4 * u) * ld + f];
      s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; r < r1; r += 4) s += x[r * ld + f];
  }
  lds[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && f < F) part[(int64_t)blockIdx.y * F + f] = lds[0][threadIdx.x] + lds[1][threadIdx.x] + lds[2][threadIdx.x] + lds[3][threadIdx.x];
}


// weight + bias gradient of a layer with a NARROW input (K_in <= 4: the one-column degree / constant feature of the IMDB sets):
// dW[k, f] = sum_r z[r, k] du[r, f] and db[f] = sum_r du[r, f] from ONE pass over du — the column-sum kernel with K_in extra
// weighted accumulators (an MFMA tile would be 1/32 full; the split-K product + its reduction + the column sums were 19 us
// for a [1 x 128] result).  grid (ceil(F/64), nchunk); partial layout [chunk][K_in + 1][F], summed by splitk_reduce_kernel.
constexpr int WN_KMAX = 4;
__global__ __launch_bounds__(256) void wgrad_narrow_partial(const float* __restrict__ z, int64_t ldz, const float* __restrict__ du,
                                                            int64_t lddu, int64_t rows, int K_in, int F, int64_t rows_per_chunk,
                                                            float* __restrict__ part) {
  __shared__ float lds[WN_KMAX + 1][4][64];
  const int fl = threadIdx.x & 63, f = blockIdx.x * 64 + fl;
  const int rl = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float acc[WN_KMAX + 1];
#pragma unroll
  for (int k = 0; k <= WN_KMAX; ++k) acc[k] = 0.f;
  if (f < F) {
    for (int64_t rb = r0 + rl; rb < r1; rb += 32) {        // eight rows in flight per thread, fixed summation order
      float t[8], zz[8][WN_KMAX];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = rb + 4 * u;
        const bool ok = r < r1;
        t[u] = ok ? du[r * lddu + f] : 0.f;
#pragma unroll
        for (int k = 0; k < WN_KMAX; ++k) zz[u][k] = (ok && k < K_in) ? z[r * ldz + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int k = 0; k < WN_KMAX; ++k) acc[k] = fmaf(zz[u][k], t[u], acc[k]);
        acc[WN_KMAX] += t[u];
      }
    }
  }
#pragma unroll
  for (int k = 0; k <= WN_KMAX; ++k) lds[k][rl][fl] = acc[k];
  __syncthreads();
  if (rl == 0 && f < F) {
    float* o = part + (int64_t)blockIdx.y * (K_in + 1) * F;
    for (int k = 0; k < K_in; ++k) o[(int64_t)k * F + f] = (lds[k][0][fl] + lds[k][1][fl]) + (lds[k][2][fl] + lds[k][3][fl]);
    o[(int64_t)K_in * F + f] = (lds[WN_KMAX][0][fl] + lds[WN_KMAX][1][fl]) + (lds[WN_KMAX][2][fl] + lds[WN_KMAX][3][fl]);
  }
}

template <int MT, int NTt, int NY>
__global__ __launch_bounds__(256) void gemm_tn_rows_kernel(TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) float tn_smem[];   // two stages of [Z chunk | dU chunk]
  tn_rows_body<MT, NTt, NY>(g, tn_smem, blockIdx.x, blockIdx.y, gridDim.x);
}

// out[e] = sum_s slabs[s][e]; e < K_in*N -> dW, else -> db.  64 outputs x 4 slab groups per block.
__global__ __launch_bounds__(256) void tn_rows_reduce(const float* __restrict__ slabs, int nslab, int64_t per_slab, int64_t n_w,
                                                      float* __restrict__ dw, float* __restrict__ db,
                                                      const int* __restrict__ seg_slab_ptr) {
  __shared__ float lds[4][64];
  const int e_l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + e_l;
  // blockIdx.y = segment (graph): sums only that segment's slabs into dw[seg]
  const int sb = seg_slab_ptr ? seg_slab_ptr[blockIdx.y] : 0;
  const int se = seg_slab_ptr ? seg_slab_ptr[blockIdx.y + 1] : nslab;
  dw += (int64_t)blockIdx.y * n_w;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < per_slab) {
    const int per = (se - sb + 3) / 4;
    const int s0 = sb + grp * per, s1 = min(se, s0 + per);
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      a0 += slabs[(int64_t)s * per_slab + e];
      a1 += slabs[(int64_t)(s + 1) * per_slab + e];
      a2 += slabs[(int64_t)(s + 2) * per_slab + e];
      a3 += slabs[(int64_t)(s + 3) * per_slab + e];
    }
    for (; s < s1; ++s) a0 += slabs[(int64_t)s * per_slab + e];
  }
  lds[grp][e_l] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && e < per_slab) {
    const float v = (lds[0][e_l] + lds[1][e_l]) + (lds[2][e_l] + lds[3][e_l]);
    if (e < n_w) dw[e] = v;
    else if (db) db[e - n_w] = v;
  }
}


// tn_rows_reduce with one passenger: the last block adds up the per-graph partial rows of the SAGPool score-layer gradients
// (du_reduce_body) — both reductions wait for the same producer launch, so they share one launch (4 us per pooled level)
__global__ __launch_bounds__(256) void tn_rows_reduce_du(const float* __restrict__ slabs, int nslab, int64_t per_slab, int64_t n_w,
                                                         float* __restrict__ dw, float* __restrict__ db, float* __restrict__ part, int nb,
                                                         int F_du, float* __restrict__ dws, float* __restrict__ dbs) {
  __shared__ float4 s_part[256];
  if (blockIdx.x == gridDim.x - 1) {
    du_reduce_body(part, nb, F_du, dws, dbs, nb, 1, 0, s_part);
    return;
  }
  float (*lds)[64] = reinterpret_cast<float (*)[64]>(s_part);      // [4][64]
  const int e_l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + e_l;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < per_slab) {                                                 // the arithmetic (and order) of tn_rows_reduce
    const int per = (nslab + 3) / 4;
    const int s0 = grp * per, s1 = min(nslab, s0 + per);
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      a0 += slabs[(int64_t)s * per_slab + e];
      a1 += slabs[(int64_t)(s + 1) * per_slab + e];
      a2 += slabs[(int64_t)(s + 2) * per_slab + e];
      a3 += slabs[(int64_t)(s + 3) * per_slab + e];
    }
    for (; s < s1; ++s) a0 += slabs[(int64_t)s * per_slab + e];
  }
  lds[grp][e_l] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && e < per_slab) {
    const float v = (lds[0][e_l] + lds[1][e_l]) + (lds[2][e_l] + lds[3][e_l]);
    if (e < n_w) dw[e] = v;
    else if (db) db[e - n_w] = v;
  }
}

struct ReduceSet { const float* slabs; int nslab; int64_t per_slab; int64_t n_w; float* dw; float* db; int64_t first_block; };
constexpr int RM_SETS = 8;
struct ReduceMulti { ReduceSet s[RM_SETS]; int n; float* normparts; float* step_state; };
// several independent slab sets (the layers of one backward pass) reduced by ONE launch
__global__ __launch_bounds__(256) void tn_rows_reduce_multi(ReduceMulti m) {
  __shared__ float lds[4][64];
  int k = 0;
#pragma unroll
  for (int t = 1; t < RM_SETS; ++t) if (t < m.n && (int64_t)blockIdx.x >= m.s[t].first_block) k = t;
  const ReduceSet r = m.s[k];
  const int e_l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = ((int64_t)blockIdx.x - r.first_block) * 64 + e_l;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < r.per_slab) {
    const int per = (r.nslab + 3) / 4;
    const int s0 = grp * per, s1 = min(r.nslab, s0 + per);
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      a0 += r.slabs[(int64_t)s * r.per_slab + e];
      a1 += r.slabs[(int64_t)(s + 1) * r.per_slab + e];
      a2 += r.slabs[(int64_t)(s + 2) * r.per_slab + e];
      a3 += r.slabs[(int64_t)(s + 3) * r.per_slab + e];
    }
    for (; s < s1; ++s) a0 += r.slabs[(int64_t)s * r.per_slab + e];
  }
  lds[grp][e_l] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0) {                                        // wave 0: one element per lane
    float sq = 0.f;
    if (e < r.per_slab) {
      const float v = (lds[0][e_l] + lds[1][e_l]) + (lds[2][e_l] + lds[3][e_l]);
      if (e < r.n_w) { r.dw[e] = v; sq = v * v; }
      else if (r.db) { r.db[e - r.n_w] = v; sq = v * v; }
    }
    if (m.normparts) {                                   // this block's share of |grad|^2 (summed in fixed order by the optimiser)
      sq = wave_sum(sq);
      if (e_l == 0) m.normparts[blockIdx.x] = sq;
    }
    if (m.step_state && blockIdx.x == 0 && e_l == 0) m.step_state[0] += 1.f;   // optimiser step counter, ahead of the update kernel
  }
}

// ---- blocked weight gradient: dW[K_in, N] for K_in, N up to 512 as 128 x 128 output blocks ("sets"), all of them and all of
// their row slabs in ONE launch of the slab body (tn_rows_body.h), then ONE fixed-order reduction.  The GAT projections have
// 92 x 264 and 256 x 264 outputs; one set at a time was a launch per set.
template <int NY>
__global__ __launch_bounds__(256) void wgrad_blocks_kernel(WgradBlocks w) {
  extern __shared__ __attribute__((aligned(16))) float tn_smem[];
  wgrad_blocks_role<NY>(w, tn_smem, blockIdx.x, blockIdx.y, gridDim.x);
}
struct WgradBlocksReduce {
  const float* slabs; int nslab; int64_t set_stride;
  int K_in, N, NB, nsets;
  float* dw; int64_t lddw;
  int oi; float* db;                                    // oi: dw is [N][K_in] (torch.nn.Linear's layout); db [N] nullable (oi form)
  int first_block[WB_MAXSETS + 1];
};
__device__ __forceinline__ void wgrad_blocks_reduce_body(const WgradBlocksReduce& r, const int bx) {
  __shared__ float lds[4][64];
  int set = 0;
#pragma unroll
  for (int t = 1; t < WB_MAXSETS; ++t) if (t < r.nsets && bx >= r.first_block[t]) set = t;
  const int kb = set / r.NB, nb = set % r.NB;
  const int kc = min(128, r.K_in - 128 * kb), nc = min(128, r.N - 128 * nb);
  const int64_t per_slab = (int64_t)(kc + 1) * nc;
  const float* slabs = r.slabs + (int64_t)set * r.set_stride;
  const int e_l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = ((int64_t)bx - r.first_block[set]) * 64 + e_l;
  const bool ok = e < ((r.db && kb == 0) ? per_slab : (int64_t)kc * nc);   // (the slabs' last row is colsum(du) of the column block)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (ok) {                                             // the arithmetic (and order) of tn_rows_reduce
    const int per = (r.nslab + 3) / 4;
    const int s0 = grp * per, s1 = min(r.nslab, s0 + per);
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      a0 += slabs[(int64_t)s * per_slab + e];
      a1 += slabs[(int64_t)(s + 1) * per_slab + e];
      a2 += slabs[(int64_t)(s + 2) * per_slab + e];
      a3 += slabs[(int64_t)(s + 3) * per_slab + e];
    }
    for (; s < s1; ++s) a0 += slabs[(int64_t)s * per_slab + e];
  }
  lds[grp][e_l] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && ok) {
    const int k = (int)(e / nc), n = (int)(e % nc);
    const float v = (lds[0][e_l] + lds[1][e_l]) + (lds[2][e_l] + lds[3][e_l]);
    if (k == kc) r.db[128 * nb + n] = v;
    else if (r.oi) r.dw[(int64_t)(128 * nb + n) * r.lddw + 128 * kb + k] = v;
    else r.dw[(int64_t)(128 * kb + k) * r.lddw + 128 * nb + n] = v;
  }
}
__global__ __launch_bounds__(256) void wgrad_blocks_reduce(WgradBlocksReduce r) { wgrad_blocks_reduce_body(r, (int)blockIdx.x); }
// the reductions of TWO blocked weight gradients (the two layers of a GAT encoder) in one launch: blocks [0, n0) work on r0
__global__ __launch_bounds__(256) void wgrad_blocks_reduce2(WgradBlocksReduce r0, WgradBlocksReduce r1, int n0) {
  if ((int)blockIdx.x < n0) wgrad_blocks_reduce_body(r0, (int)blockIdx.x);
  else wgrad_blocks_reduce_body(r1, (int)blockIdx.x - n0);
}

}  // namespace

extern "C" {

int tsgnn_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                   int64_t scm, int64_t scn, int M, int N, int K, int batch, int64_t stride_a, int64_t stride_b,
                   int64_t stride_c, const int* seg_ptr, int ragged, int max_seg, float alpha, int accumulate,
                   tsgnn_stream_t stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0 || batch <= 0) return TSGNN_EINVAL;
  if (seg_ptr && ragged != 1 && ragged != 2) return TSGNN_EINVAL;
  if (!seg_ptr && ragged != 0) return TSGNN_EINVAL;
  const int Mgrid = (seg_ptr && ragged == 2) ? max_seg : M;
  if (Mgrid <= 0 || N == 0) return TSGNN_OK;
  GemmArgs g{A, B, C, sam, sak, sbk, sbn, scm, scn, stride_a, stride_b, stride_c, M, N, K, seg_ptr, ragged, 1, 0,
             alpha, accumulate};
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((Mgrid + BM - 1) / BM), (unsigned)batch);
  gemm_f32_kernel<<<grid, 256, 0, stream>>>(g);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* floats of workspace for tsgnn_gemm_splitk_f32 */
int tsgnn_gemm_splitk_plan(int M, int N, int K, int* ksplit, int64_t* ws_floats) {
  if (!ksplit || !ws_floats || M < 0 || N < 0 || K < 0) return TSGNN_EINVAL;
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int kchunks = (K + BK - 1) / BK;
  int s = tiles > 0 ? (1024 + tiles - 1) / tiles : 1;      // aim at ~1024 blocks (4 per CU)
  if (s > kchunks / 4) s = kchunks / 4;                    // >= 4 K-chunks (128 rows) per split
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  *ksplit = s;
  *ws_floats = (s > 1) ? (int64_t)s * M * N : 0;
  return TSGNN_OK;
}

/* C[M,N] (+)= op(A)[M,K] . op(B)[K,N] with K split over blocks (K = number of graph rows, large).
 * C must be dense row-major (scn == 1, scm == N) when ksplit > 1. */
int tsgnn_gemm_splitk_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                          float* C, int M, int N, int K, int ksplit, float* ws, int accumulate,
                          tsgnn_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K < 0 || ksplit < 1 || (ksplit > 1 && !ws)) return TSGNN_EINVAL;
  GemmArgs g{A, B, ksplit > 1 ? ws : C, sam, sak, sbk, sbn, (int64_t)N, 1, 0, 0, 0, M, N, K, nullptr, 0, ksplit,
             (int64_t)M * N, 1.f, accumulate};
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)(((M + BM - 1) / BM) * ksplit), 1);
  gemm_f32_kernel<<<grid, 256, 0, stream>>>(g);
  if (ksplit > 1) {
    const int64_t n = (int64_t)M * N;
    splitk_reduce_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(ws, ksplit, n, n, C, accumulate);
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}


/* plan: number of row slabs and workspace floats for tsgnn_linear_wgrad_f32 (0 slabs = shape unsupported) */
int tsgnn_linear_wgrad_plan(int64_t rows, int K_in, int N, int64_t ldz, int64_t lddu, int* nslab, int64_t* rows_per_slab,
                            int64_t* ws_floats) {
  if (!nslab || !rows_per_slab || !ws_floats || rows < 0) return TSGNN_EINVAL;
  *nslab = 0; *rows_per_slab = 0; *ws_floats = 0;
  if (K_in <= 0 || N <= 0 || K_in > 128 || N > 128 || (ldz % 4) || (lddu % 4) || (N % 4)) return TSGNN_OK;
  // one round of equal blocks: (slabs x blocks per slab) = number of CUs.  A grid a little over the CU count would
  // put two blocks on some CUs, and those decide the kernel time (measured: 286 blocks 9.9 us, 256 blocks 6.5 us).
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    else ncu = 256;
  }
  const int tiles = ((K_in + 31) / 32) * ((N + 31) / 32);
  const int per_slab_blocks = tiles >= 8 ? 2 : 1;
  int64_t rps = ceil_div64(rows > 0 ? rows : 1, ncu / per_slab_blocks);
  rps = ((rps + 7) / 8) * 8;
  if (rps < 16) rps = 16;
  *rows_per_slab = rps;
  *nslab = (int)(rows > 0 ? ceil_div64(rows, rps) : 1);
  *ws_floats = (int64_t)(*nslab) * (K_in + 1) * N;
  return TSGNN_OK;
}

/* reduce up to four slab sets (written by tsgnn_linear_wgrad_f32 with dw == NULL) in ONE launch: the weight gradients of all
 * layers of a backward pass.  Unused sets: ws == NULL. */
int tsgnn_wgrad_reduce_multi_f32(const float* ws0, int nslab0, int K0, int N0, float* dw0, float* db0, const float* ws1, int nslab1,
                                 int K1, int N1, float* dw1, float* db1, const float* ws2, int nslab2, int K2, int N2, float* dw2,
                                 float* db2, const float* ws3, int nslab3, int K3, int N3, float* dw3, float* db3,
                                 float* normparts, float* step_state, tsgnn_stream_t stream) {
  const float* ws[4] = {ws0, ws1, ws2, ws3};
  const int ns[4] = {nslab0, nslab1, nslab2, nslab3}, Ks[4] = {K0, K1, K2, K3}, Ns[4] = {N0, N1, N2, N3};
  float* dws[4] = {dw0, dw1, dw2, dw3};
  float* dbs[4] = {db0, db1, db2, db3};
  ReduceMulti m;
  m.n = 0;
  int64_t blocks = 0;
  for (int t = 0; t < 4; ++t) {
    if (!ws[t]) continue;
    if (ns[t] <= 0 || Ks[t] <= 0 || Ns[t] <= 0 || !dws[t]) return TSGNN_EINVAL;
    const int64_t per_slab = (int64_t)(Ks[t] + 1) * Ns[t];
    m.s[m.n] = ReduceSet{ws[t], ns[t], per_slab, (int64_t)Ks[t] * Ns[t], dws[t], dbs[t], blocks};
    blocks += ceil_div64(per_slab, 64);
    ++m.n;
  }
  if (m.n == 0) return TSGNN_OK;
  for (int t = m.n; t < RM_SETS; ++t) m.s[t] = m.s[0];
  m.normparts = normparts;
  m.step_state = step_state;
  TSGNN_KNAME("tn_rows_reduce_multi");
  tn_rows_reduce_multi<<<(unsigned)blocks, 256, 0, stream>>>(m);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* dW[K_in,N] = z[:, :K_in]^T . du ; db[N] = colsum(du) (db nullable).  Plan with tsgnn_linear_wgrad_plan.
 * dw == NULL: only the slabs are produced (reduce them later with tsgnn_wgrad_reduce_multi_f32). */
static int linear_wgrad_launch(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                               int64_t rows_per_slab, int64_t bias_only_rows, float* ws, float* dw, float* db, float* du_part, int du_nb,
                               int du_F, float* du_dws, float* du_dbs, tsgnn_stream_t stream) {
  if (!z || !du || !ws || rows < 0 || nslab <= 0 || rows_per_slab <= 0 || K_in <= 0 || N <= 0 || bias_only_rows < 0) return TSGNN_EINVAL;
  if (K_in > 128 || N > 128 || (ldz % 4) || (lddu % 4) || (N % 4) || (reinterpret_cast<uintptr_t>(z) & 15) ||
      (reinterpret_cast<uintptr_t>(du) & 15))
    return TSGNN_EUNSUPPORTED;
  TnArgs g{z, ldz, du, lddu, rows, rows_per_slab, K_in, N, ws, nullptr, bias_only_rows};
  const int mt = (K_in + 31) / 32, nt = (N + 31) / 32;
  const int tiles = mt * nt;
  const unsigned ny = (tiles >= 8 && nslab < 512) ? 2u : 1u;      // two blocks per slab when there is enough tile work
  const dim3 grid_tn((unsigned)nslab, ny);
  TSGNN_KNAME("gemm_tn_rows_kernel<%d,%d,%u>", mt > 4 ? 4 : mt, nt > 4 ? 4 : nt, ny);
#define TSGNN_TN(M_, N_) do { if (ny == 2) gemm_tn_rows_kernel<M_, N_, 2><<<grid_tn, 256, 2 * TN_CH * 32 * (M_ + N_) * sizeof(float), stream>>>(g); \
    else gemm_tn_rows_kernel<M_, N_, 1><<<grid_tn, 256, 2 * TN_CH * 32 * (M_ + N_) * sizeof(float), stream>>>(g); } while (0)
  switch (mt * 10 + nt) {
    case 11: TSGNN_TN(1, 1); break; case 12: TSGNN_TN(1, 2); break; case 13: TSGNN_TN(1, 3); break; case 14: TSGNN_TN(1, 4); break;
    case 21: TSGNN_TN(2, 1); break; case 22: TSGNN_TN(2, 2); break; case 23: TSGNN_TN(2, 3); break; case 24: TSGNN_TN(2, 4); break;
    case 31: TSGNN_TN(3, 1); break; case 32: TSGNN_TN(3, 2); break; case 33: TSGNN_TN(3, 3); break; case 34: TSGNN_TN(3, 4); break;
    case 41: TSGNN_TN(4, 1); break; case 42: TSGNN_TN(4, 2); break; case 43: TSGNN_TN(4, 3); break; default: TSGNN_TN(4, 4); break;
  }
#undef TSGNN_TN
  const int64_t per_slab = (int64_t)(K_in + 1) * N;
  if (dw && du_part)
    tn_rows_reduce_du<<<(unsigned)ceil_div64(per_slab, 64) + 1, 256, 0, stream>>>(ws, nslab, per_slab, (int64_t)K_in * N, dw, db, du_part,
                                                                                   du_nb, du_F, du_dws, du_dbs);
  else if (dw)
    tn_rows_reduce<<<(unsigned)ceil_div64(per_slab, 64), 256, 0, stream>>>(ws, nslab, per_slab, (int64_t)K_in * N, dw, db, nullptr);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_linear_wgrad_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                           int64_t rows_per_slab, int64_t bias_only_rows, float* ws, float* dw, float* db, tsgnn_stream_t stream) {
  return linear_wgrad_launch(z, ldz, du, lddu, rows, K_in, N, nslab, rows_per_slab, bias_only_rows, ws, dw, db, nullptr, 0, 0, nullptr,
                             nullptr, stream);
}

/* tsgnn_linear_wgrad_f32 (dw != NULL) whose reduction launch also sums the nb <= 256 partial rows part[nb][F_du + 4] that
 * tsgnn_sag_pool_graph_bwd_f32 (called with dws = dbs = NULL) left behind: dws[F_du], dbs[1]. */
int tsgnn_linear_wgrad_du_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                              int64_t rows_per_slab, float* ws, float* dw, float* db, float* part, int nb, int F_du, float* dws,
                              float* dbs, tsgnn_stream_t stream) {
  if (!dw || !part || !dws || !dbs || nb <= 0 || nb > 256 || F_du <= 0 || (F_du % 4) || (reinterpret_cast<uintptr_t>(part) & 15) ||
      (reinterpret_cast<uintptr_t>(dws) & 15))
    return TSGNN_EINVAL;
  return linear_wgrad_launch(z, ldz, du, lddu, rows, K_in, N, nslab, rows_per_slab, 0, ws, dw, db, part, nb, F_du, dws, dbs, stream);
}


/* only the reduction of tsgnn_linear_wgrad_du_f32 (part nullable: tsgnn_linear_wgrad_f32's), for slabs [nslab][K_in + 1][N] another
 * launch produced (tsgnn_gat_bwd_products_f32: the slab blocks beside the input-gradient product) */
int tsgnn_linear_wgrad_du_reduce_f32(const float* ws, int nslab, int K_in, int N, float* dw, float* db, float* part, int nb, int F_du,
                                     float* dws, float* dbs, tsgnn_stream_t stream) {
  if (!ws || !dw || nslab <= 0 || K_in <= 0 || N <= 0 || K_in > 128 || N > 128) return TSGNN_EINVAL;
  if (part && (!dws || !dbs || nb <= 0 || nb > 256 || F_du <= 0 || (F_du % 4) || (reinterpret_cast<uintptr_t>(part) & 15) ||
               (reinterpret_cast<uintptr_t>(dws) & 15)))
    return TSGNN_EINVAL;
  const int64_t per_slab = (int64_t)(K_in + 1) * N;
  if (part)
    tn_rows_reduce_du<<<(unsigned)ceil_div64(per_slab, 64) + 1, 256, 0, stream>>>(ws, nslab, per_slab, (int64_t)K_in * N, dw, db, part, nb,
                                                                                   F_du, dws, dbs);
  else
    tn_rows_reduce<<<(unsigned)ceil_div64(per_slab, 64), 256, 0, stream>>>(ws, nslab, per_slab, (int64_t)K_in * N, dw, db, nullptr);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* Ragged batched  out[b][K,N] = S[rows_b, :K]^T . X[rows_b, :N]  (DiffPool's S^T Z and S^T (A S), encoders.py:374-375):
 * every graph is cut into row slabs (slab_row_ptr[nslab+1], graph b owns slabs [seg_slab_ptr[b], seg_slab_ptr[b+1])),
 * each slab is one workgroup of MFMA work, slabs are summed per graph in fixed order.  ceil(K/32)*ceil(N/32) <= 16. */
int tsgnn_ragged_tn_f32(const float* s_mat, int64_t lds_, const float* x, int64_t ldx, int K, int N, const int* slab_row_ptr,
                        int nslab, const int* seg_slab_ptr, int nseg, float* ws, float* out, tsgnn_stream_t stream) {
  if (!s_mat || !x || !slab_row_ptr || !seg_slab_ptr || !ws || !out || K <= 0 || N <= 0 || nslab <= 0 || nseg <= 0) return TSGNN_EINVAL;
  const int mt = (K + 31) / 32, nt = (N + 31) / 32;
  if (mt * nt > 16 || mt > 4 || nt > 8 || (lds_ % 4) || (ldx % 4) || (N % 4) || (reinterpret_cast<uintptr_t>(s_mat) & 15) ||
      (reinterpret_cast<uintptr_t>(x) & 15))
    return TSGNN_EUNSUPPORTED;
  TnArgs g{s_mat, lds_, x, ldx, 0, 0, K, N, ws, slab_row_ptr, 0};
  const unsigned ny = (mt * nt >= 8) ? 2u : 1u;
  const dim3 grid((unsigned)nslab, ny);
#define TSGNN_RT(M_, N_) do { if (ny == 2) gemm_tn_rows_kernel<M_, N_, 2><<<grid, 256, 2 * TN_CH * 32 * (M_ + N_) * sizeof(float), stream>>>(g); \
    else gemm_tn_rows_kernel<M_, N_, 1><<<grid, 256, 2 * TN_CH * 32 * (M_ + N_) * sizeof(float), stream>>>(g); } while (0)
  switch (mt * 10 + nt) {
    case 11: TSGNN_RT(1, 1); break; case 12: TSGNN_RT(1, 2); break; case 13: TSGNN_RT(1, 3); break; case 14: TSGNN_RT(1, 4); break;
    case 15: TSGNN_RT(1, 5); break; case 16: TSGNN_RT(1, 6); break; case 17: TSGNN_RT(1, 7); break; case 18: TSGNN_RT(1, 8); break;
    case 21: TSGNN_RT(2, 1); break; case 22: TSGNN_RT(2, 2); break; case 23: TSGNN_RT(2, 3); break; case 24: TSGNN_RT(2, 4); break;
    case 25: TSGNN_RT(2, 5); break; case 26: TSGNN_RT(2, 6); break; case 27: TSGNN_RT(2, 7); break; case 28: TSGNN_RT(2, 8); break;
    case 31: TSGNN_RT(3, 1); break; case 32: TSGNN_RT(3, 2); break; case 33: TSGNN_RT(3, 3); break; case 34: TSGNN_RT(3, 4); break;
    case 35: TSGNN_RT(3, 5); break;
    case 41: TSGNN_RT(4, 1); break; case 42: TSGNN_RT(4, 2); break; case 43: TSGNN_RT(4, 3); break; default: TSGNN_RT(4, 4); break;
  }
#undef TSGNN_RT
  const int64_t per_slab = (int64_t)(K + 1) * N;
  dim3 rgrid((unsigned)ceil_div64((int64_t)K * N, 64), (unsigned)nseg);
  tn_rows_reduce<<<rgrid, 256, 0, stream>>>(ws, nslab, per_slab, (int64_t)K * N, out, nullptr, seg_slab_ptr);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* out[f] (+)= sum_r x[r,f];  ws >= nchunk*F floats with nchunk = ceil(rows/512) */
/* floats of workspace for tsgnn_wgrad_narrow_f32 */
int tsgnn_wgrad_narrow_plan(int64_t rows, int K_in, int N, int64_t* ws_floats) {
  if (!ws_floats || rows < 0 || K_in <= 0 || N <= 0) return TSGNN_EINVAL;
  const int64_t rpc = rows > 65536 ? 512 : 128;
  *ws_floats = (rows > 0 ? ceil_div64(rows, rpc) : 1) * (int64_t)(K_in + 1) * N;
  return TSGNN_OK;
}

/* dwb[(K_in + 1), N]: rows 0..K_in-1 = dW = z[:, :K_in]^T du, row K_in = db = column sums of du; K_in <= 4 (narrow inputs) */
int tsgnn_wgrad_narrow_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, float* ws,
                           float* dwb, tsgnn_stream_t stream) {
  if (!dwb || rows < 0 || K_in <= 0 || N <= 0) return TSGNN_EINVAL;
  if (K_in > WN_KMAX) return TSGNN_EUNSUPPORTED;
  if (rows == 0) {                                          // no rows: zero gradients (empty tensors have no storage)
    (void)hipMemsetAsync(dwb, 0, sizeof(float) * (size_t)(K_in + 1) * N, stream);
    return TSGNN_OK;
  }
  if (!z || !du || !ws || ldz < K_in || lddu < N) return TSGNN_EINVAL;
  const int64_t rpc = rows > 65536 ? 512 : 128;
  const int nchunk = (int)(rows > 0 ? ceil_div64(rows, rpc) : 1);
  const int64_t n = (int64_t)(K_in + 1) * N;
  wgrad_narrow_partial<<<dim3((unsigned)((N + 63) / 64), (unsigned)nchunk), 256, 0, stream>>>(z, ldz, du, lddu, rows, K_in, N, rpc, ws);
  splitk_reduce_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(ws, nchunk, n, n, dwb, 0);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_colsum_f32(const float* x, int64_t ld, int64_t rows, int F, float* out, float* ws, int accumulate,
                     tsgnn_stream_t stream) {
  if (!x || !out || !ws || rows < 0 || F <= 0 || ld < F) return TSGNN_EINVAL;
  const int64_t rpc = rows > 65536 ? 512 : 128;          // short chunks: a bias gradient is a latency chain, not a stream
  const int nchunk = (int)(rows > 0 ? ceil_div64(rows, rpc) : 1);
  dim3 grid((unsigned)((F + 63) / 64), (unsigned)nchunk);
  colsum_partial<<<grid, 256, 0, stream>>>(x, ld, rows, F, rpc, ws);
  splitk_reduce_kernel<<<(unsigned)ceil_div64(F, 256), 256, 0, stream>>>(ws, nchunk, F, F, out, accumulate);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* the same for up to 8 slab sets described in HOST memory: desc = [n, n x (ws, nslab, K, N, dw, db)] (the gradients of two
 * stacks that share their launches, sage_stack._SageStackPair) */
int tsgnn_wgrad_reduce_sets_f32(const int64_t* desc, tsgnn_stream_t stream) {
  if (!desc) return TSGNN_EINVAL;
  const int n = (int)desc[0];
  if (n < 0 || n > RM_SETS) return TSGNN_EINVAL;
  if (n == 0) return TSGNN_OK;
  ReduceMulti m;
  m.n = n;
  int64_t blocks = 0;
  for (int t = 0; t < n; ++t) {
    const int64_t* d = desc + 1 + 6 * t;
    const float* ws = reinterpret_cast<const float*>(d[0]);
    const int nslab = (int)d[1], K = (int)d[2], N = (int)d[3];
    float* dw = reinterpret_cast<float*>(d[4]);
    float* db = reinterpret_cast<float*>(d[5]);
    if (!ws || !dw || nslab <= 0 || K <= 0 || N <= 0) return TSGNN_EINVAL;
    const int64_t per_slab = (int64_t)(K + 1) * N;
    m.s[t] = ReduceSet{ws, nslab, per_slab, (int64_t)K * N, dw, db, blocks};
    blocks += ceil_div64(per_slab, 64);
  }
  for (int t = n; t < RM_SETS; ++t) m.s[t] = m.s[0];
  m.normparts = nullptr;
  m.step_state = nullptr;
  TSGNN_KNAME("tn_rows_reduce_multi");
  tn_rows_reduce_multi<<<(unsigned)blocks, 256, 0, stream>>>(m);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

static int wgrad_blocks_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    else ncu = 256;
  }
  return ncu;
}

int tsgnn_wgrad_blocks_plan(int64_t rows, int K_in, int N, int64_t ldz, int64_t lddu, int* nslab, int64_t* rows_per_slab,
                            int64_t* ws_floats) {
  if (!nslab || !rows_per_slab || !ws_floats || rows < 0) return TSGNN_EINVAL;
  *nslab = 0; *rows_per_slab = 0; *ws_floats = 0;
  if (K_in <= 0 || N <= 0 || K_in > 512 || N > 512 || (ldz % 4) || (lddu % 4) || (N % 4)) return TSGNN_OK;
  const int nsets = ((K_in + 127) / 128) * ((N + 127) / 128);
  if (nsets > WB_MAXSETS) return TSGNN_OK;
  // two workgroups of 64 KB LDS fit a CU: (slabs x sets x 2 blocks per set) ~ 2 x CUs, every block in flight at once
  static const int per_cu = [] { const char* e = getenv("TSGNN_WGRAD_BLOCKS_PER_CU"); const int v = e ? atoi(e) : 2; return v > 0 ? v : 2; }();
  int64_t ns = (int64_t)per_cu * wgrad_blocks_cus() / (2 * nsets);
  if (ns < 1) ns = 1;
  int64_t rps = ceil_div64(rows > 0 ? rows : 1, ns);
  rps = ((rps + 31) / 32) * 32;                           // whole staged chunks
  *rows_per_slab = rps;
  *nslab = (int)(rows > 0 ? ceil_div64(rows, rps) : 1);
  *ws_floats = (int64_t)nsets * (*nslab) * WB_SET_FLOATS;
  return TSGNN_OK;
}

static int wgrad_blocks_reduce_launch(const float* ws, int nslab, int K_in, int N, float* dw, int64_t lddw, int oi, float* db,
                                      tsgnn_stream_t stream);
static int wgrad_blocks_launch(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                               int64_t rows_per_slab, float* ws, float* dw, int64_t lddw, int oi, float* db, tsgnn_stream_t stream) {
  if (!z || !du || !ws || rows < 0 || nslab <= 0 || rows_per_slab <= 0 || K_in <= 0 || N <= 0 || (dw && lddw < (oi ? K_in : N))) return TSGNN_EINVAL;
  if (K_in > 512 || N > 512 || (ldz % 4) || (lddu % 4) || (N % 4) || ldz < ((K_in + 3) / 4) * 4 || lddu < N ||
      ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(du)) & 15))
    return TSGNN_EUNSUPPORTED;
  const int KB = (K_in + 127) / 128, NB = (N + 127) / 128, nsets = KB * NB;
  if (nsets > WB_MAXSETS || (int64_t)nslab * rows_per_slab < rows) return TSGNN_EINVAL;
  WgradBlocks w{TnArgs{z, ldz, du, lddu, rows, rows_per_slab, K_in, N, ws, nullptr, 0}, NB, nsets, (int64_t)nslab * WB_SET_FLOATS};
  static bool attr = false;
  constexpr size_t lds = 2 * TN_CH * 32 * (4 + 4) * sizeof(float);
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_blocks_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  TSGNN_KNAME("wgrad_blocks_kernel<2>");
  wgrad_blocks_kernel<2><<<dim3((unsigned)nslab, (unsigned)(2 * nsets)), 256, lds, stream>>>(w);
  if (!dw) {                                              // slabs only: the caller reduces them later (tsgnn_wgrad_blocks_reduce[2]_f32)
    TSGNN_CHECK_LAUNCH();
    return TSGNN_OK;
  }
  return wgrad_blocks_reduce_launch(ws, nslab, K_in, N, dw, lddw, oi, db, stream);
}

static WgradBlocksReduce wgrad_blocks_reduce_args(const float* ws, int nslab, int K_in, int N, float* dw, int64_t lddw, int oi, float* db,
                                                  int* nblocks) {
  const int KB = (K_in + 127) / 128, NB = (N + 127) / 128, nsets = KB * NB;
  WgradBlocksReduce r{ws, nslab, (int64_t)nslab * WB_SET_FLOATS, K_in, N, NB, nsets, dw, lddw, oi, db, {0}};
  int blocks = 0;
  for (int t = 0; t < nsets; ++t) {
    r.first_block[t] = blocks;
    const int kc = K_in - 128 * (t / NB) < 128 ? K_in - 128 * (t / NB) : 128, nc = N - 128 * (t % NB) < 128 ? N - 128 * (t % NB) : 128;
    blocks += (((db && t / NB == 0) ? kc + 1 : kc) * nc + 63) / 64;
  }
  for (int t = nsets; t <= WB_MAXSETS; ++t) r.first_block[t] = blocks;
  *nblocks = blocks;
  return r;
}
static int wgrad_blocks_reduce_launch(const float* ws, int nslab, int K_in, int N, float* dw, int64_t lddw, int oi, float* db,
                                      tsgnn_stream_t stream) {
  int blocks = 0;
  const WgradBlocksReduce r = wgrad_blocks_reduce_args(ws, nslab, K_in, N, dw, lddw, oi, db, &blocks);
  wgrad_blocks_reduce<<<(unsigned)blocks, 256, 0, stream>>>(r);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* the fixed-order reduction of slabs a blocked weight-gradient launch left in ws (tsgnn_gat_bwd_products_f32) */
int tsgnn_wgrad_blocks_reduce_f32(const float* ws, int nslab, int K_in, int N, float* dw, int64_t lddw, tsgnn_stream_t stream) {
  if (!ws || !dw || nslab <= 0 || K_in <= 0 || N <= 0 || K_in > 512 || N > 512 || lddw < N) return TSGNN_EINVAL;
  if (((K_in + 127) / 128) * ((N + 127) / 128) > WB_MAXSETS) return TSGNN_EINVAL;
  return wgrad_blocks_reduce_launch(ws, nslab, K_in, N, dw, lddw, 0, nullptr, stream);
}

/* the same into torch.nn.Linear's layout: dw_oi[N][K_in], db[N] (nullable) — after tsgnn_linear_bwd_products_f32 */
int tsgnn_wgrad_blocks_reduce_oi_f32(const float* ws, int nslab, int K_in, int N, float* dw_oi, int64_t lddw, float* db, tsgnn_stream_t stream) {
  if (!ws || !dw_oi || nslab <= 0 || K_in <= 0 || N <= 0 || K_in > 512 || N > 512 || lddw < K_in) return TSGNN_EINVAL;
  if (((K_in + 127) / 128) * ((N + 127) / 128) > WB_MAXSETS) return TSGNN_EINVAL;
  return wgrad_blocks_reduce_launch(ws, nslab, K_in, N, dw_oi, lddw, 1, db, stream);
}

/* two such reductions in one launch (the two layers of a GAT encoder's backward) */
int tsgnn_wgrad_blocks_reduce2_f32(const float* ws0, int nslab0, int K0, int N0, float* dw0, int64_t lddw0, const float* ws1, int nslab1,
                                   int K1, int N1, float* dw1, int64_t lddw1, tsgnn_stream_t stream) {
  if (!ws0 || !dw0 || !ws1 || !dw1 || nslab0 <= 0 || nslab1 <= 0 || K0 <= 0 || N0 <= 0 || K1 <= 0 || N1 <= 0 || K0 > 512 || N0 > 512 ||
      K1 > 512 || N1 > 512 || lddw0 < N0 || lddw1 < N1)
    return TSGNN_EINVAL;
  if (((K0 + 127) / 128) * ((N0 + 127) / 128) > WB_MAXSETS || ((K1 + 127) / 128) * ((N1 + 127) / 128) > WB_MAXSETS) return TSGNN_EINVAL;
  int n0 = 0, n1 = 0;
  const WgradBlocksReduce r0 = wgrad_blocks_reduce_args(ws0, nslab0, K0, N0, dw0, lddw0, 0, nullptr, &n0);
  const WgradBlocksReduce r1 = wgrad_blocks_reduce_args(ws1, nslab1, K1, N1, dw1, lddw1, 0, nullptr, &n1);
  TSGNN_KNAME("wgrad_blocks_reduce2");
  wgrad_blocks_reduce2<<<(unsigned)(n0 + n1), 256, 0, stream>>>(r0, r1, n0);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_wgrad_blocks_slabs_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                                 int64_t rows_per_slab, float* ws, tsgnn_stream_t stream) {
  return wgrad_blocks_launch(z, ldz, du, lddu, rows, K_in, N, nslab, rows_per_slab, ws, nullptr, 0, 0, nullptr, stream);
}

int tsgnn_wgrad_blocks_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                           int64_t rows_per_slab, float* ws, float* dw, int64_t lddw, tsgnn_stream_t stream) {
  return wgrad_blocks_launch(z, ldz, du, lddu, rows, K_in, N, nslab, rows_per_slab, ws, dw, lddw, 0, nullptr, stream);
}

int tsgnn_wgrad_blocks_oi_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                              int64_t rows_per_slab, float* ws, float* dw_oi, int64_t lddw, float* db, tsgnn_stream_t stream) {
  return wgrad_blocks_launch(z, ldz, du, lddu, rows, K_in, N, nslab, rows_per_slab, ws, dw_oi, lddw, 1, db, stream);
}

}  // extern "C"
