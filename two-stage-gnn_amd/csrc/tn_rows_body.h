// Body of the weight-gradient slab kernel (see gemm.hip), as a device function so that it can share a launch with
// independent work (layer_bwd.hip).  (bx, by_) = (slab, block within the slab), nslab_grid = number of slabs.
#pragma once
#include "common.h"
#include <type_traits>

namespace {

typedef float tn_f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- weight gradient over row slabs
// dW[K_in, N] = Z[:, :K_in]^T . dU   and   db[N] = colsum(dU)   (backward of encoders.py:36-38).
// The reduction runs over the graph rows (thousands+), the output is tiny, so each workgroup owns a SLAB of
// rows and produces a full [K_in+1, N] partial (row K_in = bias partial) from coalesced 16-byte row loads;
// both MFMA operands are natural row-major LDS tiles (lane i reads consecutive floats: conflict-free).
// Slabs are summed in fixed order by tn_rows_reduce (bitwise reproducible, no float atomics).
constexpr int TN_CH = 32;            // rows per staged chunk

__device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct TnArgs {
  const float* z; int64_t ldz;
  const float* du; int64_t lddu;
  int64_t rows, rows_per_slab;
  int K_in, N;
  float* slabs;                      // [nslab][K_in + 1][N]
  const int* slab_row_ptr;           // nullable: slab s covers rows [slab_row_ptr[s], slab_row_ptr[s+1]) (ragged, per graph)
  int64_t bias_only_rows;            // rows after `rows` whose dU counts for the bias partial only
};

template <int MT, int NTt, int NY>
__device__ __forceinline__ void tn_rows_body(const TnArgs& g, float* tn_smem, unsigned bx, unsigned by_, unsigned nslab_grid) {
  constexpr int KP = 32 * MT, NP = 32 * NTt;
  constexpr int TILES = MT * NTt;
  constexpr bool COLW = NTt == 4;                      // wave w owns output column tile w: one dU fragment set per chunk
  // NY blocks share a slab.  Tiles per wave is what sizes the register file: arch + accumulation VGPRs
  // must stay <= 256 so that two blocks fit a CU (a grid a little over 256 blocks must not need a second round).
  constexpr int TPW = COLW ? (MT + NY - 1) / NY : ((TILES + NY - 1) / NY + 3) / 4;
  constexpr int ZV = MT, UV = NTt;                     // float4 per thread per chunk (TN_CH * KP / 1024, TN_CH * NP / 1024)
  constexpr int STAGE = TN_CH * (KP + NP);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int64_t r0 = g.slab_row_ptr ? (int64_t)g.slab_row_ptr[bx] : (int64_t)bx * g.rows_per_slab;
  const int64_t r1 = g.slab_row_ptr ? (int64_t)g.slab_row_ptr[bx + 1] : min(g.rows, r0 + g.rows_per_slab);
  // NY blocks share a slab and split its output tiles (every block stages the whole chunk: L2 hits)
  constexpr int ny = NY;
  const int by = (int)by_;
  TR(0);
  int tm_[TPW], tn_[TPW];
  bool tok[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (COLW) {
      tm_[t] = by + t * ny; tn_[t] = wid; tok[t] = tm_[t] < MT;
    } else {
      const int f = (wid + 4 * t) * ny + by;
      tok[t] = (wid + 4 * t) < (TILES + ny - 1) / ny && f < TILES;
      tm_[t] = f / NTt; tn_[t] = f % NTt;
    }
  }
  tn_f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbacc = 0.f;

  // staging maps: thread -> (row m of the chunk, float4 column c4), fixed for the whole slab
  const float* zp[ZV];
  const float* up[UV];
  int zm[ZV], um[UV], znv[ZV];
  bool ucol[UV];
#pragma unroll
  for (int q = 0; q < ZV; ++q) {
    const int idx = q * 256 + tid, m = idx / (KP / 4), c4 = idx % (KP / 4);
    zm[q] = m;
    znv[q] = min(4, max(0, g.K_in - 4 * c4));            // valid floats of this float4 (row padding of Z contributes nothing)
    zp[q] = g.z + (r0 + m) * g.ldz + (znv[q] > 0 ? 4 * c4 : 0);
  }
#pragma unroll
  for (int q = 0; q < UV; ++q) {
    const int idx = q * 256 + tid, m = idx / (NP / 4), c4 = idx % (NP / 4);
    um[q] = m;
    ucol[q] = 4 * c4 < g.N;
    up[q] = g.du + (r0 + m) * g.lddu + (ucol[q] ? 4 * c4 : 0);
  }
  // two register staging sets: chunk c travels in set c & 1; the first two chunks of a slab are both in flight before
  // any MFMA, later ones are fetched two iterations ahead.  Row validity is applied at commit time so that nothing
  // waits on a load while the MFMAs run; loads never leave the matrix.
  float4 rz[2][ZV], ru[2][UV];
  unsigned zrow[2] = {0, 0}, urow[2] = {0, 0};         // bit q: the row of rz[.][q] / ru[.][q] exists
  auto fetch = [&](auto set_, int64_t off) {
    constexpr int S = decltype(set_)::value;
    zrow[S] = 0; urow[S] = 0;
#pragma unroll
    for (int q = 0; q < ZV; ++q) {
      const bool ok = r0 + off + zm[q] < r1;
      rz[S][q] = *reinterpret_cast<const float4*>(ok ? zp[q] + off * g.ldz : g.z);
      zrow[S] |= ok ? (1u << q) : 0u;
    }
#pragma unroll
    for (int q = 0; q < UV; ++q) {
      const bool ok = r0 + off + um[q] < r1;
      ru[S][q] = *reinterpret_cast<const float4*>(ok ? up[q] + off * g.lddu : g.du);
      urow[S] |= ok ? (1u << q) : 0u;
    }
  };
  auto commit = [&](auto set_, float* st) {
    constexpr int S = decltype(set_)::value;
#pragma unroll
    for (int q = 0; q < ZV; ++q) {
      const int idx = q * 256 + tid;
      float4 v = rz[S][q];
      const int nv = ((zrow[S] >> q) & 1u) ? znv[q] : 0;
      if (nv < 4) v.w = 0.f;
      if (nv < 3) v.z = 0.f;
      if (nv < 2) v.y = 0.f;
      if (nv < 1) v.x = 0.f;
      *reinterpret_cast<float4*>(st + 4 * idx) = v;      // [m][KP] row-major == idx order
    }
#pragma unroll
    for (int q = 0; q < UV; ++q) {
      const int idx = q * 256 + tid;
      *reinterpret_cast<float4*>(st + TN_CH * KP + 4 * idx) =
          (ucol[q] && ((urow[S] >> q) & 1u)) ? ru[S][q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // iteration c (stage / set parity P = c & 1):  operands of chunk c from LDS stage P | MFMAs | fetch chunk c+2 into
  // set P | commit chunk c+1 (set P^1) into stage P^1 | barrier
  auto body = [&](auto par_, int64_t off, int cidx) {
    constexpr int P = decltype(par_)::value;
    const float* Zs = tn_smem + P * STAGE;
    const float* Us = Zs + TN_CH * KP;
    TR(2 + 3 * min(cidx, 2));
    // operands of the whole chunk into registers first (A[m][k] = Z[row k][m], B[k][j] = dU[row k][j]; lane i reads
    // consecutive floats: conflict-free)
    float bfr[COLW ? 1 : TPW][TN_CH / 2];
    float afr[TPW][TN_CH / 2];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (t == 0 || tok[t]) {                            // uniform per wave
        if (!COLW || t == 0) {
#pragma unroll
          for (int s2 = 0; s2 < TN_CH / 2; ++s2) bfr[COLW ? 0 : t][s2] = Us[(2 * s2 + h) * NP + tn_[t] * 32 + i];
        }
#pragma unroll
        for (int s2 = 0; s2 < TN_CH / 2; ++s2) afr[t][s2] = Zs[(2 * s2 + h) * KP + (tok[t] ? tm_[t] : 0) * 32 + i];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                 // keep the LDS reads above the MFMA chains
    // a short last chunk runs only the MFMA groups (4 steps = 8 rows) that hold rows; the rest of the stage is zero
    const int nsteps = (int)min<int64_t>(TN_CH / 2, (r1 - r0 - off + 1) / 2);
#pragma unroll
    for (int s4 = 0; s4 < TN_CH / 2; s4 += 4) {
      if (s4 < nsteps) {
#pragma unroll
        for (int s2 = s4; s2 < s4 + 4; ++s2)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[0][s2], bfr[0][s2], acc[0], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (r0 + off + 2 * TN_CH < r1) fetch(par_, off + 2 * TN_CH);
    if (COLW) {                                        // bias gradient: column sums straight from the dU fragments
#pragma unroll
      for (int s2 = 0; s2 < TN_CH / 2; ++s2) dbacc += bfr[0][s2];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < TPW; ++t) {
      if (tok[t]) {
#pragma unroll
        for (int s4 = 0; s4 < TN_CH / 2; s4 += 4) {
          if (s4 < nsteps) {
#pragma unroll
            for (int s2 = s4; s2 < s4 + 4; ++s2)
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[t][s2], bfr[COLW ? 0 : t][s2], acc[t], 0, 0, 0);
          }
        }
      }
    }
    TR(3 + 3 * min(cidx, 2));
    if (!COLW && by == 0 && tid < NP) {
#pragma unroll 8
      for (int m = 0; m < TN_CH; ++m) dbacc += Us[m * NP + tid];
    }
    if (r0 + off + TN_CH < r1) commit(std::integral_constant<int, P ^ 1>{}, tn_smem + (P ^ 1) * STAGE);
    __syncthreads();
    TR(4 + 3 * min(cidx, 2));
  };
  if (r0 < r1) {
    fetch(S0{}, 0);
    if (r0 + TN_CH < r1) fetch(S1{}, TN_CH);
    TR(1);
    commit(S0{}, tn_smem);
    __syncthreads();
    int cidx = 0;
    for (int64_t off = 0; r0 + off < r1; off += 2 * TN_CH, cidx += 2) {
      body(S0{}, off, cidx);
      if (r0 + off + TN_CH < r1) body(S1{}, off + TN_CH, cidx + 1);
    }
  }
  TR(11);
  float* slab = g.slabs + (int64_t)bx * (g.K_in + 1) * g.N;
  const bool full = g.K_in == KP && g.N == NP;          // uniform: no per-element predicates
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (tok[t]) {
      const int cn = tn_[t] * 32 + i;
      float* sp = slab + (int64_t)(tm_[t] * 32 + 4 * h) * g.N + cn;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st_out(sp + ((r & 3) + 8 * (r >> 2)) * g.N, acc[t][r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cm = tm_[t] * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (cm < g.K_in && cn < g.N) sp[((r & 3) + 8 * (r >> 2)) * g.N] = acc[t][r];
        }
      }
    }
  }
  if (by == 0 && g.bias_only_rows > 0) {                // this slab's share of the rows that only feed the bias gradient
    const int64_t per = ceil_div_dev(g.bias_only_rows, (int64_t)nslab_grid);
    const int64_t e0 = g.rows + (int64_t)bx * per, e1 = min(g.rows + g.bias_only_rows, e0 + per);
    if (COLW) {
      if (wid * 32 + i < g.N)
        for (int64_t e = e0 + h; e < e1; e += 2) dbacc += g.du[e * g.lddu + wid * 32 + i];
    } else if (tid < g.N) {
      for (int64_t e = e0; e < e1; ++e) dbacc += g.du[e * g.lddu + tid];
    }
  }
  if (by == 0) {
    if (COLW) {
      const float v = dbacc + __shfl_xor(dbacc, 32, 64);   // rows of parity h=0 and h=1
      if (lane < 32 && wid * 32 + i < g.N) slab[(int64_t)g.K_in * g.N + wid * 32 + i] = v;
    } else if (tid < g.N) {
      slab[(int64_t)g.K_in * g.N + tid] = dbacc;
    }
  }
  TR(12);
  TR_END();
}

template <int MT, int NTt>
constexpr size_t tn_rows_lds_bytes() { return 2 * TN_CH * 32 * (MT + NTt) * sizeof(float); }

}  // namespace
