// Large-batch variant of the row-panel fp32 MFMA product (rowgemm.hip):  C[R, N] = A[R, K] . B (+ bias, row L2 normalise),
// K <= 128, N <= 128, R in the hundreds of thousands (the GraphConv transform of encoders.py:36-40 and dZ = dU . W^T on
// batches of thousands of graphs, where the stack runs the aggregation as its own launch).
//
// The small-batch kernel re-stages W (64 KB) through LDS for every 32-row panel and keeps one wave per SIMD waiting through
// each of its phases: 0.37 of the fp32-MFMA peak at scale (profiles/r01/gemm_sweep.txt).  This variant is B-STATIONARY:
//   * a block of 4 waves is persistent over panels p = blockIdx.x, + gridDim.x, ...; wave w owns output columns [32w, 32w + 32)
//     and holds its slice of B for the WHOLE K in registers (K/2 = 64 VGPRs: one operand per v_mfma_f32_32x32x2_f32), loaded
//     once per block — no B traffic and no B fragments in LDS afterwards;
//   * the A panel (32 rows x K, contiguous rows) is fetched with 128-byte row segments two panels ahead of its use
//     (global -> registers during the MFMAs of the current panel -> LDS before the barrier), double-buffered in LDS with the
//     XOR-swizzled float4 columns of the gather kernel (conflict-free ds_read_b128 fragments);
//   * ONE barrier per panel: it hands over the next A panel and the four waves' partial row sums of squares together;
//   * 32 KB of LDS and <= 128 VGPRs + 16 accumulators per wave: several blocks per CU, so a SIMD always has another wave's
//     MFMA chain to issue while one waits at the barrier or on its stores.
// The MFMA sequence (k pairing {8u + c, 8u + 4 + c}, u then c ascending) and the epilogue arithmetic are those of
// rowgemm_body.h, so the two kernels agree bit for bit on the same operands (tests/test_gpu_kernels.py).
#pragma once
#include "rowgemm_body.h"

namespace {

#ifndef TSGNN_BIG_MINBLOCKS
#define TSGNN_BIG_MINBLOCKS 2
#endif

template <int U, bool TRANS_B>   // U = groups of 8 k: K <= 8 U
__global__ __launch_bounds__(256, TSGNN_BIG_MINBLOCKS) void rowgemm_big_kernel(RowGemmArgs g, unsigned npanels) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (gridDim.y > 1) {                       // column blocks of 128 (products without a row epilogue: N up to 384)
    const int n0 = (int)blockIdx.y * 128;
    g.N = min(128, g.N - n0);
    g.c += n0;
    if (g.bias) g.bias += n0;
    g.b += TRANS_B ? (int64_t)n0 * g.ldb : (int64_t)n0;
  }
  constexpr int LDB = 8 * U;                 // floats per LDS row
  constexpr int QV = U / 4;                  // float4 per thread and panel (8 threads per row)
  float* red = smem + 2 * 32 * LDB;          // [2][32 rows][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int cn = wid * 32 + i;               // this lane's output column
  const bool col_ok = cn < g.N;

  if (blockIdx.x == gridDim.x - 1 && g.fill_rows > 0) {
    // the ghost rows' constant output ([normalised] bias), exactly as the small kernel's filler block writes it
    const int N4 = g.N / 4, rpp = 256 / N4;
    const int c4 = tid % N4, rsub = tid / N4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && rsub < rpp) bv = ldg4(g.bias + 4 * c4);
    float ss = rsub == 0 ? (bv.x * bv.x + bv.y * bv.y) + (bv.z * bv.z + bv.w * bv.w) : 0.f;
    ss = wave_sum(ss);
    if (lane == 0) red[wid] = ss;
    __syncthreads();
    float sc = 1.f;
    if (g.normalize) sc = fminf(__builtin_amdgcn_rsqf((red[0] + red[1]) + (red[2] + red[3])), 1.0f / NORM_EPS);
    const float4 out = make_float4(bv.x * sc, bv.y * sc, bv.z * sc, bv.w * sc);
    if (rsub < rpp) {
      for (int64_t r = rsub; r < g.fill_rows; r += rpp) {
        *reinterpret_cast<float4*>(g.c + (g.rows + r) * g.ldc + 4 * c4) = out;
        if (g.rinv && c4 == 0) g.rinv[g.rows + r] = sc;
      }
    }
    __syncthreads();
  }

  // B slice of this wave, whole K: operand of MFMA step (u, c) is B[k = 8u + 4h + c][cn]
  float bq[4 * U];
  const int cnc = col_ok ? cn : 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {                          // all requests first (clamped addresses) ...
    if (TRANS_B) {                                         // W row cn is contiguous in k: one 16-byte load per group (K % 4 == 0)
      const int k = 8 * u + 4 * h;
      const float4 t = ldg4(g.b + (int64_t)cnc * g.ldb + (k < g.K ? k : 0));
      bq[4 * u] = t.x; bq[4 * u + 1] = t.y; bq[4 * u + 2] = t.z; bq[4 * u + 3] = t.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = 8 * u + 4 * h + c;
        bq[4 * u + c] = g.b[(int64_t)(k < g.K ? k : 0) * g.ldb + cnc];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u)                            // ... then the masks (columns >= N, k >= K contribute zero)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (!(col_ok && (8 * u + 4 * h + c) < g.K)) bq[4 * u + c] = 0.f;
  const float bias_v = (g.bias && col_ok) ? g.bias[cn] : 0.f;

  // staging map: thread -> row sr of the panel, float4 columns (tid & 7) + 8 q
  const int sr = tid >> 3, sc0 = tid & 7;
  float4 st[QV];
  // Loads are unconditional, from clamped (always mapped) addresses: a predicated load sits in its own exec-masked block and
  // the waitcnt pass then drains vmcnt before the first MFMA, i.e. the prefetch would not overlap anything.  Validity is
  // applied when the registers go to LDS.
  bool st_ok[QV];
  auto gload = [&](unsigned panel) {
    const int64_t row = (int64_t)panel * 32 + sr;
    const bool rok = row < g.rows;
    const float* ap = g.a + (rok ? row : 0) * g.lda;
#pragma unroll
    for (int q = 0; q < QV; ++q) {
      const int k4 = sc0 + 8 * q;
      const bool kok = 4 * k4 < g.K;                       // a float4 that straddles K is taken whole: the row padding is zero
      st_ok[q] = rok && kok;                               // and B's rows >= K are zero
      st[q] = ldg4(ap + (kok ? 4 * k4 : 0));
    }
  };
  auto lstore = [&](float* buf) {
#pragma unroll
    for (int q = 0; q < QV; ++q) {
      const int k4 = sc0 + 8 * q;
      *reinterpret_cast<float4*>(buf + sr * LDB + 4 * (k4 ^ (sr & 7))) = st_ok[q] ? st[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  unsigned p = blockIdx.x;
  if (p < npanels) { gload(p); lstore(smem); }
  __syncthreads();
  for (int it = 0; p < npanels; p += gridDim.x, ++it) {
    const int cur = it & 1;
    const unsigned pn = p + gridDim.x;
    const bool more = pn < npanels;
    gload(more ? pn : p);                                  // lands while this panel's MFMAs run (last round: a harmless re-read)
    __builtin_amdgcn_sched_barrier(0);                     // (the scheduler would sink the requests below the MFMA chain)
    const float* As = smem + cur * (32 * LDB);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (32 * wid < g.N) {                                  // (a wave without columns only helps staging: narrow column blocks)
    // A fragments four k-groups at a time, the next four requested before the current sixteen MFMAs issue (32 VGPRs instead
    // of 64 for the whole K: the kernel fits three waves per SIMD)
    constexpr int NCH = U / 4;
    float4 av[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) av[0][q] = *reinterpret_cast<const float4*>(As + i * LDB + 4 * ((2 * q + h) ^ (i & 7)));
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (ch + 1 < NCH) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          av[(ch + 1) & 1][q] = *reinterpret_cast<const float4*>(As + i * LDB + 4 * ((2 * (4 * (ch + 1) + q) + h) ^ (i & 7)));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u = 4 * ch + q;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ch & 1][q].x, bq[4 * u + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ch & 1][q].y, bq[4 * u + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ch & 1][q].z, bq[4 * u + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ch & 1][q].w, bq[4 * u + 3], acc, 0, 0, 0);
      }
    }
    }
    // epilogue, part 1 (registers): + bias, partial row sums of squares over this wave's 32 columns
    const int64_t m0 = (int64_t)p * 32;
    float* rd = red + cur * 128;
    if (g.bias || g.normalize) {
      float ss[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = col_ok ? acc[r] + bias_v : 0.f;
        acc[r] = v;
        ss[r] = fmaf(v, v, 0.f);
      }
      if (g.normalize) {
        float tot = row16_sum_transpose(ss);
        tot += __shfl_xor(tot, 16, 64);
        if ((lane & 16) == 0) {
          const int r = lane & 15;
          rd[((r & 3) + 8 * (r >> 2) + 4 * h) * 4 + wid] = tot;
        }
      }
    }
    if (more) lstore(smem + (cur ^ 1) * (32 * LDB));
    __syncthreads();                                       // next panel in LDS, the four waves' partial sums in `rd`
    float scale[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) scale[r] = 1.f;
    if (g.normalize) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 q4 = *reinterpret_cast<const float4*>(rd + ((r & 3) + 8 * (r >> 2) + 4 * h) * 4);
        scale[r] = fminf(__builtin_amdgcn_rsqf((q4.x + q4.y) + (q4.z + q4.w)), 1.0f / NORM_EPS);
      }
      if (g.rinv && wid == 0 && lane < 32 && (m0 + lane) < g.rows) {
        const float4 q4 = *reinterpret_cast<const float4*>(rd + lane * 4);
        g.rinv[m0 + lane] = fminf(__builtin_amdgcn_rsqf((q4.x + q4.y) + (q4.z + q4.w)), 1.0f / NORM_EPS);
      }
    }
    if (col_ok) {
      float* cp = g.c + (m0 + 4 * h) * g.ldc + cn;
      if (m0 + 32 <= g.rows) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = acc[r] * scale[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (gm < g.rows) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = acc[r] * scale[r];
        }
      }
    }
  }
}

template <int U, bool TRANS_B>
constexpr size_t rowgemm_big_lds_bytes() { return sizeof(float) * (2 * 32 * 8 * U + 2 * 128); }

}  // namespace
