// Ragged batched  out[b][K, N] = S[rows_b, :K]^T . X[rows_b, :N]  for SHORT segments and a narrow S (DiffPool's first contraction,
// encoders.py:374-375: S^T Z and S^T (A S) over each graph's few hundred rows, K = 64 assignment columns), both products in ONE launch
// and without partial slabs: workgroup (32 x 32 output tile, graph) runs over ALL rows of its graph.  The reduction dimension is the row
// index, so both MFMA operands are natural row-major rows — lane (i, h) of v_mfma_f32_32x32x2 takes S[row 2s + h][32 tm + i] and
// X[row 2s + h][32 tn + i] straight from global memory (128 contiguous bytes per half wave, no LDS staging, 32 requests per lane in
// flight); the eight waves of a workgroup split the graph's rows, their accumulators meet in LDS and are added in wave
// order (bitwise reproducible).  The slab kernel + per-graph reduction this replaces (gemm.hip: tsgnn_ragged_tn_f32) took four
// launches for the two products.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

typedef float rt_f32x16 __attribute__((ext_vector_type(16)));

struct RaggedTn {
  const float* s; int64_t lds_; int K;
  const int* graph_ptr;
  const float* x[2]; int64_t ldx[2]; int N[2]; float* out[2];
  int nt0;                                    // column tiles of operand 0 (blocks x >= nt0 work on operand 1)
  // nullable: the max readout of operand 0 over each graph's node slots (encoders.py:353) from the values the tm = 0 workgroups
  // hold anyway.  ro_ghost: the graph's padded slots (sizes < nmax) are ZERO rows n_real + slot that take part (trap T5).
  float* ro_out; int64_t ro_ldo; int* ro_arg; int ro_nmax; int64_t ro_n_real; int ro_ghost;
};

constexpr int RT_WAVES = 8;
constexpr int RT_STEPS = 16;                  // row pairs per batch: 2 * 16 requests per lane in flight

// grid (mt * (nt0 + nt1), B): one 32 x 32 tile of one product of one graph per workgroup (DD b16, K = 64, N = 192 + 64: 256 workgroups)
__global__ __launch_bounds__(64 * RT_WAVES) void ragged_tn_direct_kernel(RaggedTn a, int mt) {
  __shared__ __attribute__((aligned(16))) float part[RT_WAVES * 1024];       // [RT_WAVES][16][64]
  __shared__ unsigned long long ro_best[RT_WAVES][32];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int tm = (int)blockIdx.x % mt, tcol = (int)blockIdx.x / mt;
  const int op = tcol >= a.nt0 ? 1 : 0;
  const int tn = tcol - (op ? a.nt0 : 0);
  const float* __restrict__ x = a.x[op];
  const int64_t ldx = a.ldx[op];
  const int N = a.N[op];
  const int r0 = a.graph_ptr[b], r1 = a.graph_ptr[b + 1];
  // rows of this wave: an even share of the graph's row PAIRS
  const int pairs = (r1 - r0 + 1) / 2;
  const int per = (pairs + RT_WAVES - 1) / RT_WAVES;
  const int p0 = min(pairs, wid * per), p1 = min(pairs, p0 + per);
  const int cn = tn * 32 + i;
  const float* xp = x + (cn < N ? cn : 0);
  const bool s_ok = tm * 32 + i < a.K;
  const float* sp = a.s + (s_ok ? tm * 32 + i : 0);
  rt_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int last = max(r0, r1 - 1);                                   // clamp target (an empty graph runs no batch)
  const bool ro = a.ro_out && op == 0 && tm == 0;                     // uniform over the workgroup
  unsigned long long best = 0ull;                                     // packed (ordered value, ~row): ties go to the smallest row
  for (int p = p0; p < p1; p += RT_STEPS) {
    float av[RT_STEPS], bv[RT_STEPS];
#pragma unroll
    for (int u = 0; u < RT_STEPS; ++u) {
      const int row = min(r0 + 2 * (p + u) + h, last);                // always a mapped row; validity is applied to the A value
      bv[u] = xp[(int64_t)row * ldx];
      av[u] = sp[(int64_t)row * a.lds_];
    }
#pragma unroll
    for (int u = 0; u < RT_STEPS; ++u) {
      const int row = r0 + 2 * (p + u) + h;
      const bool row_ok = (p + u) < p1 && row < r1;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32((row_ok && s_ok) ? av[u] : 0.f, bv[u], acc, 0, 0, 0);
      if (ro && row_ok) {
        const unsigned long long q = ((unsigned long long)f32_ordered(bv[u]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)row);
        best = q > best ? q : best;
      }
    }
  }
  if (ro) {
    const unsigned long long o = __shfl_xor(best, 32, 64);            // the other row parity of the same column
    best = o > best ? o : best;
    if (h == 0) ro_best[wid][i] = best;
  }
  // the waves' accumulators meet in LDS: element (r, lane) of wave w at (w * 16 + r) * 64 + lane
#pragma unroll
  for (int r = 0; r < 16; ++r) part[(wid * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  float* __restrict__ out = a.out[op] + (int64_t)b * a.K * N;
  for (int e = tid; e < 1024; e += 64 * RT_WAVES) {
    float v = part[e];
#pragma unroll
    for (int w = 1; w < RT_WAVES; ++w) v += part[w * 1024 + e];
    const int l = e & 63, r = e >> 6;
    const int cm = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    const int c = tn * 32 + (l & 31);
    if (cm < a.K && c < N) out[(int64_t)cm * N + c] = v;
  }
  if (ro && tid < 32 && cn < N) {
    unsigned long long m = ro_best[0][tid];
#pragma unroll
    for (int w = 1; w < RT_WAVES; ++w) { const unsigned long long o = ro_best[w][tid]; m = o > m ? o : m; }
    const int sz = r1 - r0;
    if (a.ro_ghost && sz < a.ro_nmax) {                               // the first padded slot's (zero) row
      const unsigned long long gq = ((unsigned long long)f32_ordered(0.f) << 32) |
                                    (unsigned long long)(0xFFFFFFFFu - (unsigned)(a.ro_n_real + sz));
      m = gq > m ? gq : m;
    }
    a.ro_out[(int64_t)b * a.ro_ldo + cn] = m ? ordered_f32((unsigned)(m >> 32)) : 0.f;
    a.ro_arg[(int64_t)b * N + cn] = m ? (int)(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull)) : -1;
  }
}

}  // namespace

extern "C" {

int tsgnn_ragged_tn_direct_supported(int K, int64_t max_rows) { return K > 0 && K <= 128 && max_rows <= 4096; }

/* out0[b] = S[rows_b]^T X0[rows_b]  ([B, K, N0]) and, when x1 is given, out1[b] = S[rows_b]^T X1[rows_b]  ([B, K, N1]) in one launch;
 * rows_b = [graph_ptr[b], graph_ptr[b + 1]).  K <= 128.  Meant for segments of up to a few thousand rows: a workgroup walks a whole
 * segment (tsgnn_ragged_tn_f32 cuts long segments into slabs instead). */
int tsgnn_ragged_tn_direct_f32(const float* s_mat, int64_t lds_, int K, const int* graph_ptr, int B, const float* x0, int64_t ldx0,
                               int N0, float* out0, const float* x1, int64_t ldx1, int N1, float* out1, tsgnn_stream_t stream) {
  return tsgnn_ragged_tn_direct_ro_f32(s_mat, lds_, K, graph_ptr, B, x0, ldx0, N0, out0, x1, ldx1, N1, out1, nullptr, 0, nullptr, 0, 0, 0,
                                       stream);
}

/* the same + the max readout of x0 over each segment's node slots (encoders.py:353; trap T5): ro_out [B, N0] (leading dimension
 * ro_ldo), ro_arg [B, N0] = winning row, from the values the product's workgroups hold anyway.  ghost_zero != 0: a segment shorter
 * than nmax also has ZERO rows behind the real ones (row n_real + slot for its padded slots — the masked embeddings of a packed
 * batch): the first of them takes part, as in tsgnn_readout_max_fwd_f32; ghost_zero = 0: the segment's rows are all its slots. */
int tsgnn_ragged_tn_direct_ro_f32(const float* s_mat, int64_t lds_, int K, const int* graph_ptr, int B, const float* x0, int64_t ldx0,
                                  int N0, float* out0, const float* x1, int64_t ldx1, int N1, float* out1, float* ro_out,
                                  int64_t ro_ldo, int* ro_arg, int nmax, int64_t n_real, int ghost_zero, tsgnn_stream_t stream) {
  if ((ro_out == nullptr) != (ro_arg == nullptr) || (ro_out && (ro_ldo < N0 || nmax <= 0 || n_real < 0))) return TSGNN_EINVAL;
  if (!s_mat || !graph_ptr || !x0 || !out0 || K <= 0 || B <= 0 || N0 <= 0 || lds_ < K || ldx0 < N0) return TSGNN_EINVAL;
  if (x1 && (!out1 || N1 <= 0 || ldx1 < N1)) return TSGNN_EINVAL;
  if (K > 128 || B > 65535) return TSGNN_EUNSUPPORTED;
  RaggedTn a{};
  a.s = s_mat; a.lds_ = lds_; a.K = K; a.graph_ptr = graph_ptr;
  a.x[0] = x0; a.ldx[0] = ldx0; a.N[0] = N0; a.out[0] = out0;
  a.x[1] = x1 ? x1 : x0; a.ldx[1] = x1 ? ldx1 : ldx0; a.N[1] = x1 ? N1 : N0; a.out[1] = x1 ? out1 : out0;
  a.nt0 = (N0 + 31) / 32;
  a.ro_out = ro_out; a.ro_ldo = ro_ldo; a.ro_arg = ro_arg; a.ro_nmax = nmax; a.ro_n_real = n_real; a.ro_ghost = ghost_zero;
  const int nt = a.nt0 + (x1 ? (N1 + 31) / 32 : 0);
  const int mt = (K + 31) / 32;
  TSGNN_KNAME("ragged_tn_direct_kernel");
  ragged_tn_direct_kernel<<<dim3((unsigned)(nt * mt), (unsigned)B), 64 * RT_WAVES, 0, stream>>>(a, mt);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
