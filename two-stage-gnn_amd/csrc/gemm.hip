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
#include "../../include/tsgnn.h"

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
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    // stage A tile [BM][BK] and B tile [BK][BN]; thread mapping follows the unit stride
#pragma unroll
    for (int it = 0; it < (BM * BK) / 256; ++it) {
      const int idx = it * 256 + tid;
      int m, k;
      if (a_k_fast) { m = idx / BK; k = idx % BK; } else { k = idx / BM; m = idx % BM; }
      const int gm = m0 + m, gk = k0 + k;
      As[m * LDA_S + k] = (gm < M && gk < kend) ? A[(int64_t)gm * g.sam + (int64_t)gk * g.sak] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < (BK * BN) / 256; ++it) {
      const int idx = it * 256 + tid;
      int k, n;
      if (b_n_fast) { k = idx / BN; n = idx % BN; } else { n = idx / BK; k = idx % BK; }
      const int gk = k0 + k, gn = n0 + n;
      Bs[k * LDB_S + n] = (gk < kend && gn < g.N) ? B[(int64_t)gk * g.sbk + (int64_t)gn * g.sbn] : 0.f;
    }
    __syncthreads();
    const int i = lane & 31, h = lane >> 5;
    const float* ap = As + (wr * 32 + i) * LDA_S + h;
    const float* bp = Bs + h * LDB_S + wc * 32 + i;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk * LDB_S], acc, 0, 0, 0);
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
  for (int k = 0; k < nsplit; ++k) s += slabs[(int64_t)k * slab_stride + i];
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
  if (f < F) for (int64_t r = r0 + rl; r < r1; r += 4) s += x[r * ld + f];
  lds[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && f < F) part[(int64_t)blockIdx.y * F + f] = lds[0][threadIdx.x] + lds[1][threadIdx.x] + lds[2][threadIdx.x] + lds[3][threadIdx.x];
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

/* out[f] (+)= sum_r x[r,f];  ws >= nchunk*F floats with nchunk = ceil(rows/512) */
int tsgnn_colsum_f32(const float* x, int64_t ld, int64_t rows, int F, float* out, float* ws, int accumulate,
                     tsgnn_stream_t stream) {
  if (!x || !out || !ws || rows < 0 || F <= 0 || ld < F) return TSGNN_EINVAL;
  const int64_t rpc = 512;
  const int nchunk = (int)(rows > 0 ? ceil_div64(rows, rpc) : 1);
  dim3 grid((unsigned)((F + 63) / 64), (unsigned)nchunk);
  colsum_partial<<<grid, 256, 0, stream>>>(x, ld, rows, F, rpc, ws);
  splitk_reduce_kernel<<<(unsigned)ceil_div64(F, 256), 256, 0, stream>>>(ws, nchunk, F, F, out, accumulate);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
