// Row-panel fp32 MFMA GEMM for the GraphConv transform and its input gradient:
//
//   C[R, N] = A[R, K] . B[K, N]            B = W            (forward,  encoders.py:36)
//   C[R, N] = A[R, K] . B^T (B is [N, K])  B = W            (dZ = dU . W^T)
//   optional epilogue: + bias, row L2 normalise (F.normalize eps 1e-12, encoders.py:38-40), rinv out
//
// R is the number of graph rows (thousands to millions), K and N are feature widths (<= 256).
// One 256-thread block owns 32 rows x all N columns.  K is consumed in 32-wide chunks staged through
// LDS with 16-byte global loads and a register prefetch of the next chunk (loads fly under the MFMAs).
// A fragments are read as ds_read_b128 from a [32][K+4] image: the K order inside an MFMA group is
// permuted (half h of the wave takes k = 8u+4h+c) so one 16-byte read feeds four v_mfma_f32_32x32x2_f32;
// with the +4 padding every 16-lane read group touches 64 distinct banks.  B fragments are ds_read_b32
// of 32 consecutive floats (conflict-free).  Exact fp32 (k-ordered fma chains).
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 32;
constexpr int LDA_S = KC + 4;
constexpr float NORM_EPS = 1e-12f;

struct RowGemmArgs {
  const float* a; int64_t lda;
  const float* b; int64_t ldb;        // B[K][N] row-major, or (TRANS_B) W[N][K] row-major
  const float* bias;
  float* c; int64_t ldc;
  float* rinv;
  int64_t rows; int K; int N;
  int normalize;
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int NT, bool TRANS_B>
__global__ __launch_bounds__(256) void rowgemm_kernel(RowGemmArgs g) {
  constexpr int NP = 32 * NT;
  constexpr int TPW = (NT + 3) / 4;
  constexpr int LDB_S = TRANS_B ? NP + 1 : NP;
  constexpr int BV = (KC * NP) / (256 * 4);            // float4 of B per thread per chunk (= NT)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                    // [32][LDA_S]
  float* Bs = smem + 32 * LDA_S;                       // [KC][LDB_S]
  float* Cs = smem;                                    // epilogue tile [32][NP+1], aliases As/Bs
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * 32;

  // staging maps ------------------------------------------------------------------------------
  const int am = tid >> 3, ak4 = tid & 7;              // A: row am, floats 4*ak4..+3 of the chunk
  const int64_t arow = m0 + am;
  float4 ra;
  float4 rb[BV];
  // loads are unconditional (clamped addresses) so nothing waits on them before the MFMAs; validity
  // masks are applied when the registers are written to LDS.
  int a_valid = 0;                                     // number of valid floats of ra (0..4)
  unsigned b_valid = 0;                                // bit q: rb[q] valid
  auto load_chunk = [&](int k0) {
    const int gk = k0 + 4 * ak4;
    const bool ok = arow < g.rows && gk < g.K;
    a_valid = ok ? min(4, g.K - gk) : 0;
    ra = ldg4(ok ? g.a + arow * g.lda + gk : g.a);
    b_valid = 0;
#pragma unroll
    for (int q = 0; q < BV; ++q) {
      const int idx = q * 256 + tid;
      bool okb;
      const float* p;
      if (!TRANS_B) {
        const int k = idx / (NP / 4), n4 = idx % (NP / 4);
        okb = (k0 + k) < g.K && 4 * n4 < g.N;          // N % 4 == 0 on this path
        p = g.b + (int64_t)(k0 + k) * g.ldb + 4 * n4;
      } else {
        const int n = idx / (KC / 4), k4 = idx % (KC / 4);
        okb = n < g.N && (k0 + 4 * k4) < g.K;          // K % 4 == 0 on this path
        p = g.b + (int64_t)n * g.ldb + k0 + 4 * k4;
      }
      rb[q] = ldg4(okb ? p : g.b);
      b_valid |= okb ? (1u << q) : 0u;
    }
  };
  auto store_chunk = [&]() {
    float4 va = ra;
    if (a_valid < 4) va.w = 0.f;
    if (a_valid < 3) va.z = 0.f;
    if (a_valid < 2) va.y = 0.f;
    if (a_valid < 1) va.x = 0.f;
    *reinterpret_cast<float4*>(As + am * LDA_S + 4 * ak4) = va;
#pragma unroll
    for (int q = 0; q < BV; ++q) {
      const int idx = q * 256 + tid;
      const float4 vb = ((b_valid >> q) & 1u) ? rb[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!TRANS_B) {
        const int k = idx / (NP / 4), n4 = idx % (NP / 4);
        *reinterpret_cast<float4*>(Bs + k * LDB_S + 4 * n4) = vb;
      } else {
        const int n = idx / (KC / 4), k4 = idx % (KC / 4);
        Bs[(4 * k4 + 0) * LDB_S + n] = vb.x;
        Bs[(4 * k4 + 1) * LDB_S + n] = vb.y;
        Bs[(4 * k4 + 2) * LDB_S + n] = vb.z;
        Bs[(4 * k4 + 3) * LDB_S + n] = vb.w;
      }
    }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int i = lane & 31, h = lane >> 5;
  load_chunk(0);
  store_chunk();
  __syncthreads();
  for (int k0 = 0; k0 < g.K; k0 += KC) {
    const bool more = (k0 + KC) < g.K;
    if (more) load_chunk(k0 + KC);                     // in flight under the MFMAs below
#pragma unroll
    for (int u = 0; u < KC / 8; ++u) {
      const float4 af = *reinterpret_cast<const float4*>(As + i * LDA_S + 8 * u + 4 * h);
      const float av[4] = {af.x, af.y, af.z, af.w};
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tile = wid + 4 * t;
        if (4 * (t + 1) <= NT || tile < NT) {           // compile-time true for full groups of 4 tiles
          const float* bp = Bs + (8 * u + 4 * h) * LDB_S + tile * 32 + i;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bp[c * LDB_S], acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  if (!g.normalize && !g.bias) {
    // plain product: accumulators straight to global (32 lanes = 128 contiguous bytes per row)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wid + 4 * t;
      if (tile < NT) {
        const int cn = tile * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < g.rows && cn < g.N) g.c[gm * g.ldc + cn] = acc[t][r];
        }
      }
    }
    return;
  }
  constexpr int LDC_S = NP + 1;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wid + 4 * t;
    if (tile < NT) {
      const int cn = tile * 32 + (lane & 31);
      const float bv = (g.bias && cn < g.N) ? g.bias[cn] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cm = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Cs[cm * LDC_S + cn] = acc[t][r] + bv;
      }
    }
  }
  __syncthreads();
  for (int m = wid; m < 32; m += 4) {
    const int64_t gm = m0 + m;
    if (gm >= g.rows) break;
    float u[(NP + 63) / 64];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < (NP + 63) / 64; ++j) {
      const int c = lane + 64 * j;
      u[j] = (c < g.N) ? Cs[m * LDC_S + c] : 0.f;
      ss = fmaf(u[j], u[j], ss);
    }
    float denom = 1.f;
    if (g.normalize) {
      ss = wave_sum(ss);
      denom = fmaxf(sqrtf(ss), NORM_EPS);
    }
#pragma unroll
    for (int j = 0; j < (NP + 63) / 64; ++j) {
      const int c = lane + 64 * j;
      if (c < g.N) g.c[gm * g.ldc + c] = g.normalize ? u[j] / denom : u[j];
    }
    if (g.rinv && lane == 0) g.rinv[gm] = 1.0f / denom;
  }
}



template <int NT, bool TRANS_B>
void launch_rowgemm(const RowGemmArgs& g, hipStream_t s) {
  constexpr int NP = 32 * NT;
  constexpr int LDB_S = TRANS_B ? NP + 1 : NP;
  const size_t ab = 32 * LDA_S + KC * LDB_S, c = 32 * (NP + 1);
  const size_t lds = sizeof(float) * (ab > c ? ab : c);
  rowgemm_kernel<NT, TRANS_B><<<(unsigned)ceil_div64(g.rows, 32), 256, lds, s>>>(g);
}

template <bool TRANS_B>
void dispatch_rowgemm(const RowGemmArgs& g, hipStream_t s) {
  switch ((g.N + 31) / 32) {
    case 1: launch_rowgemm<1, TRANS_B>(g, s); break;
    case 2: launch_rowgemm<2, TRANS_B>(g, s); break;
    case 3: launch_rowgemm<3, TRANS_B>(g, s); break;
    case 4: launch_rowgemm<4, TRANS_B>(g, s); break;
    case 5: launch_rowgemm<5, TRANS_B>(g, s); break;
    case 6: launch_rowgemm<6, TRANS_B>(g, s); break;
    case 7: launch_rowgemm<7, TRANS_B>(g, s); break;
    default: launch_rowgemm<8, TRANS_B>(g, s); break;
  }
}

}  // namespace

extern "C" {

/* 1 if tsgnn_rowgemm_f32 accepts these operands (16-byte rows / pointers, widths <= 256) */
int tsgnn_rowgemm_supported(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c, int64_t ldc, int K, int N,
                            int trans_b) {
  const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  if (!al || (lda % 4) || (ldb % 4) || N > 256 || N <= 0 || K <= 0) return 0;
  if (lda < ((K + 3) / 4) * 4) return 0;
  if (!trans_b && (N % 4)) return 0;
  if (trans_b && (K % 4)) return 0;
  (void)c; (void)ldc;
  return 1;
}

int tsgnn_rowgemm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int trans_b, const float* bias, float* c,
                      int64_t ldc, float* rinv, int64_t rows, int K, int N, int normalize, tsgnn_stream_t stream) {
  if (!a || !b || !c || rows < 0 || K <= 0 || N <= 0 || lda < K || ldc < N) return TSGNN_EINVAL;
  if (!tsgnn_rowgemm_supported(a, lda, b, ldb, c, ldc, K, N, trans_b)) return TSGNN_EUNSUPPORTED;
  if (rows == 0) return TSGNN_OK;
  RowGemmArgs g{a, lda, b, ldb, bias, c, ldc, rinv, rows, K, N, normalize};
  if (trans_b) dispatch_rowgemm<true>(g, stream);
  else dispatch_rowgemm<false>(g, stream);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
