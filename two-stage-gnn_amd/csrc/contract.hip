// DiffPool contraction of a POOLED level (encoders.py:374-375):  X'[b] = S_b^T Z_b,  A'[b] = S_b^T A_b S_b
// on dense per-graph operands that are tiny (DD cfg 5, second pooling: 64 nodes, 8 clusters, 192 features).
//
// As batched GEMM launches this is 3 launches forward and 6 backward of ~7.6 us each for 0.4 MFLOP per graph — launch latency
// only.  Here ONE workgroup per graph keeps S, Z, A (and T = S^T A) in LDS and produces both outputs; the backward produces
// dS, dZ, dA from (dX', dA') the same way:
//     dZ = S dX' ;  dT = dA' S^T ;  dS = Z dX'^T + T^T dA' + A dT^T ;  dA = S dT
// LDS rows are padded by one float, so that lanes that walk different rows of one operand hit different banks.
// Sums run over the contracted index in ascending order with fma: fixed order, bitwise reproducible.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

#ifndef TSGNN_CT_THREADS
#define TSGNN_CT_THREADS 512
#endif
constexpr int CT_THREADS = TSGNN_CT_THREADS;

struct CtDims {
  int N, K, F;              // nodes, clusters, features
  int ldS, ldZ, ldA, ldT;   // padded LDS row lengths: K+1, F+1, N+1, N+1 (operands whose ROWS the lanes walk: odd strides)
  int ldDX, ldDT;           // backward: dX' [K][F], dT [K][N] are read as 16-byte fragments along a row: unpadded, 16-byte aligned
};
__host__ __device__ inline int ct_up4(int n) { return (n + 3) & ~3; }
__host__ __device__ inline CtDims ct_dims(int N, int K, int F) { return CtDims{N, K, F, K + 1, F + 1, N + 1, N + 1, ct_up4(F), ct_up4(N)}; }
// floats of LDS: S, Z, A, T (+ backward: dX', dA', dT); every region starts on a 16-byte boundary
inline size_t ct_lds_floats(int N, int K, int F, bool bwd) {
  size_t n = (size_t)ct_up4(N * (K + 1)) + (size_t)ct_up4(N * (F + 1)) + (size_t)ct_up4(N * (N + 1)) + (size_t)ct_up4(K * (N + 1));
  if (bwd) n += (size_t)K * ct_up4(F) + (size_t)ct_up4(K * (K + 1)) + (size_t)K * ct_up4(N) + (size_t)ct_up4(N * K);   // (+ dS rows: softmax)
  return n;
}
// x / d by a float reciprocal and one correction step (exact for x < 2^22, d <= 4096): the element loops below computed a
// (row, column) pair per output with a ~35-instruction integer division
struct CtDiv {
  float inv; int d;
  __device__ __forceinline__ explicit CtDiv(int d_) : inv(1.0f / (float)d_), d(d_) {}
  __device__ __forceinline__ int operator()(int x) const {
    int q = (int)(((float)x + 0.5f) * inv);
    const int r = x - q * d;
    q += (r >= d) - (r < 0);
    return q;
  }
};

// rows x cols contiguous floats -> LDS rows of stride ld.  Eight requests in flight per thread before the first LDS store (a
// load-then-store loop is one dependent L2 round trip per iteration: 24 trips for a 64 x 192 operand).
__device__ __forceinline__ void ct_load(float* dst, int ld, const float* __restrict__ src, int rows, int cols) {
  const int total = rows * cols;
  for (int base = threadIdx.x; base < total; base += 8 * CT_THREADS) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = base + u * CT_THREADS; v[u] = i < total ? src[i] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * CT_THREADS;
      if (i < total) { const int r = i / cols, c = i - r * cols; dst[r * ld + c] = v[u]; }
    }
  }
}

// All operands of a kernel in ONE round trip: the operands are taken as one sequence of 16-byte elements, every thread requests
// up to sixteen of them before anything is stored (ct_load per operand: one dependent round trip each — three forward, six
// backward — with a 4-byte request and an integer division per float).  cols % 4 == 0, sources 16-byte aligned (ct_vec_ok).
struct CtOp { const float* src; float* dst; int ld, cols, n4; };
template <int NOPS>
__device__ __forceinline__ void ct_load_many(const CtOp (&op)[NOPS]) {
  int start[NOPS + 1];
  float inv[NOPS];
  start[0] = 0;
#pragma unroll
  for (int i = 0; i < NOPS; ++i) { start[i + 1] = start[i] + op[i].n4; inv[i] = 1.0f / (float)op[i].cols; }
  const int total = start[NOPS];
  for (int base = threadIdx.x; base < total; base += 16 * CT_THREADS) {
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = min(base + u * CT_THREADS, total - 1);
      const float* src = op[0].src;
      int l = e;
#pragma unroll
      for (int i = 1; i < NOPS; ++i)
        if (e >= start[i]) { src = op[i].src; l = e - start[i]; }
      v[u] = reinterpret_cast<const float4*>(src)[l];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = base + u * CT_THREADS;
      if (e < total) {
        float* dst = op[0].dst;
        int l = e, ld = op[0].ld, cols = op[0].cols;
        float iv = inv[0];
#pragma unroll
        for (int i = 1; i < NOPS; ++i)
          if (e >= start[i]) { dst = op[i].dst; l = e - start[i]; ld = op[i].ld; cols = op[i].cols; iv = inv[i]; }
        const int x = 4 * l;
        int r = (int)(((float)x + 0.5f) * iv);              // x / cols by reciprocal + one correction (exact for x < 2^22)
        int c = x - r * cols;
        if (c < 0) { --r; c += cols; } else if (c >= cols) { ++r; c -= cols; }
        float* d = dst + r * ld + c;                        // (rows are padded by one float: four 4-byte stores)
        d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
      }
    }
  }
}
__device__ __forceinline__ bool ct_vec_ok(const void* p, int cols) { return (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ __launch_bounds__(CT_THREADS) void contract_dense_fwd_kernel(const float* __restrict__ s, const float* __restrict__ z,
                                                                        const float* __restrict__ adj, int N, int K, int F,
                                                                        float* __restrict__ xo, float* __restrict__ ao,
                                                                        float* __restrict__ t_out, float* __restrict__ ro_out,
                                                                        int64_t ro_ldo, int* __restrict__ ro_arg,
                                                                        float* __restrict__ s_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const CtDims d = ct_dims(N, K, F);
  float* S = sm;
  float* Z = S + ct_up4(N * d.ldS);
  float* A = Z + ct_up4(N * d.ldZ);
  float* T = A + ct_up4(N * d.ldA);
  const int b = blockIdx.x;
  const float* sb = s + (int64_t)b * N * K;
  const float* zb = z + (int64_t)b * N * F;
  const float* ab = adj + (int64_t)b * N * N;
  if (ct_vec_ok(sb, K) && ct_vec_ok(zb, F) && ct_vec_ok(ab, N)) {
    const CtOp ops[3] = {{sb, S, d.ldS, K, N * K / 4}, {zb, Z, d.ldZ, F, N * F / 4}, {ab, A, d.ldA, N, N * N / 4}};
    ct_load_many<3>(ops);
  } else {
    ct_load(S, d.ldS, sb, N, K);
    ct_load(Z, d.ldZ, zb, N, F);
    ct_load(A, d.ldA, ab, N, N);
  }
  __syncthreads();
  if (s_out) {
    // `s` holds the assignment LOGITS: S = softmax over the K clusters of every row (nn.Softmax(dim=-1), encoders.py:369), kept
    // for the backward and the caller (a launch of its own otherwise: 16 x 64 rows of 8 numbers)
    for (int n = threadIdx.x; n < N; n += CT_THREADS) {
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmaxf(m, S[n * d.ldS + k]);
      float den = 0.f;
      for (int k = 0; k < K; ++k) den += expf(S[n * d.ldS + k] - m);
      for (int k = 0; k < K; ++k) {
        const float v = expf(S[n * d.ldS + k] - m) / den;
        S[n * d.ldS + k] = v;
        s_out[(int64_t)b * N * K + (int64_t)n * K + k] = v;
      }
    }
    __syncthreads();
  }
  if (ro_out) {
    // max readout of z over the graph's N rows (encoders.py:383), rows b * N + n: the same packed (value, row) order as
    // readout_max_direct (bn_readout.hip) — ties go to the smallest row
    for (int f = threadIdx.x; f < F; f += CT_THREADS) {
      unsigned long long best = 0ull;
      for (int n = 0; n < N; ++n) {
        const unsigned long long p = ((unsigned long long)f32_ordered(Z[n * d.ldZ + f]) << 32) |
                                     (unsigned long long)(0xFFFFFFFFu - (unsigned)(b * N + n));
        best = p > best ? p : best;
      }
      ro_out[(int64_t)b * ro_ldo + f] = best ? ordered_f32((unsigned)(best >> 32)) : 0.f;
      ro_arg[(int64_t)b * F + f] = best ? (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull)) : -1;
    }
  }
  // X' = S^T Z   [K, F]
  const CtDiv byF(F), byN(N), byK(K);
  for (int i = threadIdx.x; i < K * F; i += CT_THREADS) {
    const int k = byF(i), f = i - k * F;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(S[n * d.ldS + k], Z[n * d.ldZ + f], acc);
    xo[(int64_t)b * K * F + i] = acc;
  }
  // T = S^T A   [K, N]
  for (int i = threadIdx.x; i < K * N; i += CT_THREADS) {
    const int k = byN(i), m = i - k * N;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(S[n * d.ldS + k], A[n * d.ldA + m], acc);
    T[k * d.ldT + m] = acc;
    t_out[(int64_t)b * K * N + i] = acc;
  }
  __syncthreads();
  // A' = T S    [K, K]
  for (int i = threadIdx.x; i < K * K; i += CT_THREADS) {
    const int k = byK(i), l = i - k * K;
    float acc = 0.f;
    for (int m = 0; m < N; ++m) acc = fmaf(T[k * d.ldT + m], S[m * d.ldS + l], acc);
    ao[(int64_t)b * K * K + i] = acc;
  }
}

__global__ __launch_bounds__(CT_THREADS) void contract_dense_bwd_kernel(const float* __restrict__ s, const float* __restrict__ z,
                                                                        const float* __restrict__ adj, const float* __restrict__ t,
                                                                        const float* __restrict__ dxo, const float* __restrict__ dao,
                                                                        int N, int K, int F, float* __restrict__ ds,
                                                                        float* __restrict__ dz, float* __restrict__ dadj,
                                                                        const float* __restrict__ ro_dout, int64_t ro_ldo,
                                                                        const int* __restrict__ ro_arg, int softmax) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const CtDims d = ct_dims(N, K, F);
  float* S = sm;
  float* Z = S + ct_up4(N * d.ldS);
  float* A = Z + ct_up4(N * d.ldZ);
  float* T = A + ct_up4(N * d.ldA);
  float* DX = T + ct_up4(K * d.ldT);   // [K][up4(F)]
  float* DA = DX + K * d.ldDX;         // [K][K+1]
  float* DT = DA + ct_up4(K * d.ldS);  // [K][up4(N)]
  float* DSL = DT + K * d.ldDT;        // [N][K]  (softmax: the rows of dS meet here)
  const int b = blockIdx.x;
  const float* sb = s + (int64_t)b * N * K;
  const float* dxb = dxo + (int64_t)b * K * F;
  const float* dab = dao + (int64_t)b * K * K;
  const float* zb = ds ? z + (int64_t)b * N * F : nullptr;
  const float* ab = ds ? adj + (int64_t)b * N * N : nullptr;
  const float* tb = ds ? t + (int64_t)b * K * N : nullptr;
  const bool vec = ct_vec_ok(sb, K) && ct_vec_ok(dxb, F) && ct_vec_ok(dab, K) && ct_vec_ok(zb, F) && ct_vec_ok(ab, N) && ct_vec_ok(tb, N);
  if (vec && ds) {
    const CtOp ops[6] = {{sb, S, d.ldS, K, N * K / 4}, {dxb, DX, d.ldDX, F, K * F / 4}, {dab, DA, d.ldS, K, K * K / 4},
                         {zb, Z, d.ldZ, F, N * F / 4}, {ab, A, d.ldA, N, N * N / 4}, {tb, T, d.ldT, N, K * N / 4}};
    ct_load_many<6>(ops);
  } else if (vec) {
    const CtOp ops[3] = {{sb, S, d.ldS, K, N * K / 4}, {dxb, DX, d.ldDX, F, K * F / 4}, {dab, DA, d.ldS, K, K * K / 4}};
    ct_load_many<3>(ops);
  } else {
    ct_load(S, d.ldS, sb, N, K);
    ct_load(DX, d.ldDX, dxb, K, F);
    ct_load(DA, d.ldS, dab, K, K);
    if (ds) {
      ct_load(Z, d.ldZ, zb, N, F);
      ct_load(A, d.ldA, ab, N, N);
      ct_load(T, d.ldT, tb, K, N);
    }
  }
  __syncthreads();
  const CtDiv byN(N);
  const bool v4 = (F & 3) == 0 && (N & 3) == 0;            // outputs as 16-byte fragments (dX' and dT rows are 16-byte aligned in LDS)
  // dT = dA' S^T   [K, N]
  for (int i = threadIdx.x; i < K * N; i += CT_THREADS) {
    const int k = byN(i), m = i - k * N;
    float acc = 0.f;
    for (int l = 0; l < K; ++l) acc = fmaf(DA[k * d.ldS + l], S[m * d.ldS + l], acc);
    DT[k * d.ldDT + m] = acc;
  }
  // dZ = S dX'     [N, F]
  if (dz && v4) {
    const int F4 = F >> 2;
    const CtDiv byF4(F4);
    for (int i = threadIdx.x; i < N * F4; i += CT_THREADS) {
      const int n = byF4(i), f4 = i - n * F4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ro_arg) {                                      // + the max readout's gradient where row b * N + n won its column
        const int4 ra = *reinterpret_cast<const int4*>(ro_arg + (int64_t)b * F + 4 * f4);
        const float4 rd = *reinterpret_cast<const float4*>(ro_dout + (int64_t)b * ro_ldo + 4 * f4);
        const int row = b * N + n;
        acc = make_float4(ra.x == row ? rd.x : 0.f, ra.y == row ? rd.y : 0.f, ra.z == row ? rd.z : 0.f, ra.w == row ? rd.w : 0.f);
      }
      for (int k = 0; k < K; ++k) {
        const float sv = S[n * d.ldS + k];
        const float4 x4 = *reinterpret_cast<const float4*>(DX + k * d.ldDX + 4 * f4);
        acc.x = fmaf(sv, x4.x, acc.x); acc.y = fmaf(sv, x4.y, acc.y); acc.z = fmaf(sv, x4.z, acc.z); acc.w = fmaf(sv, x4.w, acc.w);
      }
      *reinterpret_cast<float4*>(dz + (int64_t)b * N * F + 4 * (int64_t)i) = acc;
    }
  } else if (dz) {
    const CtDiv byF(F);
    for (int i = threadIdx.x; i < N * F; i += CT_THREADS) {
      const int n = byF(i), f = i - n * F;
      float acc = (ro_arg && ro_arg[(int64_t)b * F + f] == b * N + n) ? ro_dout[(int64_t)b * ro_ldo + f] : 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(S[n * d.ldS + k], DX[k * d.ldDX + f], acc);
      dz[(int64_t)b * N * F + i] = acc;
    }
  }
  __syncthreads();
  // dS = Z dX'^T + T^T dA' + A dT^T   [N, K]   (the three terms in this order)
  if (ds) {
    for (int i = threadIdx.x; i < N * K; i += CT_THREADS) {
      const int k = byN(i), n = i - k * N;                 // n fastest: lanes walk rows of Z / A (padded: conflict-free)
      float acc = 0.f;
      for (int f = 0; f < F; ++f) acc = fmaf(Z[n * d.ldZ + f], DX[k * d.ldDX + f], acc);
      for (int l = 0; l < K; ++l) acc = fmaf(T[l * d.ldT + n], DA[l * d.ldS + k], acc);
      for (int m = 0; m < N; ++m) acc = fmaf(A[n * d.ldA + m], DT[k * d.ldDT + m], acc);
      if (softmax) DSL[n * K + k] = acc;
      else ds[(int64_t)b * N * K + (int64_t)n * K + k] = acc;
    }
    if (softmax) {
      // S = softmax(logits): d logits[n, k] = S[n, k] * (dS[n, k] - sum_l S[n, l] dS[n, l])   (row_softmax_bwd, pooling.hip)
      __syncthreads();
      const CtDiv byK(K);
      for (int i = threadIdx.x; i < N * K; i += CT_THREADS) {
        const int n = byK(i), k = i - n * K;
        float dot = 0.f;
        for (int l = 0; l < K; ++l) dot = fmaf(S[n * d.ldS + l], DSL[n * K + l], dot);
        ds[(int64_t)b * N * K + i] = S[n * d.ldS + k] * (DSL[i] - dot);
      }
    }
  }
  // dA = S dT      [N, N]
  if (dadj && v4) {
    const int N4 = N >> 2;
    const CtDiv byN4(N4);
    for (int i = threadIdx.x; i < N * N4; i += CT_THREADS) {
      const int n = byN4(i), m4 = i - n * N4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < K; ++k) {
        const float sv = S[n * d.ldS + k];
        const float4 t4 = *reinterpret_cast<const float4*>(DT + k * d.ldDT + 4 * m4);
        acc.x = fmaf(sv, t4.x, acc.x); acc.y = fmaf(sv, t4.y, acc.y); acc.z = fmaf(sv, t4.z, acc.z); acc.w = fmaf(sv, t4.w, acc.w);
      }
      *reinterpret_cast<float4*>(dadj + (int64_t)b * N * N + 4 * (int64_t)i) = acc;
    }
  } else if (dadj) {
    for (int i = threadIdx.x; i < N * N; i += CT_THREADS) {
      const int n = byN(i), m = i - n * N;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(S[n * d.ldS + k], DT[k * d.ldDT + m], acc);
      dadj[(int64_t)b * N * N + i] = acc;
    }
  }
}

constexpr size_t CT_LDS_MAX = 150 * 1024;

// ---------------------------------------------------------------- backward of the level-1 (row-layout) contraction
// X'[b] = S_b^T Z_b, A'[b] = S_b^T (A S)_b over the real rows of graph b (diffpool.py::_ContractRows).  Its backward is three
// row-ragged products with the per-graph K x F / K x K gradients,
//     dZ_b = S_b dX'_b ;  dS_b = Z_b dX'_b^T + (AS)_b dA'_b^T ;  d(AS)_b = S_b dA'_b ,
// four batched-GEMM launches + three zero fills before.  One workgroup per slab of <= 32 rows of one graph holds dX'_b, dA'_b and
// the slab's rows of S, Z, AS in LDS (odd row strides: both the row-wise and the transposed fragment reads are conflict-free)
// and runs the 32 x 32 output tiles as fp32 MFMA jobs (v_mfma_f32_32x32x2_f32); the host deals the jobs to the four waves by
// cost (longest first), e.g. K = 64, F = 192: two dS tiles of depth 256, six dZ and two d(AS) tiles of depth 64 -> 128 MFMAs per
// wave.  (A first version with scalar FMAs out of LDS took longer than the launches it replaced.)
// Rows outside every graph (ghost rows) are zeroed by the trailing blocks.
typedef float ct_f32x16 __attribute__((ext_vector_type(16)));
constexpr int CR_KMAX = 64, CR_FMAX = 256, CR_MAXJOBS = 8;
struct CtRows {
  const float* S; int64_t ldS; const float* Z; int64_t ldZ; const float* AS; int64_t ldAS;
  const float* dxo; const float* dao;            // [B, K, F], [B, K, K]
  const int* slab_row_ptr; const int* slab_graph; int nslab;
  int K, F;
  float* dZ; int64_t lddZ; float* dS; int64_t lddS; float* dAS; int64_t lddAS;
  int64_t zero_from, zero_to;                    // rows [zero_from, zero_to) of the three outputs are cleared
  const float* ro_dout; int64_t ro_ldo; const int* ro_arg;   // nullable: dZ += the max readout's gradient (dout [B, F], arg = winning row)
  signed char job[4][CR_MAXJOBS];                // per wave: job codes, -1 ends.  0..7: dZ tile t; 8..9: d(AS) tile; 10..11: dS tile
};

// acc += A[32 x depth] . B[depth x 32]:  A[i][k] = ap[i * ald + k];  B[k][n] = bp[k * bks + n * bns], columns n >= nvalid are zero
__device__ __forceinline__ void ct_mfma(ct_f32x16& acc, const float* ap, int ald, const float* bp, int bks, int bns, int depth, int nvalid) {
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const bool nok = i < nvalid;
  const float* arow = ap + i * ald + h;
  const float* bcol = bp + (nok ? i : 0) * bns + h * bks;
  // eight MFMA steps per round; the operands of round n + 1 are requested before the chain of round n issues (one wave per SIMD:
  // nothing else hides the LDS latency).  Reads are unconditional from clamped addresses and masked afterwards (a predicated
  // read would be waited for on its own).
  auto fetch = [&](int s0, float (&av)[8], float (&bv)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = s0 + 2 * u, kc = k < depth ? k : 0;
      const float ta = arow[kc], tb = bcol[kc * bks];
      av[u] = k < depth ? ta : 0.f;
      bv[u] = (k < depth && nok) ? tb : 0.f;
    }
  };
  float a0[8], b0[8], a1[8], b1[8];
  if ((depth & 31) == 0 && nvalid >= 32) {
    // the common shapes (depth 64 / 192 / 256, full column tiles): a branch-free body of sixteen MFMAs whose reads carry no
    // predicates — one basic block per iteration, so the scheduler keeps the reads of the next round in flight under the chain
    auto load8 = [&](int s0, float (&av)[8], float (&bv)[8]) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { av[u] = arow[s0 + 2 * u]; bv[u] = bcol[(s0 + 2 * u) * bks]; }
    };
    load8(0, a0, b0);
    for (int s0 = 0; s0 < depth; s0 += 32) {
      load8(s0 + 16, a1, b1);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
      load8(s0 + 32 < depth ? s0 + 32 : 0, a0, b0);       // (last round: a harmless re-read)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);
    }
    return;
  }
  fetch(0, a0, b0);
  for (int s0 = 0; s0 < depth; s0 += 32) {
    if (s0 + 16 < depth) fetch(s0 + 16, a1, b1);
    __builtin_amdgcn_sched_barrier(0);                     // (the scheduler would sink the reads below the MFMA chain)
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
    if (s0 + 16 < depth) {
      if (s0 + 32 < depth) fetch(s0 + 32, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256) void contract_rows_bwd_kernel(CtRows a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.K, F = a.F, tid = threadIdx.x;
  if ((int)blockIdx.x >= a.nslab) {                      // ghost rows: zeros
    const int zb = (int)blockIdx.x - a.nslab, nzb = (int)gridDim.x - a.nslab;
    for (int64_t r = a.zero_from + zb; r < a.zero_to; r += nzb) {
      for (int c = tid; c < F; c += 256) a.dZ[r * a.lddZ + c] = 0.f;
      for (int c = tid; c < K; c += 256) { a.dS[r * a.lddS + c] = 0.f; a.dAS[r * a.lddAS + c] = 0.f; }
    }
    return;
  }
  TR(0);
  const int ldx = F + 1, lda = K + 1;
  float* DX = sm;                       // [K][F + 1]
  float* DA = DX + K * ldx;             // [K][K + 1]
  float* Sp = DA + K * lda;             // [32][K + 1]
  float* Zp = Sp + 32 * lda;            // [32][F + 1]
  float* Ap = Zp + 32 * ldx;            // [32][K + 1]
  const int r0 = a.slab_row_ptr[blockIdx.x], r1 = a.slab_row_ptr[blockIdx.x + 1];
  const int nr = r1 - r0;
  const int b = a.slab_graph[blockIdx.x];
  const int F4 = F / 4, K4 = K / 4;
  {
    // all five operands are requested before the first LDS store: ONE memory round trip for the block's 104 KB (five staged
    // calls were five dependent trips).  One wave per SIMD here, so the <= 32 float4 per thread are free registers.
    constexpr int NX = CR_KMAX * (CR_FMAX / 4) / 256, NA = CR_KMAX * (CR_KMAX / 4) / 256, NZ = 32 * (CR_FMAX / 4) / 256, NS = 32 * (CR_KMAX / 4) / 256;
    float4 vx[NX], va[NA], vz[NZ], vs[NS], vp[NS];
    const float* gx = a.dxo + (int64_t)b * K * F;
    const float* ga = a.dao + (int64_t)b * K * K;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NX; ++u) { const int i = tid + 256 * u; vx[u] = i < K * F4 ? *reinterpret_cast<const float4*>(gx + 4 * (int64_t)i) : zero4; }
#pragma unroll
    for (int u = 0; u < NA; ++u) { const int i = tid + 256 * u; va[u] = i < K * K4 ? *reinterpret_cast<const float4*>(ga + 4 * (int64_t)i) : zero4; }
#pragma unroll
    for (int u = 0; u < NZ; ++u) {
      const int i = tid + 256 * u, r = i / F4, c = i - r * F4;
      vz[u] = (i < 32 * F4 && r < nr) ? *reinterpret_cast<const float4*>(a.Z + (int64_t)(r0 + r) * a.ldZ + 4 * c) : zero4;
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int i = tid + 256 * u, r = i / K4, c = i - r * K4;
      const bool ok = i < 32 * K4 && r < nr;
      vs[u] = ok ? *reinterpret_cast<const float4*>(a.S + (int64_t)(r0 + r) * a.ldS + 4 * c) : zero4;
      vp[u] = ok ? *reinterpret_cast<const float4*>(a.AS + (int64_t)(r0 + r) * a.ldAS + 4 * c) : zero4;
    }
    auto put = [](float* d, const float4& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; };
#pragma unroll
    for (int u = 0; u < NX; ++u) { const int i = tid + 256 * u; if (i < K * F4) { const int k = i / F4, c = i - k * F4; put(DX + k * ldx + 4 * c, vx[u]); } }
#pragma unroll
    for (int u = 0; u < NA; ++u) { const int i = tid + 256 * u; if (i < K * K4) { const int k = i / K4, c = i - k * K4; put(DA + k * lda + 4 * c, va[u]); } }
#pragma unroll
    for (int u = 0; u < NZ; ++u) { const int i = tid + 256 * u; if (i < 32 * F4) { const int r = i / F4, c = i - r * F4; put(Zp + r * ldx + 4 * c, vz[u]); } }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int i = tid + 256 * u;
      if (i < 32 * K4) { const int r = i / K4, c = i - r * K4; put(Sp + r * lda + 4 * c, vs[u]); put(Ap + r * lda + 4 * c, vp[u]); }
    }
  }
  TR(1);
  __syncthreads();
  TR(2);
  const int wid = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
  // the readout's (winning row, gradient) of this lane's column in every dZ tile of the wave: requested NOW, with the operands, and
  // parked in LDS — requested per job they were a global round trip on every job's critical path, waited for behind the previous
  // job's sixteen stores in the same vmcnt counter (measured: 18 -> 24 us for the launch; 20 this way)
  float* rq = Ap + 32 * lda + wid * (CR_MAXJOBS * 64);   // [wave][job][(row, gradient) x 32 columns]
  if (a.ro_arg && h == 0) {
    for (int q = 0; q < CR_MAXJOBS; ++q) {
      const int code = a.job[wid][q];
      if (code < 0) break;
      if (code < 8) {
        const int cc = min(32 * code + i, F - 1);
        rq[q * 64 + i] = __int_as_float(a.ro_arg[(int64_t)b * F + cc]);
        rq[q * 64 + 32 + i] = a.ro_dout[(int64_t)b * a.ro_ldo + cc];
      }
    }
  }
  for (int q = 0; q < CR_MAXJOBS; ++q) {
    const int code = a.job[wid][q];                      // uniform over the wave
    if (code < 0) break;
    ct_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float* out;
    int64_t ldo;
    int t, ncols;
    int ra = -1;
    float rd = 0.f;
    if (code < 8) {                                      // dZ tile: S . dX'
      t = code; ncols = F; out = a.dZ; ldo = a.lddZ;
      if (a.ro_arg) { ra = __float_as_int(rq[q * 64 + i]); rd = rq[q * 64 + 32 + i]; }   // (written by this wave's own lanes: no barrier)
      ct_mfma(acc, Sp, lda, DX + 32 * t, ldx, 1, K, F - 32 * t);
    } else if (code < 10) {                              // d(AS) tile: S . dA'
      t = code - 8; ncols = K; out = a.dAS; ldo = a.lddAS;
      ct_mfma(acc, Sp, lda, DA + 32 * t, lda, 1, K, K - 32 * t);
    } else {                                             // dS tile: Z . dX'^T + AS . dA'^T
      t = code - 10; ncols = K; out = a.dS; ldo = a.lddS;
      ct_mfma(acc, Zp, ldx, DX + 32 * t * ldx, 1, ldx, F, K - 32 * t);
      ct_mfma(acc, Ap, lda, DA + 32 * t * lda, 1, lda, K, K - 32 * t);
    }
    TR_AFTER(__float_as_int(acc[0]), 3 + 2 * min(q, 3));
    const int c = 32 * t + i;
    if (c < ncols) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < nr) out[(int64_t)(r0 + m) * ldo + c] = acc[r] + ((r0 + m) == ra ? rd : 0.f);
      }
    }
    TR(4 + 2 * min(q, 3));
  }
  TR(11);
  TR_END();
}

inline size_t ct_rows_lds_floats(int K, int F) {
  return (size_t)K * (F + 1) + (size_t)K * (K + 1) + 32 * (size_t)(K + 1) * 2 + 32 * (size_t)(F + 1) + 4 * CR_MAXJOBS * 64;
}

// deal the tile jobs to the four waves: longest processing time first
inline void ct_rows_jobs(int K, int F, signed char (&job)[4][CR_MAXJOBS]) {
  int codes[12], cost[12], n = 0;
  for (int t = 0; t * 32 < K; ++t) { codes[n] = 10 + t; cost[n++] = F + K; }
  for (int t = 0; t * 32 < F; ++t) { codes[n] = t; cost[n++] = K; }
  for (int t = 0; t * 32 < K; ++t) { codes[n] = 8 + t; cost[n++] = K; }
  int load[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
  for (int w = 0; w < 4; ++w) for (int q = 0; q < CR_MAXJOBS; ++q) job[w][q] = -1;
  for (int j = 0; j < n; ++j) {                           // (codes are listed in non-increasing cost order already)
    int best = 0;
    for (int w = 1; w < 4; ++w) if (load[w] < load[best]) best = w;
    job[best][cnt[best]++] = (signed char)codes[j];
    load[best] += cost[j];
  }
}

}  // namespace

extern "C" {

/* 1 if the one-workgroup-per-graph contraction kernels take (N nodes, K clusters, F features): operands of a graph fit in LDS */
int tsgnn_contract_dense_supported(int N, int K, int F) {
  return (N > 0 && K > 0 && F > 0 && N <= 256 && K <= 256 && F <= 1024 && ct_lds_floats(N, K, F, true) * sizeof(float) <= CT_LDS_MAX) ? 1 : 0;
}

/* X'[b] = S_b^T Z_b (xo [B,K,F]), A'[b] = S_b^T A_b S_b (ao [B,K,K]), T = S^T A (t [B,K,N], kept for the backward);
 * s [B,N,K], z [B,N,F], adj [B,N,N] contiguous (encoders.py:374-375) */
int tsgnn_contract_dense_fwd_f32(const float* s, const float* z, const float* adj, int B, int N, int K, int F, float* xo, float* ao,
                                 float* t, tsgnn_stream_t stream) {
  return tsgnn_contract_dense_fwd_ro_f32(s, z, adj, B, N, K, F, xo, ao, t, nullptr, 0, nullptr, nullptr, stream);
}

/* the same + the max readout of z over each graph's N rows (encoders.py:383) out of the staged operand: ro_out [B, F] (leading
 * dimension ro_ldo), ro_arg [B, F] = winning row b * N + n — what tsgnn_readout_max_fwd_f32 returns for the uniform batch (B, N) */
int tsgnn_contract_dense_fwd_ro_f32(const float* s, const float* z, const float* adj, int B, int N, int K, int F, float* xo, float* ao,
                                    float* t, float* ro_out, int64_t ro_ldo, int* ro_arg, float* s_out, tsgnn_stream_t stream) {
  if (!s || !z || !adj || !xo || !ao || !t || B < 0) return TSGNN_EINVAL;
  if ((ro_out == nullptr) != (ro_arg == nullptr) || (ro_out && ro_ldo < F)) return TSGNN_EINVAL;
  if (!tsgnn_contract_dense_supported(N, K, F)) return TSGNN_EUNSUPPORTED;
  if (B == 0) return TSGNN_OK;
  const size_t lds = ct_lds_floats(N, K, F, false) * sizeof(float);
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(contract_dense_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT_LDS_MAX);
    attr = CT_LDS_MAX;
  }
  contract_dense_fwd_kernel<<<(unsigned)B, CT_THREADS, lds, stream>>>(s, z, adj, N, K, F, xo, ao, t, ro_out, ro_ldo, ro_arg, s_out);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* gradients of the pair above: ds [B,N,K], dz [B,N,F], dadj [B,N,N] (each nullable) from dxo [B,K,F], dao [B,K,K] */
int tsgnn_contract_dense_bwd_f32(const float* s, const float* z, const float* adj, const float* t, const float* dxo, const float* dao,
                                 int B, int N, int K, int F, float* ds, float* dz, float* dadj, tsgnn_stream_t stream) {
  return tsgnn_contract_dense_bwd_ro_f32(s, z, adj, t, dxo, dao, B, N, K, F, ds, dz, dadj, nullptr, 0, nullptr, 0, stream);
}

/* the same; dz additionally takes the gradient of the max readout of z over each graph's N rows (ro_dout [B, F] with leading
 * dimension ro_ldo, ro_arg [B, F] = winning row b * N + n or -1): the pass tsgnn_readout_max_bwd_rows_f32 would make over dz */
int tsgnn_contract_dense_bwd_ro_f32(const float* s, const float* z, const float* adj, const float* t, const float* dxo, const float* dao,
                                    int B, int N, int K, int F, float* ds, float* dz, float* dadj, const float* ro_dout, int64_t ro_ldo,
                                    const int* ro_arg, int softmax, tsgnn_stream_t stream) {
  if (!s || !dxo || !dao || B < 0 || (ds && (!z || !adj || !t))) return TSGNN_EINVAL;
  if ((ro_dout == nullptr) != (ro_arg == nullptr) || (ro_arg && (!dz || ro_ldo < F))) return TSGNN_EINVAL;
  if (!tsgnn_contract_dense_supported(N, K, F)) return TSGNN_EUNSUPPORTED;
  if (ro_arg && (F % 4 == 0) && ((ro_ldo % 4) || ((reinterpret_cast<uintptr_t>(ro_dout) | reinterpret_cast<uintptr_t>(ro_arg)) & 15)))
    return TSGNN_EUNSUPPORTED;
  if (B == 0 || (!ds && !dz && !dadj)) return TSGNN_OK;
  const size_t lds = ct_lds_floats(N, K, F, true) * sizeof(float);
  static size_t attr = 0;
  if (lds > 64 * 1024 && lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(contract_dense_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT_LDS_MAX);
    attr = CT_LDS_MAX;
  }
  contract_dense_bwd_kernel<<<(unsigned)B, CT_THREADS, lds, stream>>>(s, z, adj, t, dxo, dao, N, K, F, ds, dz, dadj, ro_dout, ro_ldo, ro_arg, softmax);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* 1 if tsgnn_contract_rows_bwd_f32 takes (K clusters, F features): K % 4 == 0, K <= 64, F % 4 == 0, F <= 256 */
int tsgnn_contract_rows_bwd_supported(int K, int F) {
  return (K > 0 && F > 0 && K % 4 == 0 && F % 4 == 0 && K <= CR_KMAX && F <= CR_FMAX) ? 1 : 0;
}

/* backward of the row-layout contraction X'[b] = S_b^T Z_b, A'[b] = S_b^T (AS)_b:  dZ = S dX', dS = Z dX'^T + AS dA'^T,
 * dAS = S dA' for the rows of every slab (slab_row_ptr[nslab + 1]: <= 32 consecutive rows of ONE graph each, slab_graph[nslab]);
 * rows [zero_from, zero_to) of the three outputs are cleared.  16-byte aligned rows. */
int tsgnn_contract_rows_bwd_f32(const float* S, int64_t ldS, const float* Z, int64_t ldZ, const float* AS, int64_t ldAS,
                                const float* dxo, const float* dao, const int* slab_row_ptr, const int* slab_graph, int nslab, int K,
                                int F, float* dZ, int64_t lddZ, float* dS, int64_t lddS, float* dAS, int64_t lddAS,
                                int64_t zero_from, int64_t zero_to, tsgnn_stream_t stream) {
  return tsgnn_contract_rows_bwd_ro_f32(S, ldS, Z, ldZ, AS, ldAS, dxo, dao, slab_row_ptr, slab_graph, nslab, K, F, dZ, lddZ, dS, lddS, dAS,
                                        lddAS, zero_from, zero_to, nullptr, 0, nullptr, stream);
}

/* the same; dZ additionally takes the gradient of the max readout of Z over each graph's rows (ro_dout [B, F] with leading dimension
 * ro_ldo, ro_arg [B, F] = winning row or -1; winners outside the slabs' rows — ghost rows — are dropped, as the caller discards them) */
int tsgnn_contract_rows_bwd_ro_f32(const float* S, int64_t ldS, const float* Z, int64_t ldZ, const float* AS, int64_t ldAS,
                                   const float* dxo, const float* dao, const int* slab_row_ptr, const int* slab_graph, int nslab, int K,
                                   int F, float* dZ, int64_t lddZ, float* dS, int64_t lddS, float* dAS, int64_t lddAS,
                                   int64_t zero_from, int64_t zero_to, const float* ro_dout, int64_t ro_ldo, const int* ro_arg,
                                   tsgnn_stream_t stream) {
  if (!S || !Z || !AS || !dxo || !dao || !slab_row_ptr || !slab_graph || !dZ || !dS || !dAS || nslab < 0 || zero_to < zero_from)
    return TSGNN_EINVAL;
  if ((ro_dout == nullptr) != (ro_arg == nullptr) || (ro_arg && ro_ldo < F)) return TSGNN_EINVAL;
  if (!tsgnn_contract_rows_bwd_supported(K, F) || (ldS % 4) || (ldZ % 4) || (ldAS % 4) || (lddZ % 4) || (lddAS % 4) || ldS < K ||
      ldZ < F || ldAS < K || lddZ < F || lddS < K || lddAS < K ||
      ((reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(AS) | reinterpret_cast<uintptr_t>(dxo) |
        reinterpret_cast<uintptr_t>(dao) | reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(dAS)) & 15))
    return TSGNN_EUNSUPPORTED;
  const int64_t nz = zero_to - zero_from;
  const unsigned zblocks = nz > 0 ? (unsigned)(nz < 64 ? nz : 64) : 0u;
  if (nslab == 0 && zblocks == 0) return TSGNN_OK;
  const size_t lds = ct_rows_lds_floats(K, F) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(contract_rows_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT_LDS_MAX);
    attr = true;
  }
  CtRows a{S, ldS, Z, ldZ, AS, ldAS, dxo, dao, slab_row_ptr, slab_graph, nslab, K, F, dZ, lddZ, dS, lddS, dAS, lddAS, zero_from, zero_to,
           ro_dout, ro_ldo, ro_arg, {}};
  ct_rows_jobs(K, F, a.job);
  contract_rows_bwd_kernel<<<(unsigned)nslab + zblocks, 256, lds, stream>>>(a);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
