// The torch_geometric-named SAGEConv layer as fused launches (BASELINE.json north_star: "SAGEConv ... drop-ins"; SURVEY §8 a15 —
// no call site in the reference, PARITY UNPINNED: the arithmetic is PyG's documented  lin_l(mean_{j in N(i)} x_j) + lin_r(x_i)).
//
//   tsgnn_sage_conv_f32                one launch per layer and direction (sageconv_body.h): gather + scale + both products + bias
//                                      [+ ReLU] [+ L2 normalise] [+ per-graph max / sum readouts of the output in the epilogue]
//   tsgnn_sage_relu_readout_bwd_f32    du = (dxs + readout gradients) * [h > 0]: the row-wise pass between two layers' backward
//   tsgnn_sage_readout_decode_f32      packed maxima / fixed-point sums of all layers -> sum_l [gmp || gap] and the arg-max rows
//   tsgnn_sage_wgrad_reduce_oi_f32     slab partials of all layers -> nn.Linear-layout gradients (lin_l.weight, lin_l.bias, lin_r.weight)
#include <algorithm>
#include "common.h"
#include "../../include/tsgnn.h"
#include "sageconv_body.h"
#include "tn_rows_body.h"

namespace {

__global__ __launch_bounds__(512) void sage_conv_kernel(SageConvArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  sageconv_body(g, smem, blockIdx.x);
}

// the weight-gradient slabs of BOTH weights of a layer in one launch: blockIdx.z = 0: (z^T du, colsum du) -> lin_l, 1: x^T du -> lin_r
// (the unchanged slab body of gemm.hip; a CU hosts one block of each role)
template <int MT, int NTt, int NY>
__global__ __launch_bounds__(256) void sage_wgrad_pair_kernel(TnArgs a, TnArgs b) {
  extern __shared__ __attribute__((aligned(16))) float tn_smem[];
  tn_rows_body<MT, NTt, NY>(blockIdx.z ? b : a, tn_smem, blockIdx.x, blockIdx.y, gridDim.x);
}

// du[r, c] = ( dxs[r, c] + [arg[b, c] == r] dread[b, c] + dread[b, F + c] / n_b ) * [h[r, c] > 0]      b = row_graph[r]
// (backward of h = relu(u) feeding the next layer, the max readout and the mean readout, Code/sag/network.py:34-46 shape)
__global__ __launch_bounds__(256) void sage_relu_readout_bwd_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ dxs,
                                                                    int64_t lddxs, const float* __restrict__ dread, int64_t lddr,
                                                                    const int* __restrict__ arg, const int* __restrict__ row_graph,
                                                                    const int* __restrict__ graph_ptr, int64_t rows, int F, int relu,
                                                                    float* __restrict__ du, int64_t lddu, const float* __restrict__ row_scale,
                                                                    float* __restrict__ dus, int64_t lddus) {
  const int F4 = F >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * F4) return;
  const int64_t r = idx / F4;
  const int c = 4 * (int)(idx % F4);
  const int b = row_graph[r];
  const float4 hv = *reinterpret_cast<const float4*>(h + r * ldh + c);
  float4 d = dxs ? *reinterpret_cast<const float4*>(dxs + r * lddxs + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (dread) {
    const float inv_n = 1.0f / (float)max(graph_ptr[b + 1] - graph_ptr[b], 1);
    const float4 dm = *reinterpret_cast<const float4*>(dread + (int64_t)b * lddr + c);
    const float4 ds = *reinterpret_cast<const float4*>(dread + (int64_t)b * lddr + F + c);
    const int4 a = *reinterpret_cast<const int4*>(arg + (int64_t)b * F + c);
    const int ri = (int)r;
    d.x += (a.x == ri ? dm.x : 0.f) + ds.x * inv_n;
    d.y += (a.y == ri ? dm.y : 0.f) + ds.y * inv_n;
    d.z += (a.z == ri ? dm.z : 0.f) + ds.z * inv_n;
    d.w += (a.w == ri ? dm.w : 0.f) + ds.w * inv_n;
  }
  if (relu) {
    d.x = hv.x > 0.f ? d.x : 0.f; d.y = hv.y > 0.f ? d.y : 0.f; d.z = hv.z > 0.f ? d.z : 0.f; d.w = hv.w > 0.f ? d.w : 0.f;
  }
  st_out(reinterpret_cast<float4*>(du + r * lddu + c), d);
  if (dus) {                                              // the same rows scaled by 1 / deg: what the next launch GATHERS
    const float sc = row_scale[r];
    st_out(reinterpret_cast<float4*>(dus + r * lddus + c), make_float4(d.x * sc, d.y * sc, d.z * sc, d.w * sc));
  }
}

// read[b, f] = sum_l max_l[b, f] ; read[b, F + f] = sum_l sum_l[b, f] / n_b ; arg[l, b, f] = the row that holds layer l's max;
// the packed maxima and the integer sums are left ZERO for the next step (this launch is their only reader)
__global__ __launch_bounds__(128) void sage_readout_decode_kernel(unsigned long long* __restrict__ packed, unsigned long long* __restrict__ sums,
                                                                  const int* __restrict__ graph_ptr, int B, int L, int F,
                                                                  float* __restrict__ read, int64_t ldr, int* __restrict__ arg) {
  const int b = blockIdx.x;
  const float inv_n = 1.0f / (float)max(graph_ptr[b + 1] - graph_ptr[b], 1);
  for (int f = threadIdx.x; f < F; f += 128) {
    float m = 0.f, s = 0.f;
    for (int l = 0; l < L; ++l) {
      const int64_t o = ((int64_t)l * B + b) * F + f;
      const unsigned long long pk = packed[o];
      const long long q = (long long)sums[o];
      packed[o] = 0ull; sums[o] = 0ull;
      m += pk ? ordered_f32((unsigned)(pk >> 32)) : 0.f;                 // (pk == 0: a graph without rows)
      s += (float)((double)q * (1.0 / SC_RO_FIX)) * inv_n;
      arg[o] = (int)(0xFFFFFFFFu - (unsigned)(pk & 0xFFFFFFFFull));
    }
    read[(int64_t)b * ldr + f] = m;
    read[(int64_t)b * ldr + F + f] = s;
  }
}

// ---- fragment-major copies of the weights (the B operand of sageconv_body.h)
struct PackSet {
  const float* w; int64_t ldw; int K, N; int kn;        // kn = 0: w[n * ldw + k] (nn.Linear's [out, in]); 1: w[k * ldw + n]
  float4* out;                                          // [4 waves][16 steps][64 lanes]
};
struct PackArgs { PackSet s[16]; int nsets; };

// out[(wv * 16 + u) * 64 + lane] = W[k = 8u + 4h + 0..3][n = 32 wv + i]  (lane = 32 h + i), zero beyond K / N
__global__ __launch_bounds__(256) void sage_conv_pack_kernel(PackArgs a) {
  const PackSet& s = a.s[blockIdx.y];
  const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;         // 0 .. 4095
  const int lane = e & 63, u = (e >> 6) & 15, wv = e >> 10;
  const int i = lane & 31, h = lane >> 5;
  const int n = 32 * wv + i, k0 = 8 * u + 4 * h;
  float v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = k0 + c;
    const bool ok = n < s.N && k < s.K;
    v[c] = ok ? (s.kn ? s.w[(int64_t)k * s.ldw + n] : s.w[(int64_t)n * s.ldw + k]) : 0.f;
  }
  s.out[e] = make_float4(v[0], v[1], v[2], v[3]);
}

struct OiSet {
  const float* ws; int nslab, K, N;     // slabs [nslab][K + 1][N] (row K = bias partial); K = 0: rows of partial column sums only
  float* dw; int64_t lddw;              // dw[n * lddw + k]  (nn.Linear's [out, in]; kn = 1: dw[k * lddw + n], GCNConv's [in, out]); unused when K = 0
  int kn;
  float* db;                            // nullable [n_db]
  int n_db;                             // columns of the bias row that go to db (N for a weight set)
  float* tail;                          // nullable: column n_db of the bias row goes to tail[0] (the SAGPool score layer's
                                        // partial rows [nb][F + 4]: dw_s in columns 0 .. F-1, db_s in column F)
  int first_block;
};
constexpr int OI_MAX_SETS = 12;
struct OiArgs { OiSet s[OI_MAX_SETS]; int nsets; float* normparts; float* step_state; };

// block -> 64 consecutive entries (k, n) of one set's [K + 1][N] slab image, n fastest: coalesced slab reads; four wave groups
// split the slabs (fixed ranges), their partial sums meet in LDS and are added in group order: the same bits every run
__global__ __launch_bounds__(256) void sage_wgrad_reduce_oi_kernel(OiArgs a) {
  __shared__ float lds[4][64];
  int si = 0;
#pragma unroll
  for (int t = 1; t < OI_MAX_SETS; ++t)
    if (t < a.nsets && (int)blockIdx.x >= a.s[t].first_block) si = t;
  const OiSet& s = a.s[si];
  const int e_l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = ((int)blockIdx.x - s.first_block) * 64 + e_l;
  const int tot = (s.K + 1) * s.N;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < tot) {
    const int64_t stride = (int64_t)tot;
    const int per = (s.nslab + 3) / 4;
    const int s0 = grp * per, s1 = min(s.nslab, s0 + per);
    int sl = s0;
    for (; sl + 4 <= s1; sl += 4) {
      a0 += s.ws[(int64_t)sl * stride + e];
      a1 += s.ws[(int64_t)(sl + 1) * stride + e];
      a2 += s.ws[(int64_t)(sl + 2) * stride + e];
      a3 += s.ws[(int64_t)(sl + 3) * stride + e];
    }
    for (; sl < s1; ++sl) a0 += s.ws[(int64_t)sl * stride + e];
  }
  lds[grp][e_l] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp != 0) return;
  float sq = 0.f;
  if (e < tot) {
    const float v = (lds[0][e_l] + lds[1][e_l]) + (lds[2][e_l] + lds[3][e_l]);
    const int k = e / s.N, n = e % s.N;
    if (k < s.K) { s.dw[s.kn ? (int64_t)k * s.lddw + n : (int64_t)n * s.lddw + k] = v; sq = v * v; }
    else if (s.db && n < s.n_db) { s.db[n] = v; sq = v * v; }
    else if (s.tail && n == s.n_db) { s.tail[0] = v; sq = v * v; }
  }
  if (a.normparts) {                                    // this block's share of |grad|^2 (summed in fixed order by the optimiser)
    sq = wave_sum(sq);
    if (e_l == 0) a.normparts[blockIdx.x] = sq;
  }
  if (a.step_state && blockIdx.x == 0 && e_l == 0) a.step_state[0] += 1.f;   // optimiser step counter, ahead of the update kernel
}

}  // namespace

extern "C" {

int tsgnn_sage_conv_supported(int K, int N) { return (K >= 1 && K <= 128 && N >= 1 && N <= 128) ? 1 : 0; }

/* desc (HOST memory): [nsets <= 16, nsets x (w, ldw, K, N, kn, out)]: fragment-major copies (16,384 floats each) of weight matrices */
int tsgnn_sage_conv_pack_f32(const int64_t* desc, tsgnn_stream_t stream) {
  if (!desc) return TSGNN_EINVAL;
  const int nsets = (int)desc[0];
  if (nsets <= 0 || nsets > 16) return TSGNN_EINVAL;
  PackArgs a{};
  a.nsets = nsets;
  const int64_t* d = desc + 1;
  for (int t = 0; t < nsets; ++t, d += 6) {
    PackSet& s = a.s[t];
    s.w = reinterpret_cast<const float*>(d[0]); s.ldw = d[1]; s.K = (int)d[2]; s.N = (int)d[3]; s.kn = (int)d[4];
    s.out = reinterpret_cast<float4*>(d[5]);
    if (!s.w || !s.out || !tsgnn_sage_conv_supported(s.K, s.N) || s.ldw < (s.kn ? s.N : s.K)) return TSGNN_EINVAL;
    if (reinterpret_cast<uintptr_t>(s.out) & 15) return TSGNN_EUNSUPPORTED;
  }
  TSGNN_KNAME("sage_conv_pack_kernel");
  sage_conv_pack_kernel<<<dim3(16, (unsigned)nsets), 256, 0, stream>>>(a);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sage_conv_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* xg, int64_t ldxg, const float* xs,
                        int64_t ldxs, const float* dst_scale, const float* wl_pk, const float* wr_pk,
                        const float* bias, float* out, int64_t ldo, float* zout, int64_t ldz, float* rinv,
                        int64_t rows, int K, int N, int relu_out, int normalize, unsigned long long* ro_packed,
                        unsigned long long* ro_sums, const int* ro_row_graph, const int* ro_graph_ptr, const float* post_h, int64_t post_ldh,
                        const float* post_dread, int64_t post_lddr, const int* post_arg, const float* post_row_scale, float* out2,
                        int64_t ldo2, tsgnn_stream_t stream) {
  if (!ell || !xg || !xs || !wl_pk || !wr_pk || !out || rows < 0 || K <= 0 || N <= 0) return TSGNN_EINVAL;
  if ((tail_ptr == nullptr) != (tail_col == nullptr)) return TSGNN_EINVAL;
  if ((ro_packed == nullptr) != (ro_sums == nullptr) || (ro_packed && (!ro_row_graph || !ro_graph_ptr))) return TSGNN_EINVAL;
  if (post_h && (!post_dread || !post_arg || !ro_row_graph || !ro_graph_ptr || post_ldh < N || post_lddr < 2 * N || ro_packed || relu_out ||
                 normalize || (out2 && (!post_row_scale || ldo2 < N))))
    return TSGNN_EINVAL;
  if (!post_h && out2) return TSGNN_EINVAL;
  if (ell_w != 4 && ell_w != 8 && ell_w != 16) return TSGNN_EUNSUPPORTED;
  if (!tsgnn_sage_conv_supported(K, N)) return TSGNN_EUNSUPPORTED;
  const int K4 = (K + 3) / 4 * 4;
  if ((ldxg % 4) || (ldxs % 4) || ldxg < K4 || ldxs < K4 || ldo < N || (zout && ((ldz % 4) || ldz < K4))) return TSGNN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(ell) | reinterpret_cast<uintptr_t>(xg) | reinterpret_cast<uintptr_t>(xs) | reinterpret_cast<uintptr_t>(zout) |
       reinterpret_cast<uintptr_t>(wl_pk) | reinterpret_cast<uintptr_t>(wr_pk)) & 15)
    return TSGNN_EUNSUPPORTED;
  if (rows == 0) return TSGNN_OK;
  if (rows >= (int64_t)1 << 31) return TSGNN_EUNSUPPORTED;
  SageConvArgs g{xg, ldxg, xs, ldxs, ell, ell_w, tail_ptr, tail_col, dst_scale, reinterpret_cast<const float4*>(wl_pk),
                 reinterpret_cast<const float4*>(wr_pk), bias, out, ldo, zout, ldz, rinv, rows, K, N, relu_out, normalize, ro_packed, ro_sums,
                 ro_row_graph, ro_graph_ptr, post_h, post_ldh, post_dread, post_lddr, post_arg, post_row_scale, out2, ldo2};
  constexpr size_t lds = sageconv_lds_bytes();
  static bool attr = false;
  if (!attr && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sage_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  TSGNN_KNAME("sage_conv_kernel");
  sage_conv_kernel<<<(unsigned)ceil_div64(rows, 32), 512, lds, stream>>>(g);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* slab partials of (z[:, :K]^T du, colsum du) into ws_l and of x[:, :K]^T du into ws_r ([nslab][K + 1][N] each, the layout of
 * tsgnn_linear_wgrad_f32's dw == NULL form; plan with tsgnn_linear_wgrad_plan) in ONE launch */
int tsgnn_sage_wgrad_pair_f32(const float* z, int64_t ldz, const float* x, int64_t ldx, const float* du, int64_t lddu, int64_t rows, int K,
                              int N, int nslab, int64_t rows_per_slab, float* ws_l, float* ws_r, tsgnn_stream_t stream) {
  if (!z || !x || !du || !ws_l || !ws_r || rows < 0 || nslab <= 0 || rows_per_slab <= 0 || K <= 0 || N <= 0) return TSGNN_EINVAL;
  if (K > 128 || N > 128 || (ldz % 4) || (ldx % 4) || (lddu % 4) || (N % 4) ||
      ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(du)) & 15))
    return TSGNN_EUNSUPPORTED;
  TnArgs a{z, ldz, du, lddu, rows, rows_per_slab, K, N, ws_l, nullptr, 0};
  TnArgs b{x, ldx, du, lddu, rows, rows_per_slab, K, N, ws_r, nullptr, 0};
  const int mt = (K + 31) / 32, nt = (N + 31) / 32;
  const unsigned ny = (mt * nt >= 8 && nslab < 512) ? 2u : 1u;
  const dim3 grid((unsigned)nslab, ny, 2);
  TSGNN_KNAME("sage_wgrad_pair_kernel<%d,%d,%u>", mt, nt, ny);
#define TSGNN_TN(M_, N_) do { \
    constexpr size_t lds_ = tn_rows_lds_bytes<M_, N_>(); \
    static bool attr_ = false; \
    if (ny == 2) { \
      if (!attr_ && lds_ > 48 * 1024) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sage_wgrad_pair_kernel<M_, N_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); attr_ = true; } \
      sage_wgrad_pair_kernel<M_, N_, 2><<<grid, 256, lds_, stream>>>(a, b); \
    } else { \
      if (!attr_ && lds_ > 48 * 1024) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sage_wgrad_pair_kernel<M_, N_, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); attr_ = true; } \
      sage_wgrad_pair_kernel<M_, N_, 1><<<grid, 256, lds_, stream>>>(a, b); \
    } } while (0)
  switch (mt * 10 + nt) {
    case 11: TSGNN_TN(1, 1); break; case 12: TSGNN_TN(1, 2); break; case 13: TSGNN_TN(1, 3); break; case 14: TSGNN_TN(1, 4); break;
    case 21: TSGNN_TN(2, 1); break; case 22: TSGNN_TN(2, 2); break; case 23: TSGNN_TN(2, 3); break; case 24: TSGNN_TN(2, 4); break;
    case 31: TSGNN_TN(3, 1); break; case 32: TSGNN_TN(3, 2); break; case 33: TSGNN_TN(3, 3); break; case 34: TSGNN_TN(3, 4); break;
    case 41: TSGNN_TN(4, 1); break; case 42: TSGNN_TN(4, 2); break; case 43: TSGNN_TN(4, 3); break; default: TSGNN_TN(4, 4); break;
  }
#undef TSGNN_TN
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sage_relu_readout_bwd_f32(const float* h, int64_t ldh, const float* dxs, int64_t lddxs, const float* dread, int64_t lddr,
                                    const int* arg, const int* row_graph, const int* graph_ptr, int64_t rows, int F, int relu, float* du,
                                    int64_t lddu, const float* row_scale, float* dus, int64_t lddus, tsgnn_stream_t stream) {
  if (!h || !du || rows < 0 || F <= 0 || (dread && (!arg || !row_graph || !graph_ptr)) || (dus && (!row_scale || (lddus % 4))))
    return TSGNN_EINVAL;
  if ((F % 4) || (ldh % 4) || (lddu % 4) || (dxs && (lddxs % 4)) || (dread && (lddr % 4)) ||
      ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(dxs) | reinterpret_cast<uintptr_t>(dread) |
        reinterpret_cast<uintptr_t>(arg)) & 15))
    return TSGNN_EUNSUPPORTED;
  if (!row_graph) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  const int64_t n = rows * (F / 4);
  TSGNN_KNAME("sage_relu_readout_bwd_kernel");
  sage_relu_readout_bwd_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(h, ldh, dxs, lddxs, dread, lddr, arg, row_graph, graph_ptr, rows, F,
                                                                              relu, du, lddu, row_scale, dus, lddus);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sage_readout_decode_f32(unsigned long long* packed, unsigned long long* sums, const int* graph_ptr, int B, int L, int F, float* read,
                                  int64_t ldr, int* arg, tsgnn_stream_t stream) {
  if (!packed || !sums || !graph_ptr || !read || !arg || B <= 0 || L <= 0 || F <= 0 || ldr < 2 * F) return TSGNN_EINVAL;
  TSGNN_KNAME("sage_readout_decode_kernel");
  sage_readout_decode_kernel<<<(unsigned)B, 128, 0, stream>>>(packed, sums, graph_ptr, B, L, F, read, ldr, arg);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* desc (HOST memory): [nsets <= 12, nsets x (ws, nslab, K, N, dw_oi, lddw, db, n_db, tail, kn)] — slab sets in the layout of
 * tsgnn_linear_wgrad_f32 (dw == NULL form), summed in slab order and written transposed: dw_oi[n * lddw + k] (kn = 1: as they lie,
 * dw[k * lddw + n]); db nullable, takes the first n_db columns of the bias row (n_db = N for a weight set).  K = 0: a set of partial ROWS [nslab][N] only (dw_oi unused) whose
 * column sums go to db[0 .. n_db) and, column n_db, to tail[0] when tail != NULL (the SAGPool score layer's per-graph partial rows
 * [nb][F + 4] left by tsgnn_sag_pool_graph_bwd_f32: the work of tsgnn_sag_du_reduce_f32 riding in this launch). */
int tsgnn_sage_wgrad_reduce_oi_blocks(const int64_t* desc) {
  if (!desc || desc[0] <= 0 || desc[0] > OI_MAX_SETS) return -1;
  int blocks = 0;
  for (int t = 0; t < (int)desc[0]; ++t) blocks += (((int)desc[1 + 10 * t + 2] + 1) * (int)desc[1 + 10 * t + 3] + 63) / 64;
  return blocks;
}

int tsgnn_sage_wgrad_reduce_oi_f32(const int64_t* desc, float* normparts, float* step_state, tsgnn_stream_t stream) {
  if (!desc) return TSGNN_EINVAL;
  const int nsets = (int)desc[0];
  if (nsets <= 0 || nsets > OI_MAX_SETS) return TSGNN_EINVAL;
  OiArgs a{};
  a.nsets = nsets; a.normparts = normparts; a.step_state = step_state;
  int blocks = 0;
  const int64_t* d = desc + 1;
  for (int t = 0; t < nsets; ++t, d += 10) {
    OiSet& s = a.s[t];
    s.ws = reinterpret_cast<const float*>(d[0]); s.nslab = (int)d[1]; s.K = (int)d[2]; s.N = (int)d[3];
    s.dw = reinterpret_cast<float*>(d[4]); s.lddw = d[5]; s.db = reinterpret_cast<float*>(d[6]); s.n_db = (int)d[7];
    s.tail = reinterpret_cast<float*>(d[8]); s.kn = d[9] ? 1 : 0;
    if (!s.ws || s.nslab <= 0 || s.K < 0 || s.N <= 0 || s.n_db < 0 || s.n_db > s.N) return TSGNN_EINVAL;
    if (s.K > 0 && (!s.dw || s.lddw < (s.kn ? s.N : s.K))) return TSGNN_EINVAL;
    if (s.tail && s.n_db >= s.N) return TSGNN_EINVAL;
    s.first_block = blocks;
    blocks += ((s.K + 1) * s.N + 63) / 64;
  }
  TSGNN_KNAME("sage_wgrad_reduce_oi_kernel");
  sage_wgrad_reduce_oi_kernel<<<(unsigned)blocks, 256, 0, stream>>>(a);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

namespace {
struct CopyJob { const float* src; int64_t lds; int rows, cols; float* dst; int64_t ldd; int dst_cols; };
struct CopyArgs { CopyJob j[16]; int njobs; };
__global__ __launch_bounds__(256) void copy2d_multi_kernel(CopyArgs a) {
  const CopyJob& j = a.j[blockIdx.y];
  const int64_t tot = (int64_t)j.rows * j.dst_cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / j.dst_cols), c = (int)(e % j.dst_cols);
    j.dst[(int64_t)r * j.ldd + c] = c < j.cols ? j.src[(int64_t)r * j.lds + c] : 0.f;
  }
}
}  // namespace

/* desc (HOST memory): [njobs <= 16, njobs x (src, lds, rows, cols, dst, ldd, dst_cols)]: dst[r, 0 .. dst_cols) = src[r, 0 .. cols) then
 * zeros — several small matrices placed (and zero-padded) in ONE launch: the [W_l | W_r] images of every level of a SAGEConv stack
 * (sag_stack_sage.py), which torch builds with two pads and a concatenation per level. */
int tsgnn_copy2d_multi_f32(const int64_t* desc, tsgnn_stream_t stream) {
  if (!desc) return TSGNN_EINVAL;
  const int njobs = (int)desc[0];
  if (njobs <= 0 || njobs > 16) return TSGNN_EINVAL;
  CopyArgs a{};
  a.njobs = njobs;
  int64_t most = 0;
  const int64_t* d = desc + 1;
  for (int t = 0; t < njobs; ++t, d += 7) {
    CopyJob& j = a.j[t];
    j.src = reinterpret_cast<const float*>(d[0]); j.lds = d[1]; j.rows = (int)d[2]; j.cols = (int)d[3];
    j.dst = reinterpret_cast<float*>(d[4]); j.ldd = d[5]; j.dst_cols = (int)d[6];
    if (!j.src || !j.dst || j.rows <= 0 || j.cols <= 0 || j.dst_cols < j.cols || j.lds < j.cols || j.ldd < j.dst_cols) return TSGNN_EINVAL;
    most = std::max<int64_t>(most, (int64_t)j.rows * j.dst_cols);
  }
  const unsigned gx = (unsigned)std::min<int64_t>((most + 255) / 256, 64);
  TSGNN_KNAME("copy2d_multi_kernel");
  copy2d_multi_kernel<<<dim3(gx, (unsigned)njobs), 256, 0, stream>>>(a);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
