// Backward of a GAT layer's packed projection hp = x W' (gat_fused.hip; encoders_GAT.py:29-36): both dense products that consume
// dhp in ONE launch —
//
//   weight gradient slabs   dW' = x^T dhp        (blocked: 128 x 128 output sets x row slabs; wgrad_blocks_body.h / tn_rows_body.h)
//   input gradient          dx  = dhp W'^T       (row panels x column blocks of 128; rowgemm_body.h, B given transposed)
//
// Alone, each is a few hundred two-per-CU blocks with a tail: the 8,518-row DD batch has 267 row panels = 534 column blocks on the
// 512 slots of the chip, and the 22 that wait run on a nearly empty machine (18.1 us at 256 panels, 23.7 at 257); the
// slab launch ends the same way.  Launched together — the input-gradient blocks first, the slab blocks behind them — the slab blocks
// fill the slots as the panels drain, one kernel boundary is gone, and the layer below gets dx no later than before.  The slabs are
// reduced by tsgnn_wgrad_blocks_reduce_f32 as after a launch of their own (same bits as the separate launches: same blocks, same order).
#include "common.h"
#include "../../include/tsgnn.h"
#include "rowgemm_body.h"
#include "wgrad_blocks_body.h"

namespace {

// TRANS_B: the input-gradient product's B operand is given as W[N_out][K] (the GAT layer: dx = dhp W'^T with W' [K_in, N]) or as
// B[K][N_out] (a torch.nn.Linear: dx = dy W with W [out, in])
template <bool TRANS_B>
__global__ __launch_bounds__(256) void gat_bwd_products_kernel(RowGemmArgs g, WgradBlocks w, unsigned npan, unsigned n_dx, unsigned nslab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.x < n_dx) {
    // column block y of row panel x, in the dispatch order of the (npan, ny) grid of rowgemm_colsplit_kernel (rowgemm.hip)
    const unsigned y = blockIdx.x / npan, x = blockIdx.x % npan;
    const int n0 = (int)y * 128;
    g.N = min(128, g.N - n0);
    g.c += n0;
    g.b += TRANS_B ? (int64_t)n0 * g.ldb : (int64_t)n0;   // (transposed: rows of W'[N][K] = output columns)
    rowgemm_body<4, TRANS_B, false>(g, smem, x);
  } else {
    const unsigned b = blockIdx.x - n_dx;
    wgrad_blocks_role<2>(w, smem, b % nslab, b / nslab, nslab);
  }
}

}  // namespace

extern "C" {

/* dw_slabs (ws) <- the slab partials of dW'[K_in, N] = x[:, :K_in]^T du  (reduce with tsgnn_wgrad_blocks_reduce_f32; plan with
 * tsgnn_wgrad_blocks_plan(rows, K_in, N, ldx, lddu)) and dx[rows, K_in] = du[rows, N] . wp[K_in, N]^T, one launch.
 * K_in, N <= 512, both multiples of 4, 16-byte aligned rows everywhere.  (K_in = N = 128: the slabs have tsgnn_linear_wgrad_f32's
 * layout [nslab][K_in + 1][N] — the SAGPool conv layers reduce them with tsgnn_linear_wgrad_du_reduce_f32.) */
int tsgnn_gat_bwd_products_f32(const float* x, int64_t ldx, const float* du, int64_t lddu, int64_t rows, int K_in, int N, const float* wp,
                               int64_t ldwp, float* dx, int64_t lddx, int nslab, int64_t rows_per_slab, float* ws, tsgnn_stream_t stream) {
  if (!x || !du || !wp || !dx || !ws || rows <= 0 || K_in <= 0 || N <= 0 || nslab <= 0 || rows_per_slab <= 0 || ldx < K_in || lddu < N ||
      ldwp < N || lddx < K_in || (int64_t)nslab * rows_per_slab < rows)
    return TSGNN_EINVAL;
  const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(wp) |
                       reinterpret_cast<uintptr_t>(dx);
  if ((al & 15) || (ldx % 4) || (lddu % 4) || (ldwp % 4) || (lddx % 4) || (N % 4) || (K_in % 4) || K_in > 512 || N > 512)
    return TSGNN_EUNSUPPORTED;
  const int KB = (K_in + 127) / 128, NB = (N + 127) / 128, nsets = KB * NB;
  if (nsets > WB_MAXSETS) return TSGNN_EUNSUPPORTED;
  // dx = du . wp^T : A = du [rows, N], B = wp as W[N_out = K_in][K = N] row-major (trans_b form), no epilogue
  RowGemmArgs g{du, lddu, wp, ldwp, nullptr, dx, lddx, nullptr, rows, N, K_in, 0, 0, nullptr, 0, nullptr, 0};
  WgradBlocks w{TnArgs{x, ldx, du, lddu, rows, rows_per_slab, K_in, N, ws, nullptr, 0}, NB, nsets, (int64_t)nslab * WB_SET_FLOATS};
  const unsigned npan = (unsigned)ceil_div64(rows, 32), ny = (unsigned)KB;
  const unsigned n_dx = npan * ny, n_w = (unsigned)nslab * 2u * (unsigned)nsets;
  constexpr size_t la = rowgemm_lds_bytes<4, true, false>(), lt = 2 * TN_CH * 32 * (4 + 4) * sizeof(float);
  constexpr size_t lds = la > lt ? la : lt;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gat_bwd_products_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  TSGNN_KNAME("gat_bwd_products_kernel<true>");
  gat_bwd_products_kernel<true><<<n_dx + n_w, 256, lds, stream>>>(g, w, npan, n_dx, (unsigned)nslab);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* The same pairing for a torch.nn.Linear y = x W^T + b with W [N = out, K_in = in] (message_passing._LinearOI: DiffPool's assignment
 * predictor, encoders.py:362-372): the slab partials of (dW^T, db) from x and dy into ws (reduce with tsgnn_wgrad_blocks_reduce_oi_f32)
 * and dx[rows, K_in] = dy[rows, N] . W in one launch.  K_in, N <= 512, K_in % 4 == 0, N % 4 == 0. */
int tsgnn_linear_bwd_products_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int K_in, int N, const float* w,
                                  int64_t ldw, float* dx, int64_t lddx, int nslab, int64_t rows_per_slab, float* ws, tsgnn_stream_t stream) {
  if (!x || !dy || !w || !dx || !ws || rows <= 0 || K_in <= 0 || N <= 0 || nslab <= 0 || rows_per_slab <= 0 || ldx < K_in || lddy < N ||
      ldw < K_in || lddx < K_in || (int64_t)nslab * rows_per_slab < rows)
    return TSGNN_EINVAL;
  const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(w) |
                       reinterpret_cast<uintptr_t>(dx);
  if ((al & 15) || (ldx % 4) || (lddy % 4) || (ldw % 4) || (lddx % 4) || (N % 4) || (K_in % 4) || K_in > 512 || N > 512)
    return TSGNN_EUNSUPPORTED;
  const int KB = (K_in + 127) / 128, NB = (N + 127) / 128, nsets = KB * NB;
  if (nsets > WB_MAXSETS) return TSGNN_EUNSUPPORTED;
  // dx = dy . W : A = dy [rows, N], B = W[K = N][N_out = K_in] row-major, no epilogue
  RowGemmArgs g{dy, lddy, w, ldw, nullptr, dx, lddx, nullptr, rows, N, K_in, 0, 0, nullptr, 0, nullptr, 0};
  WgradBlocks wb{TnArgs{x, ldx, dy, lddy, rows, rows_per_slab, K_in, N, ws, nullptr, 0}, NB, nsets, (int64_t)nslab * WB_SET_FLOATS};
  const unsigned npan = (unsigned)ceil_div64(rows, 32), ny = (unsigned)KB;
  const unsigned n_dx = npan * ny, n_w = (unsigned)nslab * 2u * (unsigned)nsets;
  constexpr size_t la = rowgemm_lds_bytes<4, false, false>(), lt = 2 * TN_CH * 32 * (4 + 4) * sizeof(float);
  constexpr size_t lds = la > lt ? la : lt;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gat_bwd_products_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  TSGNN_KNAME("gat_bwd_products_kernel<false>");
  gat_bwd_products_kernel<false><<<n_dx + n_w, 256, lds, stream>>>(g, wb, npan, n_dx, (unsigned)nslab);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
