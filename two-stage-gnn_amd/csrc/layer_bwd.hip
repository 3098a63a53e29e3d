// One launch for the two independent GEMM-shaped halves of a hidden GraphConv layer's backward (both consume dU):
//
//   weight / bias gradient slabs   dW_s = Z_s^T dU_s , db_s = colsum(dU_s)              (tn_rows_body, gemm.hip)
//   input gradient                 dX   = (A dU) W^T   for a symmetric A               (rowgemm_body GATHER, rowgemm.hip)
//
// Each is a grid of <= #CU latency-bound blocks with one wave per SIMD; launched together, a CU hosts one block of each
// (registers 214+32 and 220+16 per lane, LDS 64 + 67 KB) and the two MFMA streams interleave, instead of running back to
// back with a kernel boundary between them.  Blocks [0, n_tn) are the slab blocks, the rest the row panels.
#include "common.h"
#include "../../include/tsgnn.h"
#include "rowgemm_body.h"
#include "tn_rows_body.h"
#include <cstdlib>

namespace {

// PANELS_FIRST: the row panels take the low block indices (dispatched first), the slab blocks follow
template <bool PANELS_FIRST, bool UNITS = false>
__global__ __launch_bounds__(256) void sage_layer_bwd_kernel(RowGemmArgs ga, TnArgs gt, unsigned n_tn, unsigned nslab, unsigned n_pan, int slab_delay) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (PANELS_FIRST) {
    if (blockIdx.x < n_pan) rowgemm_body<4, true, true, 1, false, false, false, UNITS>(ga, smem, blockIdx.x);
    else {
      // the slab blocks finish well before the row panels (whose gather prologue is two dependent round trips): let the panels'
      // requests go first instead of competing with the slabs' 64 KB per block for the same first microseconds
      for (int i = 0; i < slab_delay; ++i) __builtin_amdgcn_s_sleep(16);
      // XCD-aware slab order: workgroups b, b + 8, ... share an XCD, and the row panels of XCD x gather a contiguous eighth of the
      // rows (xcd_remap) — a slab block is given the slab whose rows (dU, read by both roles) that XCD's L2 holds already
      const unsigned b = blockIdx.x - n_pan, by = b / nslab, sp = b % nslab;
      unsigned sl = sp;
      if ((nslab & 7u) == 0) sl = ((blockIdx.x & 7u) * (nslab >> 3)) + (sp >> 3);
      tn_rows_body<4, 4, 2>(gt, smem, sl, by, nslab);
    }
  } else {
    if (blockIdx.x < n_tn) tn_rows_body<4, 4, 2>(gt, smem, blockIdx.x % nslab, blockIdx.x / nslab, nslab);
    else rowgemm_body<4, true, true, 1, false, false, false, UNITS>(ga, smem, blockIdx.x - n_tn);
  }
}

}  // namespace

extern "C" {

int tsgnn_sage_layer_bwd_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* du, int64_t lddu, const float* w, int64_t ldw, float* dxs,
                             int64_t lddxs, const float* z, int64_t ldz, int64_t rows, int nslab, int64_t rows_per_slab,
                             int64_t bias_only_rows, float* ws, tsgnn_stream_t stream) {
  if (!ell || !du || !w || !dxs || !z || !ws || rows <= 0 || nslab <= 0 || rows_per_slab <= 0 || bias_only_rows < 0) return TSGNN_EINVAL;
  if (ell_w != 4 && ell_w != 8 && ell_w != 16) return TSGNN_EUNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(ell) | reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(w) |
                       reinterpret_cast<uintptr_t>(dxs) | reinterpret_cast<uintptr_t>(z);
  if ((al & 15) || (lddu % 4) || (ldw % 4) || (lddxs % 4) || (ldz % 4) || lddu < 128 || ldw < 128 || lddxs < 128 || ldz < 128)
    return TSGNN_EUNSUPPORTED;
  // dX = (A dU) W^T : a = dU (gathered), b = W [K_in = 128, N_out = 128] used transposed, reduction over N_out
  if ((tail_ptr == nullptr) != (tail_col == nullptr)) return TSGNN_EINVAL;
  RowGemmArgs ga{du, lddu, w, ldw, nullptr, dxs, lddxs, nullptr, rows, 128, 128, 0, 0, ell, ell_w, nullptr, 0, tail_ptr, tail_col};
  TnArgs gt{z, ldz, du, lddu, rows, rows_per_slab, 128, 128, ws, nullptr, bias_only_rows};
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const unsigned n_tn = 2u * (unsigned)nslab, n_pan = panel_split(rows, ncu, &ga.n_full, &ga.unit);   // (a few more panels than CUs: 16-row units)
  constexpr size_t la = rowgemm_lds_bytes<4, true, true>(), lt = tn_rows_lds_bytes<4, 4>();
  static const int panels_first = [] { const char* e = getenv("TSGNN_BWD_PANELS_FIRST"); return e ? atoi(e) : 1; }();
  // the slab blocks wait `delay` x ~0.43 us before their first request: with at most one row panel per CU the panels' two dependent gather
  // trips then start ahead of the slabs' 64 KB per block (headline batch, 255 panels: 0.1274 -> 0.1261 ms with 2 or 3, alternating runs on
  // one box; 1 and 4+: nothing); batches of more panels than CUs: no difference either way (seeds 1-7 within 0.1 us), so only the
  // one-panel-per-CU launches delay.  TSGNN_SLAB_DELAY overrides.
  static const int slab_delay_env = [] { const char* e = getenv("TSGNN_SLAB_DELAY"); return e ? atoi(e) : -1; }();
  const int slab_delay = slab_delay_env >= 0 ? slab_delay_env : ((int64_t)n_pan <= (int64_t)ncu ? 2 : 0);
  const bool units = ga.unit == 8 || ga.unit == 16;
  TSGNN_KNAME("sage_layer_bwd_kernel<%s,%s>", panels_first ? "true" : "false", units ? "true" : "false");   // (the demangled name, without blanks)
  if (panels_first && units) sage_layer_bwd_kernel<true, true><<<n_tn + n_pan, 256, la > lt ? la : lt, stream>>>(ga, gt, n_tn, (unsigned)nslab, n_pan, slab_delay);
  else if (panels_first) sage_layer_bwd_kernel<true><<<n_tn + n_pan, 256, la > lt ? la : lt, stream>>>(ga, gt, n_tn, (unsigned)nslab, n_pan, slab_delay);
  else if (units) sage_layer_bwd_kernel<false, true><<<n_tn + n_pan, 256, la > lt ? la : lt, stream>>>(ga, gt, n_tn, (unsigned)nslab, n_pan, slab_delay);
  else sage_layer_bwd_kernel<false><<<n_tn + n_pan, 256, la > lt ? la : lt, stream>>>(ga, gt, n_tn, (unsigned)nslab, n_pan, slab_delay);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
