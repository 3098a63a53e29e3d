// Row-panel fp32 MFMA GEMM for the GraphConv transform and its input gradient:
//
//   C[R, N] = A[R, K] . B[K, N]            B = W            (forward,  encoders.py:36)
//   C[R, N] = A[R, K] . B^T (B is [N, K])  B = W            (dZ = dU . W^T)
//   optional epilogue: + bias, row L2 normalise (F.normalize eps 1e-12, encoders.py:38-40), rinv out
//
// R is the number of graph rows (thousands to millions), K and N are feature widths (<= 256).
// One 256-thread block owns 32 rows x all N columns.  K is consumed in 32-wide chunks through a software
// pipeline (16-byte global loads two chunks ahead -> registers -> two LDS stages -> operand registers one chunk
// ahead -> MFMA chain; one barrier per chunk).
// A fragments are read as ds_read_b128 from a [32][K+4] image: the K order inside an MFMA group is
// permuted (half h of the wave takes k = 8u+4h+c) so one 16-byte read feeds four v_mfma_f32_32x32x2_f32;
// with the +4 padding every 16-lane read group touches 64 distinct banks.  B fragments are ds_read_b32
// of 32 consecutive floats (conflict-free); for B = W^T the W chunk keeps its [n][k] layout in LDS and is read
// with the same 16-byte fragment pattern as A.  The bias / L2-normalise epilogue works on the accumulators in
// registers (row sums of squares: DPP over 32 lanes, then 4 waves through LDS).  fp32 MFMA throughout.
#include "common.h"
#include "../../include/tsgnn.h"

#include <cstdlib>
#include "rowgemm_body.h"
#include "ingest_rider.h"

int tsgnn_panel_split_on_ = 1;
#include "rowgemm_big_body.h"

namespace {

inline int device_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t p;
      if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
    }
    return cus;
  }();
  return n;
}

inline unsigned ks2_max_blocks() {                       // TSGNN_KS2_MAX_BLOCKS: up to how many row panels the two-group kernel is used
  static const unsigned n = [] { const char* e = getenv("TSGNN_KS2_MAX_BLOCKS"); return e ? (unsigned)atoi(e) : (unsigned)device_cu_count(); }();
  return n;
}
inline bool rowgemm_ks2_enabled() {                      // TSGNN_ROWGEMM_KS2=0 selects the one-group kernel (A/B measurements)
  static const bool on = [] { const char* e = getenv("TSGNN_ROWGEMM_KS2"); return !(e && e[0] == '0'); }();
  return on;
}

template <int NT, bool TRANS_B, bool GATHER>
__global__ __launch_bounds__(256) void rowgemm_kernel(RowGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  rowgemm_body<NT, TRANS_B, GATHER>(g, smem, blockIdx.x);
}

// split-K variant: two groups of four waves per row panel (rowgemm_body.h, KS = 2)
template <int NT, bool TRANS_B>
__global__ __launch_bounds__(512) void rowgemm_gather_ks2_kernel(RowGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  rowgemm_body<NT, TRANS_B, true, 2>(g, smem, blockIdx.x);
}

// the same two kernels with the statistics epilogue (rowgemm_body.h, STATS): the layer in front of a slot batch-norm that has no
// launch of its own (tsgnn_gather_rowgemm_st_f32)
template <bool UNITS>
__global__ __launch_bounds__(256) void rowgemm_gather_st_kernel(RowGemmArgs g, PullRider pr, unsigned nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.x >= nblk) { pull_rider_body(pr, blockIdx.x - nblk); return; }    // passengers, as below
  rowgemm_body<4, false, true, 1, false, false, true, UNITS>(g, smem, blockIdx.x);
}
// pr: passengers (csrc/ingest_rider.h) — a share of the NEXT mini-batch's staging buffer -> its mirror as the launch's last workgroups
__global__ __launch_bounds__(512) void rowgemm_gather_ks2_st_kernel(RowGemmArgs g, PullRider pr, unsigned nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.x >= nblk) { pull_rider_body(pr, blockIdx.x - nblk); return; }    // (the launch's LAST workgroups)
  rowgemm_body<4, false, true, 2, false, false, true>(g, smem, blockIdx.x);
}

// column-split variant for products WITHOUT the row epilogue (no normalise: rows need not be whole): grid.y column blocks of
// 32 * NT.  A 1,000-row x 256-column product (GAT projection, DiffPool's dagg = du W^T) is 32 panels on a 256-CU chip whose
// waves each run two 32 x 32 tiles over the whole K: split in two, twice the CUs work and each wave's MFMA chain is half as long
// (16.0 -> 9.5 us).  The one-tile-per-wave body is also the faster one at scale (K = N = 256: 89 vs 76 TF at 131 k rows, 79 vs 72 TF
// at 555 k; scripts/colsplit_sweep.py), so every such product takes this path.
template <int NT, bool TRANS_B>
__global__ __launch_bounds__(256) void rowgemm_colsplit_kernel(RowGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n0 = (int)blockIdx.y * 32 * NT;
  g.N = min(32 * NT, g.N - n0);
  g.c += n0;
  if (g.bias) g.bias += n0;
  g.b += TRANS_B ? (int64_t)n0 * g.ldb : (int64_t)n0;
  g.rinv = nullptr;
  // a narrow last block (the 2H score columns of the packed GAT projection, N = H * Fh + 2H) runs the one-tile body
  if (!TRANS_B && g.N <= 32) rowgemm_body<1, TRANS_B, false>(g, smem, blockIdx.x);
  else rowgemm_body<NT, TRANS_B, false>(g, smem, blockIdx.x);
}

inline bool rowgemm_colsplit_enabled() {
  static const bool on = [] { const char* e = getenv("TSGNN_ROWGEMM_COLSPLIT"); return !(e && e[0] == '0'); }();
  return on;
}

inline unsigned colsplit_max_panels() {                   // TSGNN_ROWGEMM_COLSPLIT_PANELS=0 disables the column blocks (A/B measurements)
  static const unsigned n = [] {
    const char* e = getenv("TSGNN_ROWGEMM_COLSPLIT_PANELS");
    return e ? (unsigned)atoi(e) : 0xFFFFFFFFu;
  }();
  return n;
}

// Large batches: the B-stationary persistent kernel (rowgemm_big_body.h).  TSGNN_ROWGEMM_BIG_ROWS sets the row count from which it
// is used (0 disables it; A/B measurements).
inline int64_t rowgemm_big_min_rows() {
  static const int64_t n = [] {
    const char* e = getenv("TSGNN_ROWGEMM_BIG_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)49152;
  }();
  return n;
}

// rows from which products WIDER than one 128-column block (the first GAT projection: 92 -> 264) take the B-stationary kernel on
// column blocks instead of the column-split row-panel kernel (0 disables)
inline int64_t rowgemm_big_wide_min_rows() {
  static const int64_t n = [] {
    const char* e = getenv("TSGNN_ROWGEMM_BIG_WIDE_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)2048;
  }();
  return n;
}

template <int U, bool TRANS_B>
void launch_big(const RowGemmArgs& g, hipStream_t s) {
  constexpr size_t lds = rowgemm_big_lds_bytes<U, TRANS_B>();
  static const int bpc = [] {                           // resident blocks per CU (registers decide; LDS allows 4)
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_big_kernel<U, TRANS_B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(rowgemm_big_kernel<U, TRANS_B>), 256, lds) != hipSuccess || n < 1) n = 1;
    return n;
  }();
  const unsigned npanels = (unsigned)ceil_div64(g.rows, 32);
  const unsigned ny = (unsigned)((g.N + 127) / 128);
  // a last column block of <= 32 columns (the 2H score columns of a packed GAT projection) is light: only one of its waves
  // has MFMA work.  The heavy blocks share the CUs; the light ones fill in behind them.
  const unsigned heavy = (unsigned)(g.N / 128 + ((g.N % 128) > 32 ? 1 : 0));
  unsigned grid = (unsigned)device_cu_count() * (unsigned)bpc / (heavy ? heavy : 1u);
  if (grid > npanels) grid = npanels;
  if (grid == 0) grid = 1;
  if (ny > 1) {                                         // equal shares: every block the same number of panels (+- 1)
    const unsigned per = (npanels + grid - 1) / grid;
    grid = (npanels + per - 1) / per;
  }
  TSGNN_KNAME("rowgemm_big_kernel<%d,%s>", U, TRANS_B ? "true" : "false");
  rowgemm_big_kernel<U, TRANS_B><<<dim3(grid, ny), 256, lds, s>>>(g, npanels);
}

template <bool TRANS_B>
bool try_big(const RowGemmArgs& g, hipStream_t s) {
  const bool wide = g.N > 128;
  const int64_t mn = wide ? rowgemm_big_wide_min_rows() : rowgemm_big_min_rows();
  // K <= 128: the wave's slice of B fits 64 registers.  (Holding K = 256 costs 128 + 38 accumulation registers, one wave per
  // SIMD: measured 45 us against 28 us of the column-split row-panel kernel on the 8,518 x 256 x 264 GAT projection.)
  if (mn <= 0 || g.rows < mn || g.N > 384 || g.K > 128 || (TRANS_B && (g.K % 4)) || (g.N % 4)) return false;
  if (g.N > 128 && (g.normalize || g.fill_rows > 0 || g.rinv)) return false;     // column blocks: no row epilogue
  if (g.K <= 96) launch_big<12, TRANS_B>(g, s);
  else launch_big<16, TRANS_B>(g, s);
  return true;
}

template <bool TRANS_B>
bool try_colsplit(const RowGemmArgs& g, hipStream_t s) {
  const unsigned nblk = (unsigned)ceil_div64(g.rows, 32);
  if (g.normalize || g.fill_rows > 0 || g.N <= 128 || (g.N % 4) || nblk == 0 || nblk > colsplit_max_panels() ||
      !rowgemm_colsplit_enabled())
    return false;
  const size_t lds = rowgemm_lds_bytes<4, TRANS_B, false>();
  TSGNN_KNAME("rowgemm_colsplit_kernel<4,%s>", TRANS_B ? "true" : "false");
  rowgemm_colsplit_kernel<4, TRANS_B><<<dim3(nblk, (unsigned)((g.N + 127) / 128)), 256, lds, s>>>(g);
  return true;
}

template <int NT, bool TRANS_B, bool GATHER>
void launch_rowgemm(const RowGemmArgs& g, hipStream_t s) {
  const unsigned nblk = (unsigned)(ceil_div64(g.rows, 32) + (g.fill_rows > 0 ? 1 : 0));
  if constexpr (GATHER && NT <= 4) {
    // two wave groups per panel only while every panel has a CU to itself: with more panels than CUs the one-group kernel's
    // second co-resident block hides the same waits and keeps the MFMA pipe busier (measured: 308 panels 15.3 vs 13.6 us,
    // 17,324 panels 528 vs 474 us; <= 256 panels 9.5 vs 11.1 us)
    if (g.K > KC && nblk <= ks2_max_blocks() && rowgemm_ks2_enabled()) {
      constexpr size_t lds2 = rowgemm_lds_bytes<NT, TRANS_B, true, 2>();
      static bool attr = false;
      if (!attr && lds2 > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_gather_ks2_kernel<NT, TRANS_B>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        attr = true;
      }
      TSGNN_KNAME("rowgemm_gather_ks2_kernel<%d,%s>", NT, TRANS_B ? "true" : "false");
      rowgemm_gather_ks2_kernel<NT, TRANS_B><<<nblk, 512, lds2, s>>>(g);
      return;
    }
  }
  const size_t lds = rowgemm_lds_bytes<NT, TRANS_B, GATHER>();
  TSGNN_KNAME("rowgemm_kernel<%d,%s,%s>", NT, TRANS_B ? "true" : "false", GATHER ? "true" : "false");
  rowgemm_kernel<NT, TRANS_B, GATHER><<<nblk, 256, lds, s>>>(g);
}

template <bool TRANS_B, bool GATHER>
void dispatch_rowgemm(const RowGemmArgs& g, hipStream_t s) {
  switch ((g.N + 31) / 32) {
    case 1: launch_rowgemm<1, TRANS_B, GATHER>(g, s); break;
    case 2: launch_rowgemm<2, TRANS_B, GATHER>(g, s); break;
    case 3: launch_rowgemm<3, TRANS_B, GATHER>(g, s); break;
    case 4: launch_rowgemm<4, TRANS_B, GATHER>(g, s); break;
    default:
      if constexpr (!GATHER) {                           // the gather variant is built for widths <= 128
        switch ((g.N + 31) / 32) {
          case 5: launch_rowgemm<5, TRANS_B, false>(g, s); break;
          case 6: launch_rowgemm<6, TRANS_B, false>(g, s); break;
          case 7: launch_rowgemm<7, TRANS_B, false>(g, s); break;
          default: launch_rowgemm<8, TRANS_B, false>(g, s); break;
        }
      }
      break;
  }
}

}  // namespace

extern "C" {

/* 1 if tsgnn_rowgemm_f32 accepts these operands (16-byte rows / pointers, widths <= 256) */
int tsgnn_rowgemm_supported(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c, int64_t ldc, int K, int N,
                            int trans_b) {
  const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  if (!al || (lda % 4) || (ldb % 4) || N > 384 || N <= 0 || K <= 0) return 0;   // 256 < N <= 384: column-split kernel only (no row epilogue)
  if (lda < ((K + 3) / 4) * 4) return 0;
  if (!trans_b && (N % 4)) return 0;
  if (trans_b && (K % 4)) return 0;
  (void)c; (void)ldc;
  return 1;
}

int tsgnn_rowgemm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int trans_b, const float* bias, float* c,
                      int64_t ldc, float* rinv, int64_t rows, int K, int N, int normalize, int64_t fill_rows,
                      tsgnn_stream_t stream) {
  if (!a || !b || !c || rows < 0 || fill_rows < 0 || K <= 0 || N <= 0 || lda < K || ldc < N) return TSGNN_EINVAL;
  if (!tsgnn_rowgemm_supported(a, lda, b, ldb, c, ldc, K, N, trans_b)) return TSGNN_EUNSUPPORTED;
  if (fill_rows > 0 && ((N % 4) || (ldc % 4) || (reinterpret_cast<uintptr_t>(c) & 15) ||
                        (bias && (reinterpret_cast<uintptr_t>(bias) & 15))))
    return TSGNN_EUNSUPPORTED;
  if (rows == 0 && fill_rows == 0) return TSGNN_OK;
  RowGemmArgs g{a, lda, b, ldb, bias, c, ldc, rinv, rows, K, N, normalize, fill_rows, nullptr, 0, nullptr, 0};
  if (trans_b) {
    if (!try_big<true>(g, stream) && !try_colsplit<true>(g, stream)) {
      if (N > 256) return TSGNN_EUNSUPPORTED;
      dispatch_rowgemm<true, false>(g, stream);
    }
  } else {
    if (!try_big<false>(g, stream) && !try_colsplit<false>(g, stream)) {
      if (N > 256) return TSGNN_EUNSUPPORTED;
      dispatch_rowgemm<false, false>(g, stream);
    }
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* tsgnn_gather_rowgemm_f32 (trans_b = 0, normalize = 1, 96 < N <= 128) for a layer that is followed by the slot batch-norm
 * (apply_bn, encoders.py:134-138) WITHOUT a launch for it: the epilogue adds every real row's (sum_f relu(v), sum_f relu(v)^2) to
 * sums[2 * row_slot[r]] as 64-bit fixed-point integers (2^-40 units; order-independent, so bitwise reproducible) and the filler
 * block leaves the ghost row's two numbers in ghost[0..1].  sums: zero before the launch.  row_slot[r] < 0: row r belongs to no
 * graph (padding of a capacity-padded batch).  The consumer is tsgnn_sage_layer_fwd_bn_f32. */
int tsgnn_gather_rowgemm_st_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* b, int64_t ldb, const float* bias,
                                float* c, int64_t ldc, float* rinv, float* zout, int64_t ldz, int64_t rows, int K, int N,
                                int64_t fill_rows, const int* row_slot, unsigned long long* sums, float* ghost, tsgnn_stream_t stream) {
  if (!ell || !x || !b || !c || !row_slot || !sums || !ghost || rows <= 0 || fill_rows < 0 || K <= 0 || N <= 0 || ldx < K || ldc < N)
    return TSGNN_EINVAL;
  if (ell_w != 4 && ell_w != 8 && ell_w != 16) return TSGNN_EUNSUPPORTED;
  if (!tsgnn_rowgemm_supported(x, ldx, b, ldb, c, ldc, K, N, 0) || N > 128 || N <= 96 || K > 128 || (reinterpret_cast<uintptr_t>(ell) & 15) ||
      (reinterpret_cast<uintptr_t>(sums) & 15))
    return TSGNN_EUNSUPPORTED;
  if (zout && ((ldz % 4) || ldz < K || (reinterpret_cast<uintptr_t>(zout) & 15))) return TSGNN_EUNSUPPORTED;
  if ((N % 4) || (ldc % 4) || (reinterpret_cast<uintptr_t>(c) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15))) return TSGNN_EUNSUPPORTED;
  if ((tail_ptr == nullptr) != (tail_col == nullptr)) return TSGNN_EINVAL;
  RowGemmArgs g{x, ldx, b, ldb, bias, c, ldc, rinv, rows, K, N, 1, fill_rows, ell, ell_w, zout, ldz, tail_ptr, tail_col};
  g.st_row_slot = row_slot; g.st_sums = sums; g.st_ghost = ghost;
  unsigned nblk = (unsigned)(ceil_div64(rows, 32) + (fill_rows > 0 ? 1 : 0));
  const bool ks2 = K > KC && nblk <= ks2_max_blocks() && rowgemm_ks2_enabled();
  if (!ks2) {                                            // (the one-group kernel: a few more panels than CUs go as 16-row units)
    static int ncu = 0;
    if (ncu == 0) {
      int dev = 0, v = 0;
      ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    nblk = panel_split(rows, ncu, &g.n_full, &g.unit) + (fill_rows > 0 ? 1u : 0u);
  }
  if (ks2) {
    constexpr size_t lds2 = rowgemm_lds_bytes<4, false, true, 2>();
    static bool attr = false;
    if (!attr && lds2 > 64 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_gather_ks2_st_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      attr = true;
    }
    const PullRider pr = take_pull_rider(512);             // (blocks = 0 unless tsgnn_ingest_arm_pull_rider[_parts] armed one on this thread)
    TSGNN_KNAME("rowgemm_gather_ks2_st_kernel");
    rowgemm_gather_ks2_st_kernel<<<nblk + pr.blocks, 512, lds2, stream>>>(g, pr, nblk);
  } else {
    const PullRider pr = take_pull_rider(256);
    TSGNN_KNAME("rowgemm_gather_st_kernel<%s>", (g.unit == 8 || g.unit == 16) ? "true" : "false");
    if (g.unit == 8 || g.unit == 16) rowgemm_gather_st_kernel<true><<<nblk + pr.blocks, 256, rowgemm_lds_bytes<4, false, true>(), stream>>>(g, pr, nblk);
    else rowgemm_gather_st_kernel<false><<<nblk + pr.blocks, 256, rowgemm_lds_bytes<4, false, true>(), stream>>>(g, pr, nblk);
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* on = 0: the fused layer launches of this PROCESS keep plain 32-row panels until it is switched back on (default on) */
int tsgnn_panel_split_hint(int on) {
  tsgnn_panel_split_on_ = on ? 1 : 0;
  return TSGNN_OK;
}

/* workgroups the row panels of a `rows`-row launch take on the current device (rowgemm_body.h panel_split: 32-row panels, or — a few more
 * panels than compute units — one full panel per unit and the rest of the rows in 16-row units); callers that size a co-resident role
 * (the merged backward launch's slab blocks) plan with this */
int tsgnn_panel_blocks(int64_t rows) {
  if (rows <= 0) return 0;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  int nf = 0, un = 16;
  return (int)panel_split(rows, ncu, &nf, &un);
}

int tsgnn_gather_rowgemm_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* b, int64_t ldb, int trans_b,
                             const float* bias, float* c, int64_t ldc, float* rinv, float* zout, int64_t ldz, int64_t rows, int K,
                             int N, int normalize, int64_t fill_rows, tsgnn_stream_t stream) {
  if (!ell || !x || !b || !c || rows < 0 || fill_rows < 0 || K <= 0 || N <= 0 || ldx < K || ldc < N) return TSGNN_EINVAL;
  if (ell_w != 4 && ell_w != 8 && ell_w != 16) return TSGNN_EUNSUPPORTED;
  if (!tsgnn_rowgemm_supported(x, ldx, b, ldb, c, ldc, K, N, trans_b) || N > 128 || K > 128 || (reinterpret_cast<uintptr_t>(ell) & 15))
    return TSGNN_EUNSUPPORTED;
  if (zout && ((ldz % 4) || ldz < K || (reinterpret_cast<uintptr_t>(zout) & 15))) return TSGNN_EUNSUPPORTED;
  if (fill_rows > 0 && ((N % 4) || (ldc % 4) || (reinterpret_cast<uintptr_t>(c) & 15) ||
                        (bias && (reinterpret_cast<uintptr_t>(bias) & 15))))
    return TSGNN_EUNSUPPORTED;
  if (rows == 0 && fill_rows == 0) return TSGNN_OK;
  if ((tail_ptr == nullptr) != (tail_col == nullptr)) return TSGNN_EINVAL;
  RowGemmArgs g{x, ldx, b, ldb, bias, c, ldc, rinv, rows, K, N, normalize, fill_rows, ell, ell_w, zout, ldz, tail_ptr, tail_col};
  if (trans_b) dispatch_rowgemm<true, true>(g, stream);
  else dispatch_rowgemm<false, true>(g, stream);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
