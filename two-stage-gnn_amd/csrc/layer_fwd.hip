// One launch for a hidden GraphConv layer's forward product and the max-readout partial of its INPUT (both only read x,
// the previous layer's output):
//
//   v = normalise((A x) W + b), z = A x            (rowgemm_body GATHER, rowgemm.hip; ghost rows by the filler block)
//   packed[b, f] = max over the slots of graph b   (readout_partial_body, sage_fused.hip)
//
// The readout blocks are short and ride along on the CUs that each host one latency-bound row-panel block.
#include "common.h"
#include "../../include/tsgnn.h"
#include "rowgemm_body.h"
#include "readout_body.h"
#include "ingest_rider.h"

namespace {

// RO: the product's epilogue also folds the max readout of its OWN output into ga.ro_packed (last layer of the stack: no
// slot batch-norm follows, so the readout is taken on v itself)
template <bool RO>
__global__ __launch_bounds__(256) void sage_layer_fwd_kernel(RowGemmArgs ga, SlotArgs sa, unsigned n_gemm, unsigned ro_gx, int F4,
                                                             unsigned long long* __restrict__ packed, unsigned n_main, PullRider pr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.x < n_gemm) {
    rowgemm_body<4, false, true, 1, RO>(ga, smem, blockIdx.x);
  } else if (blockIdx.x < n_main) {
    const unsigned r = blockIdx.x - n_gemm;
    readout_partial_body<32>(sa, ga.a, ga.lda, F4, packed, r % ro_gx, r / ro_gx, reinterpret_cast<unsigned long long*>(smem));
  } else {
    // passengers: a share of the next mini-batch's staging buffer -> its mirror (ingest_rider.h).  (As the FIRST workgroups of the launch
    // they changed nothing for the ingest step and cost the resident step 0.4 us per launch — measured both ways.)
    pull_rider_body(pr, blockIdx.x - n_main);
  }
}

// The same launch for a layer whose INPUT's slot batch-norm was not materialised (rowgemm_body.h BNIN / STATS, readout_body.h):
// x = the previous layer's v; the gather and the readout partial form y = BN(relu(v)) on the fly from the previous layer's integer
// sums.  ST: this layer is followed by a batch-norm too (statistics epilogue); RO: it is the last one (readout epilogue).
template <bool RO, bool ST, bool UNITS = false>
__global__ __launch_bounds__(256, 2) void sage_layer_fwd_bn_kernel(RowGemmArgs ga, SlotArgs sa, BnReadArgs bn, unsigned n_gemm, unsigned ro_gx,
                                                                int ro_ch, int F4, unsigned long long* __restrict__ packed, unsigned n_main,
                                                                PullRider pr, const int* __restrict__ ro_map) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.x < n_gemm) {
    rowgemm_body<4, false, true, 1, RO, true, ST, UNITS>(ga, smem, blockIdx.x);
  } else if (blockIdx.x < n_main) {
    // ro_map (nullable): which (graph, chunk) this block scans — chosen on the host so that the block sits on the XCD whose row
    // panels gather that graph's rows (blocks b, b + 8, ... share an XCD): the rows are in that L2 already instead of being
    // fetched into another one a second time
    unsigned r = blockIdx.x - n_gemm;
    if (ro_map) r = (unsigned)ro_map[r];
    readout_partial_bn_body<32>(sa, bn, ga.a, ga.lda, F4, packed, r % ro_gx, r / ro_gx, ro_ch, reinterpret_cast<unsigned long long*>(smem));
  } else {
    pull_rider_body(pr, blockIdx.x - n_main);
  }
}

}  // namespace

extern "C" {

/* slots per readout block and row-panel blocks (filler included) of a tsgnn_sage_layer_fwd_bn_f32 launch: what a caller needs to build ro_map */
int tsgnn_sage_layer_fwd_bn_plan(int64_t rows, int64_t fill_rows, int B, int nslots, int* ro_ch, int* n_gemm) {
  if (!ro_ch || !n_gemm || rows <= 0 || B <= 0 || nslots <= 0) return TSGNN_EINVAL;
  int dev = 0, v = 0;
  const int ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  int n_full_ = 0, unit_ = 16;
  const unsigned ng = panel_split(rows, ncu, &n_full_, &unit_) + (fill_rows > 0 ? 1u : 0u);
  int ch = 64;
  while (ch < 256 && ng + (unsigned)((nslots + ch - 1) / ch) * (unsigned)B > 2u * (unsigned)ncu) ch *= 2;
  *ro_ch = ch; *n_gemm = (int)ng;
  return TSGNN_OK;
}

/* tsgnn_sage_layer_fwd[_ro]_f32 for a layer whose input's slot batch-norm (apply_bn, encoders.py:134-138) has NO launch of its own:
 * x = the previous layer's normalised pre-activations v, sums_in / ghost_in = what its statistics epilogue left
 * (tsgnn_gather_rowgemm_st_f32 or this entry point with row_slot != NULL), slot_count[n] = graphs with more than n nodes.  Every
 * row-panel block turns the sums into (mean, rstd) per slot and gathers y_j = (relu(v_j) - mean[slot_j]) * rstd[slot_j]; the
 * readout partial does the same for the rows it scans, and the blocks of graph 0 write mean_out / rstd_out [nslots] for the
 * backward.  ell: entry = slot << 20 | row (GraphBatch.ell_slots(); no CSR tail), nslots <= 1024, rows < 2^20.
 * row_slot != NULL: this layer is followed by a batch-norm as well: its statistics go to sums_out / ghost_out (zero before).
 * packed_out != NULL: last layer, readout epilogue (as tsgnn_sage_layer_fwd_ro_f32).
 * ro_map (nullable, B * ceil(nslots / ro_map_ch) ints, a permutation): readout block r scans work item ro_map[r] = graph * chunks + chunk;
 * tsgnn_sage_layer_fwd_bn_plan tells the chunk size the launch will use. */
int tsgnn_sage_layer_fwd_bn_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                                float* v, int64_t ldv, float* rinv, float* zout, int64_t ldz, int64_t rows, int K, int64_t fill_rows,
                                const int* graph_ptr, const int* slot_count, int B, int nslots, int n_ghost, unsigned long long* packed,
                                unsigned long long* packed_out, const int* row_graph, const unsigned long long* sums_in,
                                const float* ghost_in, float* mean_out, float* rstd_out, const int* row_slot,
                                unsigned long long* sums_out, float* ghost_out, const int* ro_map, int ro_map_ch, tsgnn_stream_t stream) {
  if (!ell || !x || !w || !v || !rinv || !graph_ptr || !slot_count || !packed || !sums_in || !ghost_in || !mean_out || !rstd_out ||
      rows <= 0 || fill_rows < 0 || K <= 0 || B <= 0 || nslots <= 0 || (n_ghost != 0 && n_ghost != nslots) || (packed_out && !row_graph) ||
      (row_slot && (!sums_out || !ghost_out)))
    return TSGNN_EINVAL;
  if (ell_w != 4 && ell_w != 8 && ell_w != 16) return TSGNN_EUNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(ell) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) |
                       reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(zout) | reinterpret_cast<uintptr_t>(bias) |
                       reinterpret_cast<uintptr_t>(sums_in) | reinterpret_cast<uintptr_t>(sums_out);
  if ((al & 15) || K != 128 || (ldx % 4) || (ldw % 4) || (ldv % 4) || (zout && (ldz % 4 || ldz < K)) || ldx < K || ldw < 128 || ldv < 128 ||
      nslots > BN_TAB || rows >= (1 << 20) || n_ghost == 0 || (packed_out && row_slot))
    return TSGNN_EUNSUPPORTED;
  if (packed_out && fill_rows <= 0) return TSGNN_EUNSUPPORTED;
  if ((tail_ptr == nullptr) != (tail_col == nullptr)) return TSGNN_EINVAL;
  RowGemmArgs ga{x, ldx, w, ldw, bias, v, ldv, rinv, rows, K, 128, 1, fill_rows, ell, ell_w, zout, ldz, tail_ptr, tail_col,
                 packed_out, graph_ptr, row_graph, B, nslots, n_ghost};
  ga.st_row_slot = row_slot; ga.st_sums = sums_out; ga.st_ghost = ghost_out;
  ga.bn_sums = sums_in; ga.bn_ghost = ghost_in; ga.bn_slot_count = slot_count; ga.bn_B = B; ga.bn_nslots = nslots; ga.bn_F = K;
  SlotArgs sa{graph_ptr, slot_count, B, nslots, rows, n_ghost};
  BnReadArgs bn{sums_in, ghost_in, K, mean_out, rstd_out};
  // slots per readout block: 64 while [panels + readout blocks] fit two blocks per compute unit (what the registers allow), else 128
  // or 256 — DD seed 2: 266 panels + 256 readout blocks = 523 > 512 ran a third round for eleven blocks (13.9 -> 17.8 us)
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const unsigned n_gemm = panel_split(rows, ncu, &ga.n_full, &ga.unit) + (fill_rows > 0 ? 1u : 0u);   // (a few more panels than CUs: 16-row units)
  int ro_ch = 64;
  while (ro_ch < 256 && n_gemm + (unsigned)((nslots + ro_ch - 1) / ro_ch) * (unsigned)B > 2u * (unsigned)ncu) ro_ch *= 2;
  const unsigned ro_gx = (unsigned)((nslots + ro_ch - 1) / ro_ch);
  if (ro_map && ro_map_ch != ro_ch) ro_map = nullptr;      // (the map was built for another chunk size: plain order)
  size_t lds = rowgemm_lds_bytes<4, false, true, 1, true>();
  const size_t lro = 8 * 128 * sizeof(unsigned long long) + 256 * sizeof(float2);
  if (lds < lro) lds = lro;
  const unsigned n_main = n_gemm + ro_gx * (unsigned)B;
  const PullRider pr = take_pull_rider();
  const bool units = ga.unit == 8 || ga.unit == 16;     // (rows beyond one panel per CU as 16- / 8-row units: the UNITS build of the kernels)
#define TSGNN_FWD_BN(RO_, ST_)                                                                                                              \
  do {                                                                                                                                      \
    if (units) sage_layer_fwd_bn_kernel<RO_, ST_, true><<<n_main + pr.blocks, 256, lds, stream>>>(ga, sa, bn, n_gemm, ro_gx, ro_ch, K / 4, packed, n_main, pr, ro_map); \
    else sage_layer_fwd_bn_kernel<RO_, ST_><<<n_main + pr.blocks, 256, lds, stream>>>(ga, sa, bn, n_gemm, ro_gx, ro_ch, K / 4, packed, n_main, pr, ro_map);             \
  } while (0)
  if (packed_out) {
    TSGNN_KNAME("sage_layer_fwd_bn_kernel<true,false,%s>", units ? "true" : "false");
    TSGNN_FWD_BN(true, false);
  } else if (row_slot) {
    TSGNN_KNAME("sage_layer_fwd_bn_kernel<false,true,%s>", units ? "true" : "false");
    TSGNN_FWD_BN(false, true);
  } else {
    TSGNN_KNAME("sage_layer_fwd_bn_kernel<false,false,%s>", units ? "true" : "false");
    TSGNN_FWD_BN(false, false);
  }
#undef TSGNN_FWD_BN
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* packed_out (nullable) [B*128], zeroed by the caller: receives the packed max readout of the layer's own OUTPUT v (real rows
 * from the product's epilogue, each graph's first ghost row from the filler block); row_graph[rows] = graph of each row. */
int tsgnn_sage_layer_fwd_ro_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                                float* v, int64_t ldv, float* rinv, float* zout, int64_t ldz, int64_t rows, int K, int64_t fill_rows,
                                const int* graph_ptr, int B, int nslots, int n_ghost, unsigned long long* packed,
                                unsigned long long* packed_out, const int* row_graph, tsgnn_stream_t stream) {
  if (!ell || !x || !w || !v || !rinv || !graph_ptr || !packed || rows <= 0 || fill_rows < 0 || K <= 0 || B <= 0 || nslots <= 0 ||
      (n_ghost != 0 && n_ghost != nslots) || (packed_out && !row_graph))
    return TSGNN_EINVAL;
  if (ell_w != 4 && ell_w != 8 && ell_w != 16) return TSGNN_EUNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(ell) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) |
                       reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(zout) | reinterpret_cast<uintptr_t>(bias);
  if ((al & 15) || K != 128 || (ldx % 4) || (ldw % 4) || (ldv % 4) || (zout && (ldz % 4 || ldz < K)) || ldx < K || ldw < 128 || ldv < 128)
    return TSGNN_EUNSUPPORTED;            /* x is also read as the [rows, 128] readout operand: hidden layers only */
  if ((tail_ptr == nullptr) != (tail_col == nullptr)) return TSGNN_EINVAL;
  if (packed_out && n_ghost && fill_rows <= 0) return TSGNN_EUNSUPPORTED;   /* the ghost rows' share comes from the filler block */
  RowGemmArgs ga{x, ldx, w, ldw, bias, v, ldv, rinv, rows, K, 128, 1, fill_rows, ell, ell_w, zout, ldz, tail_ptr, tail_col,
                 packed_out, graph_ptr, row_graph, B, nslots, n_ghost};
  SlotArgs sa{graph_ptr, nullptr, B, nslots, rows, n_ghost};
  const unsigned n_gemm = (unsigned)ceil_div64(rows, 32) + (fill_rows > 0 ? 1u : 0u);
  const unsigned ro_gx = (unsigned)((nslots + 63) / 64);
  size_t lds = rowgemm_lds_bytes<4, false, true>();
  if (lds < 8 * 128 * sizeof(unsigned long long)) lds = 8 * 128 * sizeof(unsigned long long);
  const unsigned n_main = n_gemm + ro_gx * (unsigned)B;
  const PullRider pr = take_pull_rider();                  // (blocks = 0 unless tsgnn_ingest_arm_pull_rider armed one on this thread)
  if (packed_out) {
    TSGNN_KNAME("sage_layer_fwd_kernel<true>");
    sage_layer_fwd_kernel<true><<<n_main + pr.blocks, 256, lds, stream>>>(ga, sa, n_gemm, ro_gx, K / 4, packed, n_main, pr);
  } else {
    TSGNN_KNAME("sage_layer_fwd_kernel<false>");
    sage_layer_fwd_kernel<false><<<n_main + pr.blocks, 256, lds, stream>>>(ga, sa, n_gemm, ro_gx, K / 4, packed, n_main, pr);
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_sage_layer_fwd_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                             float* v, int64_t ldv, float* rinv, float* zout, int64_t ldz, int64_t rows, int K, int64_t fill_rows,
                             const int* graph_ptr, int B, int nslots, int n_ghost, unsigned long long* packed, tsgnn_stream_t stream) {
  return tsgnn_sage_layer_fwd_ro_f32(ell, ell_w, tail_ptr, tail_col, x, ldx, w, ldw, bias, v, ldv, rinv, zout, ldz, rows, K, fill_rows,
                                     graph_ptr, B, nslots, n_ghost, packed, nullptr, nullptr, stream);
}

}  // extern "C"
