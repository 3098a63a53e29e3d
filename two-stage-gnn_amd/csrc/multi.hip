// Several INDEPENDENT problems of the GraphConv layer kernels in ONE launch (blocks of different roles side by side, as
// layer_bwd.hip does for the two halves of one layer's backward).  DiffPool's first level runs two 64-wide GCN stacks on the same
// graph (embedding and assignment, encoders.py:352-363): each of their kernels fills about half of the CUs (DD b16: 139 row
// panels, 185 slabs), so the same step of both stacks shares a launch:
//     forward   [ gather . W + bias + normalise (stack e) | (stack a) ]
//     backward  [ dW slabs (e) | dW slabs (a) | dX = (A dU) W^T (e) | (a) ]
// The bodies are the unchanged device functions (tn_rows_body<MT, 2, 1>, rowgemm_body<2, TRANS, GATHER = true>, one wave group):
// every problem gets bit for bit the result of its own launch.  All problems of one kind must share their shapes (one template).
#include "common.h"
#include "../../include/tsgnn.h"
#include "rowgemm_body.h"
#include "tn_rows_body.h"

namespace {

struct MultiArgs {
  RowGemmArgs g0, g1;
  TnArgs t0, t1;
  unsigned ntn, ng;           // problems of each kind (0..2)
  unsigned tn_blocks, nslab;  // blocks per weight-gradient problem (= slabs, NY = 1)
  unsigned g_blocks;          // blocks per product problem (row panels + filler)
  float4* zero[2];            // nullable: cleared by the filler block of product 0 / 1 AFTER its own rows (the embedding mask of
  int64_t zero_n4[2];         // a stack that returns node features: the ghost rows of the whole concatenation, the fill rows included)
};

template <int MT, int GMODE>   // MT: 32-row tiles of K_in of the weight-gradient problems (0: none); GMODE 0: no product, 1: W, 2: W^T
__global__ __launch_bounds__(256) void sage_multi_kernel(MultiArgs m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned b = blockIdx.x;
  if constexpr (MT > 0) {
    const unsigned tn_total = m.ntn * m.tn_blocks;
    if (b < tn_total) {
      const bool second = b >= m.tn_blocks;
      tn_rows_body<MT, 2, 1>(second ? m.t1 : m.t0, smem, second ? b - m.tn_blocks : b, 0, m.nslab);
      return;
    }
    b -= tn_total;
  }
  if constexpr (GMODE > 0) {
    const bool second = b >= m.g_blocks;
    const unsigned bl = second ? b - m.g_blocks : b;
    rowgemm_body<2, GMODE == 2, true>(second ? m.g1 : m.g0, smem, bl);
    float4* z = m.zero[second];
    if (z && bl == m.g_blocks - 1) {                     // the filler block: all of its threads are back here
      __syncthreads();                                   // its fill rows are stored before they are cleared
      const int64_t n4 = m.zero_n4[second];
      for (int64_t i = threadIdx.x; i < n4; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

inline bool tn_same(const TnArgs& a, const TnArgs& b) {
  return a.rows == b.rows && a.rows_per_slab == b.rows_per_slab && a.K_in == b.K_in && a.N == b.N && a.bias_only_rows == b.bias_only_rows;
}
inline bool g_same(const RowGemmArgs& a, const RowGemmArgs& b) {
  return a.rows == b.rows && a.K == b.K && a.N == b.N && a.normalize == b.normalize && a.fill_rows == b.fill_rows && a.ell_w == b.ell_w &&
         (a.tail_ptr == nullptr) == (b.tail_ptr == nullptr);
}

template <int MT, int GMODE>
void launch_multi(const MultiArgs& m, hipStream_t s) {
  constexpr size_t lt = MT > 0 ? tn_rows_lds_bytes<MT, 2>() : 0;
  constexpr size_t lg = GMODE > 0 ? rowgemm_lds_bytes<2, GMODE == 2, true>() : 0;
  constexpr size_t lds = lt > lg ? lt : lg;
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sage_multi_kernel<MT, GMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const unsigned grid = m.ntn * m.tn_blocks + m.ng * m.g_blocks;
  TSGNN_KNAME("sage_multi_kernel<%d,%d>", MT, GMODE);
  sage_multi_kernel<MT, GMODE><<<grid, 256, lds, s>>>(m);
}

}  // namespace

extern "C" {

/* words per problem in the description of tsgnn_sage_multi_f32 */
int tsgnn_sage_multi_tn_words(void) { return 11; }
int tsgnn_sage_multi_g_words(void) { return 20; }

/* Up to two weight-gradient slab problems (the arguments of tsgnn_linear_wgrad_f32 with dw = db = NULL) and up to two gather
 * products (the arguments of tsgnn_gather_rowgemm_f32) in one launch.  desc (HOST memory):
 *   [ntn, ng,
 *    ntn x (z, ldz, du, lddu, rows, K_in, N, nslab, rows_per_slab, bias_only_rows, ws),
 *    ng  x (ell, ell_w, tail_ptr, tail_col, x, ldx, b, ldb, trans_b, bias, c, ldc, rinv, zout, ldz, rows, K, N, normalize, fill_rows)]
 * N = 64 wide (33..64), K, K_in <= 128, the problems of one kind share every shape; TSGNN_EUNSUPPORTED otherwise (launch them
 * one by one then). */
int tsgnn_sage_multi_zero_f32(const int64_t* desc, float* zero0, int64_t n0, float* zero1, int64_t n1, tsgnn_stream_t stream);
int tsgnn_sage_multi_f32(const int64_t* desc, tsgnn_stream_t stream) { return tsgnn_sage_multi_zero_f32(desc, nullptr, 0, nullptr, 0, stream); }

/* The same launch; additionally zero0[0..n0) / zero1[0..n1) are cleared by the filler block of product 0 / 1 after it has written
 * its fill rows (the regions may contain them).  Needs fill_rows > 0, n % 4 == 0 and 16-byte aligned regions. */
int tsgnn_sage_multi_zero_f32(const int64_t* desc, float* zero0, int64_t n0, float* zero1, int64_t n1, tsgnn_stream_t stream) {
  if (!desc || n0 < 0 || n1 < 0 || (!zero0 && n0) || (!zero1 && n1)) return TSGNN_EINVAL;
  const int ntn = (int)desc[0], ng = (int)desc[1];
  if (ntn < 0 || ntn > 2 || ng < 0 || ng > 2 || ntn + ng == 0) return TSGNN_EINVAL;
  MultiArgs m{};
  m.ntn = (unsigned)ntn; m.ng = (unsigned)ng;
  const int64_t* d = desc + 2;
  TnArgs* ts[2] = {&m.t0, &m.t1};
  RowGemmArgs* gs[2] = {&m.g0, &m.g1};
  int nslab = 0;
  for (int i = 0; i < ntn; ++i, d += 11) {
    TnArgs& t = *ts[i];
    t.z = reinterpret_cast<const float*>(d[0]); t.ldz = d[1]; t.du = reinterpret_cast<const float*>(d[2]); t.lddu = d[3];
    t.rows = d[4]; t.K_in = (int)d[5]; t.N = (int)d[6]; t.rows_per_slab = d[8]; t.bias_only_rows = d[9];
    t.slabs = reinterpret_cast<float*>(d[10]); t.slab_row_ptr = nullptr;
    if (i == 0) nslab = (int)d[7]; else if (nslab != (int)d[7]) return TSGNN_EUNSUPPORTED;
    if (!t.z || !t.du || !t.slabs || t.rows < 0 || nslab <= 0 || t.rows_per_slab <= 0 || t.K_in <= 0 || t.N <= 0 || t.bias_only_rows < 0) return TSGNN_EINVAL;
    if (t.K_in > 128 || t.N <= 32 || t.N > 64 || (t.ldz % 4) || (t.lddu % 4) || (t.N % 4) ||
        ((reinterpret_cast<uintptr_t>(t.z) | reinterpret_cast<uintptr_t>(t.du)) & 15))
      return TSGNN_EUNSUPPORTED;
  }
  if (ntn == 2 && !tn_same(m.t0, m.t1)) return TSGNN_EUNSUPPORTED;
  if (ntn == 1) m.t1 = m.t0;
  int trans = 0;
  for (int i = 0; i < ng; ++i, d += 20) {
    RowGemmArgs& g = *gs[i];
    g.ell = reinterpret_cast<const int*>(d[0]); g.ell_w = (int)d[1];
    g.tail_ptr = reinterpret_cast<const int*>(d[2]); g.tail_col = reinterpret_cast<const int*>(d[3]);
    g.a = reinterpret_cast<const float*>(d[4]); g.lda = d[5]; g.b = reinterpret_cast<const float*>(d[6]); g.ldb = d[7];
    if (i == 0) trans = (int)d[8]; else if (trans != (int)d[8]) return TSGNN_EUNSUPPORTED;
    g.bias = reinterpret_cast<const float*>(d[9]); g.c = reinterpret_cast<float*>(d[10]); g.ldc = d[11];
    g.rinv = reinterpret_cast<float*>(d[12]); g.zout = reinterpret_cast<float*>(d[13]); g.ldz = d[14];
    g.rows = d[15]; g.K = (int)d[16]; g.N = (int)d[17]; g.normalize = (int)d[18]; g.fill_rows = d[19];
    if (!g.ell || !g.a || !g.b || !g.c || g.rows <= 0 || g.fill_rows < 0 || g.K <= 0 || g.N <= 0 || g.lda < g.K || g.ldc < g.N) return TSGNN_EINVAL;
    if ((g.tail_ptr == nullptr) != (g.tail_col == nullptr)) return TSGNN_EINVAL;
    if ((g.ell_w != 4 && g.ell_w != 8 && g.ell_w != 16) || g.N <= 32 || g.N > 64 || g.K > 128 || (g.N % 4) || (g.lda % 4) || (g.ldb % 4) ||
        (g.ldc % 4) || (trans && (g.K % 4)) || (g.zout && ((g.ldz % 4) || g.ldz < g.K)) ||
        ((reinterpret_cast<uintptr_t>(g.a) | reinterpret_cast<uintptr_t>(g.b) | reinterpret_cast<uintptr_t>(g.c) | reinterpret_cast<uintptr_t>(g.ell) |
          reinterpret_cast<uintptr_t>(g.zout) | reinterpret_cast<uintptr_t>(g.bias)) & 15))
      return TSGNN_EUNSUPPORTED;
  }
  if (ng == 2 && !g_same(m.g0, m.g1)) return TSGNN_EUNSUPPORTED;
  if (ng == 1) m.g1 = m.g0;
  m.nslab = (unsigned)nslab;
  m.tn_blocks = (unsigned)nslab;
  m.g_blocks = ng ? (unsigned)(ceil_div64(m.g0.rows, 32) + (m.g0.fill_rows > 0 ? 1 : 0)) : 0u;
  if (zero0 || zero1) {
    if (!ng || m.g0.fill_rows <= 0 || (zero1 && ng < 2)) return TSGNN_EUNSUPPORTED;
    if ((n0 % 4) || (n1 % 4) || ((reinterpret_cast<uintptr_t>(zero0) | reinterpret_cast<uintptr_t>(zero1)) & 15)) return TSGNN_EUNSUPPORTED;
    m.zero[0] = reinterpret_cast<float4*>(zero0); m.zero_n4[0] = n0 / 4;
    m.zero[1] = reinterpret_cast<float4*>(zero1); m.zero_n4[1] = n1 / 4;
  }
  const int mt = ntn ? (m.t0.K_in + 31) / 32 : 0;
  const int gm = ng ? (trans ? 2 : 1) : 0;
  switch (mt * 10 + gm) {
    case 1: launch_multi<0, 1>(m, stream); break;
    case 2: launch_multi<0, 2>(m, stream); break;
    case 10: launch_multi<1, 0>(m, stream); break;
    case 20: launch_multi<2, 0>(m, stream); break;
    case 30: launch_multi<3, 0>(m, stream); break;
    case 40: launch_multi<4, 0>(m, stream); break;
    case 22: launch_multi<2, 2>(m, stream); break;
    case 42: launch_multi<4, 2>(m, stream); break;
    default: return TSGNN_EUNSUPPORTED;
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
