// Body of the row-panel fp32 MFMA product (see rowgemm.hip for the description), as a device function so that other
// kernels can run it next to independent work in one launch (layer_bwd.hip).  `bid` = linear panel/block index,
// `smem` = the block's dynamic LDS (rowgemm_lds_bytes<...>()).
#pragma once
#include "common.h"
#include <type_traits>

extern int tsgnn_panel_split_on_;                       // rowgemm.hip; see panel_split() below
#include <cstdlib>

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
  int64_t fill_rows;                  // rows after `rows` that get the epilogue of a zero input row
  // GATHER variant: the A operand is the unit-weight aggregation  A[r,:] = sum_k a[ell[r*ell_w + k], :]  (entries < 0
  // skipped), built chunk by chunk while it is staged; zout (nullable) receives it (the weight gradient needs it).
  const int* ell; int ell_w;
  float* zout; int64_t ldz;
  const int* tail_ptr; const int* tail_col;   // nullable: CSR of the neighbours beyond ell_w (rows with longer lists)
  // READOUT variant (epilogue): the per-graph column maxima of the OUTPUT rows are folded into ro_packed[B, N] as packed
  // (ordered value, ~row) 64-bit atomicMax, the arg-max readout of encoders.py:183 without a pass of its own; the filler
  // block adds every graph's first ghost row (row rows + size_b; all ghost rows of this layer carry the same value).
  unsigned long long* ro_packed;              // nullable
  const int* ro_graph_ptr; const int* ro_row_graph;
  int ro_B, ro_nslots, ro_nghost;
  // Slot batch-norm WITHOUT a launch of its own (apply_bn, encoders.py:134-138, between two GraphConv layers):
  // STATS (epilogue): every output row adds  sum_f relu(v), sum_f relu(v)^2  to st_sums[2 * slot] as 64-bit FIXED-POINT integers
  //   (2^-40 units: integer addition is associative, so the totals do not depend on the order the panels finish in — bitwise
  //   reproducible, unlike float atomics); the filler block leaves the same two numbers of the ghost row in st_ghost[0..1].
  // BNIN (gather prologue): the gathered operand rows are the PREVIOUS layer's normalised pre-activations v; every block turns
  //   the previous layer's sums into (mean, rstd) per slot (LDS table, exact from the integer sums, ghost copies by multiplicity
  //   B - slot_count[n]) and gathers  y_j = (relu(v_j) - mean[slot_j]) * rstd[slot_j]  on the fly.  The neighbour table then
  //   carries the slot beside the row:  entry = slot << 20 | row  (GraphBatch.ell_slots()).
  const int* st_row_slot; unsigned long long* st_sums; float* st_ghost;
  const unsigned long long* bn_sums; const float* bn_ghost; const int* bn_slot_count; int bn_B, bn_nslots, bn_F;
  // HALF PANELS (round 4; 0: every block is a 32-row panel).  n_full > 0: blocks [0, n_full) are 32-row panels of rows [0, 32 n_full), the
  // blocks behind them 16-ROW panels of the remaining rows (then the filler block).  A batch of a few more panels than CUs puts a second
  // panel on some CUs, and a CU with two panels decides the launch (1.3-1.4 x): with one full panel per CU and the overflow in 16-row
  // (or 8-row) units the busiest CU gathers 40-48 rows instead of 64 (panel_split() below).
  int n_full;
  int unit;                           // rows per unit of the blocks behind the full panels (16 or 8: panel_unit_rows())
};

// blocks of a panel launch of `rows` rows on `ncu` compute units: n_full (0: plain 32-row panels) and the total number of panel blocks
// rows per unit behind the full panels: 8 while the overflow is small (up to ncu / 16 panels: the extra blocks — each reads all of W and
// runs a whole MFMA chain — stay few), 16 above.  Measured on the DD batches of seeds 0-7 (ms per step, units of 32 = plain / 16 / 8):
// 266-271 panels 0.1395-0.1437 / 0.1381-0.1420 / 0.1371-0.1414; 288 panels 0.1406 / 0.1378 / 0.1410.  TSGNN_PANEL_UNIT forces 8 or 16.
inline int panel_unit_rows(int64_t rows, int ncu) {
  static const int forced = [] { const char* e = getenv("TSGNN_PANEL_UNIT"); const int v = e ? atoi(e) : 0; return (v == 8 || v == 16) ? v : 0; }();
  if (forced) return forced;
  return ((rows + 31) / 32 - ncu) <= ncu / 16 ? 8 : 16;
}
// 0: launches keep plain 32-row panels (tsgnn_panel_split_hint; process-wide — a backward's launches are issued by autograd's own
// thread, so a thread-local switch would miss them: capacity-padded batches, whose rows beyond one panel
// per CU are mostly PADDING — cut into units they cost a block's fixed work each and gather nothing: the ingest step 0.1494 -> 0.1541 ms)
inline unsigned panel_split(int64_t rows, int ncu, int* n_full, int* unit) {
  static const bool on = [] { const char* e = getenv("TSGNN_HALF_PANELS"); return e ? atoi(e) != 0 : true; }();
  const int64_t P = (rows + 31) / 32;
  *n_full = 0; *unit = 32;
  if (!on || !tsgnn_panel_split_on_ || ncu < 8) return (unsigned)P;
  // (small batches — PROTEINS b64: 77 panels — cut entirely into 16- or 8-row units to occupy more CUs: 116.9 -> 117.2 / 125.4 us per
  // step, measured and left out: a block's fixed work — all of W, a whole MFMA chain — does not shrink with its rows)
  if (P <= ncu || P - ncu > ncu / 2) return (unsigned)P;
  const int nf = ncu & ~7;                               // (a multiple of 8: the XCD-aware order of the full panels)
  *n_full = nf;
  const int u = *unit = panel_unit_rows(rows, ncu);
  return (unsigned)nf + (unsigned)((rows - 32 * (int64_t)nf + u - 1) / u);
}

constexpr int BN_TAB = 1024;                  // slots the LDS table of (mean, rstd) holds
constexpr float BN_FWD_EPS = 1e-5f;
constexpr double BN_FIX = 1099511627776.0;    // 2^40

__device__ __forceinline__ unsigned long long rg_pack_max(float val, unsigned r) {      // = pack_max of readout_body.h
  return ((unsigned long long)f32_ordered(val) << 32) | (unsigned long long)(0xFFFFFFFFu - r);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

constexpr int GN = 8;                 // neighbour rows per output row that are gathered in one go (the rest, rare, follow)
constexpr int LDA_F = 128;            // row stride of the gathered full-K A panel (K <= 128).  No padding: float4 column c of row r
                                      // lives at column c ^ (r & 7), which spreads eight rows over the 32 banks (conflict-free
                                      // 16-byte fragment reads) and keeps the split-K kernel at 80 KiB = two blocks per CU

// KS = 2 (GATHER, NT <= 4, K > KC): 512 threads, the K range is split between two groups of four waves.  A block of the
// KS = 1 kernel keeps ONE wave per SIMD busy with a chain of dependent phases (index trip -> row gather -> K loop ->
// epilogue), so a CU idles through every memory wait; with two wave groups the gather prologue is shared by twice the
// lanes (half the rows per lane), each group runs half of the K chunks on its own LDS stages while the other group's
// waits are covered, and group 1 hands its accumulators to group 0 through LDS before the (unchanged) epilogue.
template <int NT, bool TRANS_B, bool GATHER, int KS = 1, bool READOUT = false, bool BNIN = false, bool STATS = false, bool UNITS = false>
__device__ __forceinline__ void rowgemm_body(const RowGemmArgs& g, float* smem_all, unsigned bid) {
  static_assert(KS == 1 || (KS == 2 && GATHER && NT <= 4), "the split-K variant is built for the gather kernel, widths <= 128");
  static_assert(!READOUT || (NT <= 4 && KS == 1), "the readout epilogue is built for one column tile per wave");
  static_assert(!BNIN || (GATHER && KS == 1), "batch-norm on the fly lives in the gather prologue of the one-group kernel");
  static_assert(!STATS || NT <= 4, "the statistics epilogue is built for one column tile per wave");
  constexpr int NP = 32 * NT;
  constexpr int TPW = (NT + 3) / 4;
  constexpr int BV = NT;                               // float4 of B per thread per chunk (KC * NP / 1024)
  constexpr int A_FLOATS = GATHER ? 0 : 32 * LDA_S;   // GATHER: the A operand is the gathered panel, no A stage
  constexpr int B_FLOATS = TRANS_B ? NP * LDA_S : KC * NP;
  constexpr int STAGE = A_FLOATS + B_FLOATS;           // one of the two LDS stages
  const int tid_all = threadIdx.x;
  const int grp = KS == 2 ? (tid_all >> 8) : 0;          // wave group (uniform per wave)
  const int tid = KS == 2 ? (tid_all & 255) : tid_all, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  // K split: both groups run the same number of chunk iterations (Kc, the barriers match); Kv = what is valid of it
  const int Kc = KS == 1 ? g.K : ((g.K + 2 * KC - 1) / (2 * KC)) * KC;
  const int kb = grp * Kc;
  const int Kv = KS == 1 ? g.K : min(g.K - kb, Kc);
  constexpr int GRP = (KS == 2 && 2 * STAGE < 4096 + 256) ? 4096 + 256 : 2 * STAGE;   // LDS floats per wave group
  float* smem = smem_all + grp * GRP;                    // this group's two stages
  // common to the block: the gathered panel, and 256 floats of epilogue scratch (KS = 2: inside group 1's idle stages,
  // after the accumulator exchange buffer)
  float* Apanel = smem_all + KS * GRP + (KS == 2 ? 0 : 256);
  float* scratch = KS == 2 ? smem_all + GRP + 4096 : smem_all + 2 * STAGE;
  // XCD-aware panel order: blocks b, b+8, ... share an XCD (and its L2); give each XCD one contiguous run of panels so
  // that the neighbour rows gathered by adjacent panels (same graph) are fetched into one L2 only.  The filler block
  // (last) keeps its index.
  const unsigned npanels = (unsigned)((g.rows + 31) / 32);
  // UNITS (compile time: the plain path keeps its code — the same checks against a per-block row bound cost every panel kernel
  // 0.2-0.6 us when they were decided at run time): this block's rows are [m0, rows_hi); plain: rows_hi = the batch's row count
  int64_t m0, rows_hi;
  if constexpr (UNITS) {
    const int un = g.unit == 8 ? 8 : 16;
    const unsigned nfull = (unsigned)g.n_full, nhalf = (unsigned)((g.rows - 32 * (int64_t)nfull + un - 1) / un);
    if (bid < nfull) { m0 = (int64_t)xcd_remap(bid, nfull) * 32; rows_hi = m0 + 32; }
    else if (bid < nfull + nhalf) { m0 = 32 * (int64_t)nfull + un * (int64_t)xcd_remap(bid - nfull, nhalf); rows_hi = min(m0 + un, g.rows); }
    else { m0 = (int64_t)npanels * 32; rows_hi = m0; }   // the filler block
  } else {
    m0 = (bid < npanels ? (int64_t)xcd_remap(bid, npanels) : (int64_t)bid) * 32;
    rows_hi = g.rows;
  }
  TR(0);
  if (m0 >= g.rows) {
    // filler block (launched after the panels when fill_rows > 0): every fill row = [normalised] bias
    if (KS == 2 && grp == 1) return;
    float* fred = smem_all;
    const int N4 = g.N / 4, rpp = 256 / N4;              // rows per pass
    const int c4 = tid % N4, rsub = tid / N4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && rsub < rpp) bv = ldg4(g.bias + 4 * c4);
    float ss = rsub == 0 ? (bv.x * bv.x + bv.y * bv.y) + (bv.z * bv.z + bv.w * bv.w) : 0.f;
    ss = wave_sum(ss);
    if (lane == 0) fred[wid] = ss;
    __syncthreads();
    float sc = 1.f;
    if (g.normalize) sc = fminf(__builtin_amdgcn_rsqf((fred[0] + fred[1]) + (fred[2] + fred[3])), 1.0f / NORM_EPS);
    const float4 out = make_float4(bv.x * sc, bv.y * sc, bv.z * sc, bv.w * sc);
    if constexpr (STATS) {
      if (g.st_ghost && wid == 0) {                      // the ghost row's two numbers (all of row 0's float4 live in wave 0: N4 <= 32)
        const float a = fmaxf(out.x, 0.f), b = fmaxf(out.y, 0.f), c = fmaxf(out.z, 0.f), d = fmaxf(out.w, 0.f);
        float g1 = rsub == 0 ? (a + b) + (c + d) : 0.f, g2 = rsub == 0 ? (a * a + b * b) + (c * c + d * d) : 0.f;
        g1 = wave_sum(g1); g2 = wave_sum(g2);
        if (lane == 0) { g.st_ghost[0] = g1; g.st_ghost[1] = g2; }
      }
    }
    if (rsub < rpp) {
      for (int64_t r = rsub; r < g.fill_rows; r += rpp) {
        *reinterpret_cast<float4*>(g.c + (g.rows + r) * g.ldc + 4 * c4) = out;
        if (g.rinv && c4 == 0) g.rinv[g.rows + r] = sc;
      }
      if constexpr (READOUT) {
        if (g.ro_packed && g.ro_nghost) {                // every graph's first ghost row takes part in its readout
          for (int b = rsub; b < g.ro_B; b += rpp) {
            const int sz = g.ro_graph_ptr[b + 1] - g.ro_graph_ptr[b];
            if (sz < g.ro_nslots && sz < g.fill_rows) {
              const unsigned row = (unsigned)(g.rows + sz);
              unsigned long long* dst = g.ro_packed + (int64_t)b * g.N + 4 * c4;
              atomicMax(dst + 0, rg_pack_max(out.x, row)); atomicMax(dst + 1, rg_pack_max(out.y, row));
              atomicMax(dst + 2, rg_pack_max(out.z, row)); atomicMax(dst + 3, rg_pack_max(out.w, row));
            }
          }
        }
      }
    }
    return;
  }
  int ro_gf = 0, ro_gl = 0;                            // READOUT: first / last graph of this panel (loaded early, used last)
  if constexpr (READOUT) {
    if (g.ro_packed) {
      ro_gf = g.ro_row_graph[m0];
      ro_gl = g.ro_row_graph[min(m0 + 31, rows_hi - 1)];
    }
  }

  // staging maps (computed once) ------------------------------------------------------------------
  // A: thread -> row am, floats 4*ak4..+3 of the chunk.  B: float4 q of the thread -> (k, n4) or (n, k4).
  // Loads are unconditional from clamped (always mapped) addresses; validity is applied when the registers
  // are written to LDS, so nothing waits on a load before the MFMAs of the current chunk.
  const int am = tid >> 3, ak4 = tid & 7;
  const bool a_row_ok = (m0 + am) < rows_hi;
  const float* ap = g.a + ((a_row_ok && !GATHER) ? (m0 + am) : 0) * g.lda + 4 * ak4 + (GATHER ? 0 : kb);
  const float* bp[BV];
  int b_k[BV];                                         // k (or first k of the float4) inside the chunk
  int b_lds[BV];
  bool b_nok[BV];
#pragma unroll
  for (int q = 0; q < BV; ++q) {
    const int idx = q * 256 + tid;
    if (!TRANS_B) {
      const int k = idx / (NP / 4), n4 = idx % (NP / 4);
      b_nok[q] = 4 * n4 < g.N;                         // N % 4 == 0 on this path
      bp[q] = g.b + (int64_t)(k + kb) * g.ldb + (b_nok[q] ? 4 * n4 : 0);
      b_k[q] = k;
      b_lds[q] = k * NP + 4 * n4;
    } else {
      const int n = idx / (KC / 4), k4 = idx % (KC / 4);
      b_nok[q] = n < g.N;                              // K % 4 == 0 on this path
      bp[q] = g.b + (int64_t)(b_nok[q] ? n : 0) * g.ldb + 4 * k4 + kb;
      b_k[q] = 4 * k4;
      b_lds[q] = n * LDA_S + 4 * k4;
    }
  }
  struct Staged {                                      // one K chunk on its way global -> registers -> LDS
    float4 ra;
    float4 rb[BV];
    int a_valid;                                       // number of valid floats of ra (0..4)
    unsigned b_valid;                                  // bit q: rb[q] valid
    bool plain;                                        // uniform: no masking needed
  };
  const bool panel_full = (m0 + 32) <= rows_hi && g.N == NP;   // uniform: every row and column of the panel exists
  auto fetch = [&](Staged& s, int k0) {                // G(c): global -> registers
    s.plain = panel_full && (k0 + KC) <= Kv;
    s.a_valid = 0;
    s.b_valid = 0;
    if (GATHER) {                                      // the A panel is already in LDS: only B travels
#pragma unroll
      for (int q = 0; q < BV; ++q) {
        const bool kok = (k0 + b_k[q]) < Kv;
        if (!TRANS_B) s.rb[q] = ldg4(kok ? bp[q] + (int64_t)k0 * g.ldb : bp[q] - (int64_t)b_k[q] * g.ldb);
        else s.rb[q] = ldg4(kok ? bp[q] + k0 : bp[q] - b_k[q]);
        s.b_valid |= (kok && b_nok[q]) ? (1u << q) : 0u;
      }
      s.plain = false;
      return;
    }
    if (s.plain) {
      s.ra = ldg4(ap + k0);
#pragma unroll
      for (int q = 0; q < BV; ++q) s.rb[q] = ldg4(TRANS_B ? bp[q] + k0 : bp[q] + (int64_t)k0 * g.ldb);
      return;
    }
    const int gk = k0 + 4 * ak4;
    const bool ok = a_row_ok && gk < Kv;
    s.a_valid = ok ? min(4, Kv - gk) : 0;
    s.ra = ldg4(gk < Kv ? ap + k0 : ap - 4 * ak4);
#pragma unroll
    for (int q = 0; q < BV; ++q) {
      const bool kok = (k0 + b_k[q]) < Kv;
      if (!TRANS_B) s.rb[q] = ldg4(kok ? bp[q] + (int64_t)k0 * g.ldb : bp[q] - (int64_t)b_k[q] * g.ldb);
      else s.rb[q] = ldg4(kok ? bp[q] + k0 : bp[q] - b_k[q]);
      s.b_valid |= (kok && b_nok[q]) ? (1u << q) : 0u;
    }
  };
  auto commit = [&](const Staged& s, float* st) {      // S(c): registers -> LDS stage
    if (GATHER) {
#pragma unroll
      for (int q = 0; q < BV; ++q)
        *reinterpret_cast<float4*>(st + A_FLOATS + b_lds[q]) = ((s.b_valid >> q) & 1u) ? s.rb[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    if (s.plain) {
      *reinterpret_cast<float4*>(st + am * LDA_S + 4 * ak4) = s.ra;
#pragma unroll
      for (int q = 0; q < BV; ++q) *reinterpret_cast<float4*>(st + A_FLOATS + b_lds[q]) = s.rb[q];
      return;
    }
    float4 va = s.ra;
    if (s.a_valid < 4) va.w = 0.f;
    if (s.a_valid < 3) va.z = 0.f;
    if (s.a_valid < 2) va.y = 0.f;
    if (s.a_valid < 1) va.x = 0.f;
    *reinterpret_cast<float4*>(st + am * LDA_S + 4 * ak4) = va;
#pragma unroll
    for (int q = 0; q < BV; ++q)
      *reinterpret_cast<float4*>(st + A_FLOATS + b_lds[q]) = ((s.b_valid >> q) & 1u) ? s.rb[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto frags = [&](const float* st, int k0, float (&af)[KC / 2], float (&bf)[TPW][KC / 2]) {   // R(c): LDS -> MFMA operands
    const float* As = GATHER ? Apanel : st;
    constexpr int lda_s = GATHER ? LDA_F : LDA_S;
    const int kq = GATHER ? (kb + k0) >> 2 : 0;          // first float4 column of the chunk in the panel (multiple of 8)
    const float* Bs = st + A_FLOATS;
#pragma unroll
    for (int u = 0; u < KC / 8; ++u) {
      const float4 v = GATHER ? *reinterpret_cast<const float4*>(As + i * lda_s + 4 * ((kq + 2 * u + h) ^ (i & 7)))
                              : *reinterpret_cast<const float4*>(As + i * lda_s + 8 * u + 4 * h);
      af[4 * u] = v.x; af[4 * u + 1] = v.y; af[4 * u + 2] = v.z; af[4 * u + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wid + 4 * t;
      if (4 * (t + 1) <= NT || tile < NT) {             // compile-time true for full groups of 4 tiles
#pragma unroll
        for (int u = 0; u < KC / 8; ++u) {
          if (!TRANS_B) {
#pragma unroll
            for (int c = 0; c < 4; ++c) bf[t][4 * u + c] = Bs[(8 * u + 4 * h + c) * NP + tile * 32 + i];
          } else {                                      // W row n = output column: same 16-byte fragment read as A
            const float4 v = *reinterpret_cast<const float4*>(Bs + (tile * 32 + i) * LDA_S + 8 * u + 4 * h);
            bf[t][4 * u] = v.x; bf[t][4 * u + 1] = v.y; bf[t][4 * u + 2] = v.z; bf[t][4 * u + 3] = v.w;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < KC / 2; ++j) bf[t][j] = 0.f;
      }
    }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // Software pipeline over the K chunks c = 0, 1, ...:  G(c) global->registers, S(c) registers->LDS stage c&1,
  // R(c) LDS->operand registers, M(c) the MFMA chain.  Iteration c runs  R(c+1) | M(c) | S(c+2) | G(c+4)  and ends on
  // the only barrier: operands of the next chunk are already in registers when M(c) finishes, and a chunk's global
  // loads have two iterations to land.  Stage c&1 is rewritten in iteration c, one barrier after its readers R(c).
  // NS register staging sets: with 4 (widths <= 128) the first four chunks are all in flight before any MFMA.
  constexpr int NS = (NT <= 4 && !GATHER) ? 4 : 2;    // (the gather prologue needs the registers for the neighbour rows)
  Staged st[NS];
  constexpr int NPASS = 4 / KS;                        // gather passes of 8 * KS rows
  int ids[NPASS][GN];                                  // GATHER: neighbour ids first (head of the dependent chain), the W
  if (GATHER) {                                        // fetches below fill their latency
    const int rsub = tid_all >> 5;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int64_t row = m0 + 8 * KS * p + rsub;
#pragma unroll
      for (int q = 0; q < GN / 4; ++q) {
        int4 v = make_int4(-1, -1, -1, -1);
        if (row < rows_hi && 4 * q < g.ell_w) v = *reinterpret_cast<const int4*>(g.ell + row * g.ell_w + 4 * q);
        ids[p][4 * q] = v.x; ids[p][4 * q + 1] = v.y; ids[p][4 * q + 2] = v.z; ids[p][4 * q + 3] = v.w;
      }
    }
  }
  // BNIN: the previous layer's integer sums (16 bytes a slot) and slot counts, requested right behind the neighbour ids
  constexpr int BN_PT = BN_TAB / 256;                  // slots per thread
  float2* bn_tab = reinterpret_cast<float2*>(Apanel + 32 * LDA_F);
  ulonglong2 bn_s[BNIN ? BN_PT : 1];
  int bn_have[BNIN ? BN_PT : 1];
  float bn_g1 = 0.f, bn_g2 = 0.f;
  if constexpr (BNIN) {
#pragma unroll
    for (int q = 0; q < BN_PT; ++q) {
      const int n = min(tid_all + 256 * q, g.bn_nslots - 1);
      bn_s[q] = *reinterpret_cast<const ulonglong2*>(g.bn_sums + 2 * n);
      bn_have[q] = g.bn_slot_count[n];
    }
    bn_g1 = g.bn_ghost[0]; bn_g2 = g.bn_ghost[1];
  }
  int st_slot = -1;                                    // STATS: the slot of row m0 + lane (wave 0, lanes < 32), used last
  if constexpr (STATS) {
    if (g.st_sums && wid == 0 && grp == 0 && lane < 32 && (m0 + lane) < rows_hi) st_slot = g.st_row_slot[m0 + lane];
  }
#pragma unroll
  for (int c = 0; c < NS; ++c)
    if (c == 0 || c * KC < Kc) fetch(st[c], c * KC);
  if (GATHER) {
    // A panel = aggregated rows.  Row-major like the stand-alone aggregation kernel: 32 lanes per row (one float4 column
    // each), 8 rows per pass, 4 passes; the first GN neighbour rows of all four passes are in flight together
    // (one index round trip + one row round trip for the whole panel), longer lists (rare) are finished afterwards.
    const int c4 = tid_all & 31, rsub = tid_all >> 5;
    const bool colok = 4 * c4 < g.K;                   // a float4 that straddles K is taken whole: B's rows >= K are zero in
                                                       // LDS and the row padding of x is finite (zero) by the layout rule
    if constexpr (BNIN) {
      // (mean, rstd) of every slot BEFORE the rows are requested: the sums arrived with the ids, and their registers are dead by
      // the time the 32 row requests per lane go out (with both live the kernel needed > 256 registers: one block per CU, and the
      // readout blocks of the same launch could no longer ride beside the row panels)
      const double inv_cnt = 1.0 / ((double)g.bn_B * (double)g.bn_F);
#pragma unroll
      for (int q = 0; q < BN_PT; ++q) {
        const int n = tid_all + 256 * q;
        if (n < g.bn_nslots) {
          const float2 ms = bn_stats_from_sums(bn_s[q].x, bn_s[q].y, bn_have[q], bn_g1, bn_g2, g.bn_B, inv_cnt, BN_FWD_EPS);
          bn_tab[n] = make_float2(ms.y, ms.x * ms.y);        // (rstd, mean * rstd):  y = relu(v) * rstd - mean * rstd
        }
      }
    }
    float4 nbv[NPASS][GN];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
      for (int k = 0; k < GN; ++k) {
        nbv[p][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (colok && ids[p][k] >= 0) nbv[p][k] = ldg4(g.a + (int64_t)(BNIN ? (ids[p][k] & 0xFFFFF) : ids[p][k]) * g.lda + 4 * c4);
      }
    if constexpr (BNIN) {
      __syncthreads();                                 // the table of (rstd, mean * rstd) is complete
      // y = (relu(v) - mean) rstd = fma(relu(v), rstd, -mean rstd): one max + one fma per element
      // (summing relu(v) rstd first and subtracting sum_j mean_j rstd_j once per row measured SLOWER: 14.0 -> 17.1 us)
#pragma unroll
      for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int k = 0; k < GN; ++k) {
          if (ids[p][k] >= 0) {
            const float2 rm = bn_tab[ids[p][k] >> 20];
            float4& t = nbv[p][k];
            t.x = fmaf(fmaxf(t.x, 0.f), rm.x, -rm.y); t.y = fmaf(fmaxf(t.y, 0.f), rm.x, -rm.y);
            t.z = fmaf(fmaxf(t.z, 0.f), rm.x, -rm.y); t.w = fmaf(fmaxf(t.w, 0.f), rm.x, -rm.y);
          }
        }
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int64_t row = m0 + 8 * KS * p + rsub;
      float4 va = nbv[p][0];
#pragma unroll
      for (int k = 1; k < GN; ++k) { va.x += nbv[p][k].x; va.y += nbv[p][k].y; va.z += nbv[p][k].z; va.w += nbv[p][k].w; }
      bool table_full = g.ell_w <= GN && ids[p][GN - 1] >= 0;  // only a row whose table is full can continue in the CSR tail
      if (colok && g.ell_w > GN && ids[p][GN - 1] >= 0) {      // the table fills from the left: maybe more than GN neighbours
        // the second half of the row's table entries in ONE 32-byte request (same cache line as the first half), then ALL of their
        // rows in one round trip from clamped addresses, added in table order (the same sum, bit for bit, as a loop that fetched
        // entry k, then its row, then entry k + 1 ...: two dependent trips per extra neighbour)
        static_assert(GN == 8, "the second half of a 16-wide table");
        const int4 e0 = *reinterpret_cast<const int4*>(g.ell + row * g.ell_w + 8), e1 = *reinterpret_cast<const int4*>(g.ell + row * g.ell_w + 12);
        const int js[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        table_full = js[7] >= 0;
        float4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = ldg4(g.a + (int64_t)(js[k] >= 0 ? (BNIN ? (js[k] & 0xFFFFF) : js[k]) : 0) * g.lda + 4 * c4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (js[k] >= 0) {                                  // (entries fill from the left: the first negative one ends the list)
            if constexpr (BNIN) {
              const float2 rm = bn_tab[js[k] >> 20];
              t[k].x = fmaf(fmaxf(t[k].x, 0.f), rm.x, -rm.y); t[k].y = fmaf(fmaxf(t[k].y, 0.f), rm.x, -rm.y);
              t[k].z = fmaf(fmaxf(t[k].z, 0.f), rm.x, -rm.y); t[k].w = fmaf(fmaxf(t[k].w, 0.f), rm.x, -rm.y);
            }
            va.x += t[k].x; va.y += t[k].y; va.z += t[k].z; va.w += t[k].w;
          }
        }
      }
      if (colok && g.tail_ptr && row < rows_hi && table_full) {   // lists longer than the table continue in the CSR tail (a handful
                                                                 // of rows per batch: asking every row with >= GN neighbours for its
                                                                 // tail range was a dependent round trip for 7 % of the rows)
        for (int e = g.tail_ptr[row]; e < g.tail_ptr[row + 1]; ++e) {
          const int j = g.tail_col[e];                     // BNIN: slot << 20 | row, like the table's entries
          float4 t = ldg4(g.a + (int64_t)(BNIN ? (j & 0xFFFFF) : j) * g.lda + 4 * c4);
          if constexpr (BNIN) {
            const float2 rm = bn_tab[j >> 20];
            t.x = fmaf(fmaxf(t.x, 0.f), rm.x, -rm.y); t.y = fmaf(fmaxf(t.y, 0.f), rm.x, -rm.y);
            t.z = fmaf(fmaxf(t.z, 0.f), rm.x, -rm.y); t.w = fmaf(fmaxf(t.w, 0.f), rm.x, -rm.y);
          }
          va.x += t.x; va.y += t.y; va.z += t.z; va.w += t.w;
        }
      }
      {                                                  // zeros beyond K / rows; swizzled column (see LDA_F)
        const int pr = 8 * KS * p + rsub;
        *reinterpret_cast<float4*>(Apanel + pr * LDA_F + 4 * (c4 ^ (pr & 7))) = va;
      }
      if (g.zout && colok && row < rows_hi) st_out(reinterpret_cast<float4*>(g.zout + row * g.ldz + 4 * c4), va);
    }
  }
  float bias_v[TPW];                                   // fetched now, used in the epilogue
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int cn = (wid + 4 * t) * 32 + i;
    bias_v[t] = (g.bias && (wid + 4 * t) < NT && cn < g.N) ? g.bias[cn] : 0.f;
  }
  TR(1);
  commit(st[0], smem);
  if (KC < Kc) commit(st[1], smem + STAGE);
  __syncthreads();
  TR(2);
  float fa[2][KC / 2], fb[2][TPW][KC / 2];
  frags(smem, 0, fa[0], fb[0]);
  if (NS == 2) {
    if (2 * KC < Kc) fetch(st[0], 2 * KC);
    if (3 * KC < Kc) fetch(st[1], 3 * KC);
  }
  auto body = [&](auto ci_, int c) {
    constexpr int CI = decltype(ci_)::value;           // c mod 4, compile time: register sets are picked statically
    constexpr int P = CI & 1;
    const int k0 = c * KC;
    if (k0 + KC < Kc) frags(smem + (P ^ 1) * STAGE, k0 + KC, fa[P ^ 1], fb[P ^ 1]);
    if (NS == 4 && c >= 0 && k0 + 4 * KC < Kc) fetch(st[CI % NS], k0 + 4 * KC);
    __builtin_amdgcn_sched_barrier(0);                 // the scheduler would sink the LDS reads to their uses
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[P][j], fb[P][0][j], acc[0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (k0 + 2 * KC < Kc) commit(st[(CI + 2) % NS], smem + P * STAGE);
    if (NS == 2 && k0 + 4 * KC < Kc) fetch(st[CI % NS], k0 + 4 * KC);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 8; j < KC / 2; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[P][j], fb[P][0][j], acc[0], 0, 0, 0);
#pragma unroll
    for (int t = 1; t < TPW; ++t) {
      if (4 * (t + 1) <= NT || (wid + 4 * t) < NT) {
#pragma unroll
        for (int j = 0; j < KC / 2; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[P][j], fb[P][t][j], acc[t], 0, 0, 0);
      }
    }
    TR(3 + 2 * min(c, 3));
    if (Kc > 2 * KC) __syncthreads();                  // stage reuse only exists beyond two chunks (both were committed before
    TR(4 + 2 * min(c, 3));                             // the loop): without it the two K groups of a panel run uncoupled
  };
  for (int c = 0; c * KC < Kc; c += 4) {
    body(std::integral_constant<int, 0>{}, c);
    if ((c + 1) * KC < Kc) body(std::integral_constant<int, 1>{}, c + 1);
    if ((c + 2) * KC < Kc) body(std::integral_constant<int, 2>{}, c + 2);
    if ((c + 3) * KC < Kc) body(std::integral_constant<int, 3>{}, c + 3);
  }
  if (KS == 2) {
    // group 1 hands its partial sums over through its own (now idle) stages and retires; group 0 finishes the panel
    float* xch = smem_all + GRP;                       // 4 waves x 16 registers x 64 lanes = 16 KiB, inside group 1's region
    if (grp == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(wid * 16 + r) * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    TR(7);
    if (grp == 1) { TR_END(); return; }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] += xch[(wid * 16 + r) * 64 + lane];
    TR(8);
  }

  // epilogue in registers: lane (i, h) of tile `wid + 4t` holds C[(r&3) + 8(r>>2) + 4h][tile*32 + i] in acc[t][r].
  float scale[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) scale[r] = 1.f;
  if (g.bias || g.normalize) {
    float ss[16];
    float q1[STATS ? 16 : 1], q2[STATS ? 16 : 1];      // STATS: sum_f relu(u), sum_f relu(u)^2 of the un-normalised row
#pragma unroll
    for (int r = 0; r < 16; ++r) ss[r] = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wid + 4 * t;
      const int cn = tile * 32 + i;
      const bool okc = tile < NT && cn < g.N;
      const float bv = bias_v[t];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = okc ? acc[t][r] + bv : 0.f;
        acc[t][r] = v;
        ss[r] = fmaf(v, v, ss[r]);
        if constexpr (STATS) { const float rl = fmaxf(v, 0.f); q1[r] = rl; q2[r] = rl * rl; }
      }
    }
    if (g.normalize) {
      // row sums of squares: transposing DPP reduction inside each 16-lane row (lane l ends with row r = l & 15 of its
      // half), the two rows of a half through one bpermute, the four waves (column tiles) through LDS; then each wave
      // turns the 32 totals into 1/max(|u|, eps) with one v_rsq per lane and hands them out through LDS.
      // STATS: the batch-norm statistics of the NORMALISED row ride in the same two barriers — relu(u * s) = s * relu(u) for the
      // positive row scale s, so  sum_f relu(v) = s * sum_f relu(u)  and  sum_f relu(v)^2 = s^2 * sum_f relu(u)^2 : the two extra
      // row sums travel through the (idle) first LDS stage, and ONE pair of 64-bit integer atomics per row follows.
      float* red = scratch;                            // [32 rows][4 waves]
      float* inv = red + 128 + wid * 32;               // per wave [32 rows]
      float* redq = smem_all;                          // [2][32 rows][4 waves]: the K loop's stages are idle once every wave has left it
      if constexpr (STATS) { if (Kc <= 2 * KC) __syncthreads(); }      // (the loop only synchronises beyond two chunks)
      float tot = row16_sum_transpose(ss);
      tot += __shfl_xor(tot, 16, 64);
      float t1 = 0.f, t2 = 0.f;
      if constexpr (STATS) {
        t1 = row16_sum_transpose(q1); t2 = row16_sum_transpose(q2);
        t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
      }
      TR(11);
      if ((lane & 16) == 0) {
        const int r = lane & 15, row = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[row * 4 + wid] = tot;
        if constexpr (STATS) { redq[row * 4 + wid] = t1; redq[128 + row * 4 + wid] = t2; }
      }
      __syncthreads();
      TR(12);
      if (lane < 32) {
        const float4 p = *reinterpret_cast<const float4*>(red + lane * 4);
        const float rs = fminf(__builtin_amdgcn_rsqf((p.x + p.y) + (p.z + p.w)), 1.0f / NORM_EPS);
        inv[lane] = rs;
        if (g.rinv && wid == 0 && (m0 + lane) < rows_hi) g.rinv[m0 + lane] = rs;
        if constexpr (STATS) {
          if (g.st_sums && wid == 0 && st_slot >= 0) {
            const float4 a = *reinterpret_cast<const float4*>(redq + lane * 4), b = *reinterpret_cast<const float4*>(redq + 128 + lane * 4);
            const float s1 = ((a.x + a.y) + (a.z + a.w)) * rs, s2 = ((b.x + b.y) + (b.z + b.w)) * (rs * rs);
            atomicAdd(g.st_sums + 2 * st_slot, (unsigned long long)__double2ll_rn((double)s1 * BN_FIX));
            atomicAdd(g.st_sums + 2 * st_slot + 1, (unsigned long long)__double2ll_rn((double)s2 * BN_FIX));
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(inv + 8 * q + 4 * h);
        scale[4 * q] = v.x; scale[4 * q + 1] = v.y; scale[4 * q + 2] = v.z; scale[4 * q + 3] = v.w;
      }
    }
  }
  const bool full = panel_full;                        // uniform: no per-element predicates on the common path
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wid + 4 * t;
    if (tile < NT) {
      const int cn = tile * 32 + i;
      float* cp = g.c + (m0 + 4 * h) * g.ldc + cn;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st_out(cp + (int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc, acc[t][r] * scale[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (gm < rows_hi && cn < g.N) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = acc[t][r] * scale[r];
        }
      }
    }
  }
  if constexpr (READOUT) {
    if (g.ro_packed) {
      // lane (i, h) of wave `wid` holds column cn of 16 of the panel's rows: fold them per graph, meet the other half of
      // the wave, one 64-bit atomicMax per (graph, column) and panel.  Values are the stored ones (acc * scale), so the
      // readout is bitwise what a pass over c would find; ties go to the smallest row (~row in the low word).
      const int cn = wid * 32 + i;
      const int64_t last = min(m0 + 31, rows_hi - 1);
      for (int b = ro_gf; b <= ro_gl; ++b) {
        int64_t lo = m0, hi = last + 1;
        if (ro_gf != ro_gl) { lo = max(lo, (int64_t)g.ro_graph_ptr[b]); hi = min(hi, (int64_t)g.ro_graph_ptr[b + 1]); }
        unsigned long long best = 0ull;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (gm >= lo && gm < hi) {
            const unsigned long long p = rg_pack_max(acc[0][r] * scale[r], (unsigned)gm);
            best = p > best ? p : best;
          }
        }
        const unsigned olo = __shfl_xor((unsigned)(best & 0xFFFFFFFFull), 32, 64), ohi = __shfl_xor((unsigned)(best >> 32), 32, 64);
        const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
        best = other > best ? other : best;
        if (h == 0 && best && cn < g.N) atomicMax(&g.ro_packed[(int64_t)b * g.N + cn], best);
      }
    }
  }
  TR(13);
  TR_END();
}


template <int NT, bool TRANS_B, bool GATHER, int KS = 1, bool BNIN = false>
constexpr size_t rowgemm_lds_bytes() {
  constexpr int NP = 32 * NT;
  constexpr int STAGE2 = 2 * ((GATHER ? 0 : 32 * LDA_S) + (TRANS_B ? NP * LDA_S : KC * NP));
  constexpr int GRP = (KS == 2 && STAGE2 < 4096 + 256) ? 4096 + 256 : STAGE2;
  return sizeof(float) * (KS * GRP + (KS == 2 ? 0 : 128 + 4 * 32) + (GATHER ? 32 * LDA_F : 0) + (BNIN ? 2 * BN_TAB : 0));
}

}  // namespace
