// Body of the fused PyG-style SAGEConv layer (sageconv.hip), as a device function so that other kernels can run it beside
// independent work in one launch.
//
//   out[i, :] = act( [ dst_i * sum_{j in N(i)} xg[j, :]  ||  xs[i, :] ] . [ W_l ; W_r ] + bias )
//
// forward  (SAGEConv: lin_l(mean_j x_j) + lin_r(x_i)):  xg = xs = x, dst = 1 / max(deg, 1), W = nn.Linear weights [out, in]
// backward (dx = A_mean^T (du W_l) + du W_r, symmetric A): xg = du / max(deg, 1) (rows scaled by their own degree), xs = du,
//          W = the same weights read [k][n]
//
// One 512-thread workgroup owns a 32-row panel.  Design points (what the row-panel product of rowgemm_body.h taught):
//   * the B operand never goes through LDS: wave w of a group holds ITS 32 output columns of W for the whole K in registers
//     (64 VGPRs at K = 128) — the K loop has no barrier and no staging.  The registers are filled from a FRAGMENT-MAJOR copy of the
//     weights (tsgnn_sage_conv_pack_f32: [wave][step u][lane] float4, zero padded), so that every wave-instruction reads 1 KB of
//     consecutive bytes: straight from nn.Linear's layout a lane reads ITS row of W (32 rows x 32 B per instruction, four tag
//     look-ups per 128-byte line) and the per-CU memory pipe, not latency, set the pace (measured: 8.5 k cycles until the first
//     index came back behind those loads; 20 k behind the 64 dword loads per lane of the [K][N] form);
//   * request ORDER = arrival order (a wave's loads return in order): group 1 asks for self rows, W_r, then its neighbour ids; group 0
//     for its ids, the neighbour rows, and only then W_l (needed last);
//   * two wave groups with ROLES instead of a split K: group 1 owns the self half (its rows need no index: one round trip), group 0
//     the aggregated half (index -> rows: two dependent trips); group 1's MFMA chain runs while the neighbour rows are in flight,
//     so the chain on the critical path is one half, not two (both groups' waves share each SIMD's matrix pipe);
//   * both groups share the gather (16 rows per pass, two passes: 64 row registers per lane instead of 128);
//   * the A panel [32][agg 128 | self 128] lives in LDS, float4 column c of row r at c ^ (r & 15) inside its aligned group of 16:
//     the 16 lanes of a ds_read_b128 group then hit 16 different float4 columns (all 64 banks once).
#pragma once
#include "common.h"

namespace {

typedef float sc_f32x16 __attribute__((ext_vector_type(16)));

struct SageConvArgs {
  const float* xg; int64_t ldxg;      // rows that are gathered through the neighbour table
  const float* xs; int64_t ldxs;      // rows read in place (the root / self term)
  const int* ell; int ell_w;          // fixed-width neighbour table [rows][ell_w], entries < 0 empty, filled from the left
  const int* tail_ptr; const int* tail_col;   // nullable: CSR of the neighbours beyond ell_w
  const float* dst_scale;             // nullable: the sum of row i is multiplied by dst_scale[i]
  const float4* wl_pk;                // weights of the aggregated half, fragment-major (tsgnn_sage_conv_pack_f32): [4][16][64] float4
  const float4* wr_pk;                // weights of the self half
  const float* bias;                  // nullable [N]
  float* out; int64_t ldo;            // [rows, N]
  float* zout; int64_t ldz;           // nullable: the scaled aggregate [rows, K] (the weight gradient's operand)
  float* rinv;                        // nullable (normalize): 1 / max(|row|, eps)
  int64_t rows; int K, N;             // K <= 128 per half, N <= 128
  int relu_out, normalize;
  // readout epilogue (nullable): per-graph column max of the OUTPUT rows as packed (ordered value, ~row) 64-bit atomicMax into
  // ro_packed[B, N], and the column sums as 64-bit fixed-point integers (2^-32 units) into ro_sums[B, N]
  unsigned long long* ro_packed; unsigned long long* ro_sums;
  const int* ro_row_graph; const int* ro_graph_ptr;
  // post epilogue (nullable; the input-gradient launch of layer l + 1 finishing layer l's dU): with v = the product's value,
  //   out[r, c] = ( v + [post_arg[b, c] == r] post_dread[b, c] + post_dread[b, N + c] / n_b ) * [post_h[r, c] > 0]     b = graph of row r
  //   out2[r, c] = out[r, c] * post_row_scale[r]                      (nullable: the rows the NEXT input-gradient launch gathers)
  // i.e. tsgnn_sage_relu_readout_bwd_f32 without a launch of its own.  Uses ro_row_graph / ro_graph_ptr.
  const float* post_h; int64_t post_ldh;
  const float* post_dread; int64_t post_lddr; const int* post_arg;
  const float* post_row_scale; float* out2; int64_t ldo2;
};

constexpr int SC_LDP = 256;           // floats per panel row: [agg 128 | self 128]
constexpr float SC_NORM_EPS = 1e-12f;
constexpr double SC_RO_FIX = 4294967296.0;     // 2^32

__device__ __forceinline__ float4 sc_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 sc_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void sc_fma4(float4& a, const float4& v, float s) {
  a.x = fmaf(v.x, s, a.x); a.y = fmaf(v.y, s, a.y); a.z = fmaf(v.z, s, a.z); a.w = fmaf(v.w, s, a.w);
}
__device__ __forceinline__ void sc_add4(float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
__device__ __forceinline__ unsigned long long sc_pack_max(float val, unsigned r) {
  return ((unsigned long long)f32_ordered(val) << 32) | (unsigned long long)(0xFFFFFFFFu - r);
}

constexpr size_t sageconv_lds_bytes() { return sizeof(float) * (32 * SC_LDP + 4 * 16 * 64 + 256); }

// smem: [panel 32 x 256][xch 4 waves x 16 x 64][scratch 256]
__device__ __forceinline__ void sageconv_body(const SageConvArgs& g, float* smem, unsigned bid) {
  float* P = smem;
  float* xch = smem + 32 * SC_LDP;
  float* scratch = xch + 4 * 16 * 64;
  const int tid_all = threadIdx.x;
  // 0: aggregated half, 1: self half.  Wave-uniform AND known to be so by the compiler (an SGPR): the two groups take different
  // branches with barriers inside them, which is only sound when a wave executes exactly one side.
  const int grp = __builtin_amdgcn_readfirstlane(tid_all >> 8);
  const int tid = tid_all & 255, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const unsigned npanels = (unsigned)((g.rows + 31) / 32);
  const int64_t m0 = (int64_t)xcd_remap(bid, npanels) * 32;
  const int K = g.K, N = g.N;
  TR(0);

  // gather map (both groups): 32 lanes per row (one float4 column each), 16 rows per pass, two passes
  const int c4 = tid_all & 31, rsub = tid_all >> 5;
  const int nvc = min(4, max(0, K - 4 * c4));            // valid floats of this lane's float4 column
  const bool wide = g.ell_w > 8;                          // uniform: the table has a second half (neighbours 9-16)
  int ids[2][16];
  float dsc[2];
  float4 bw4[16];                                        // this wave's 32 output columns of its group's weights, whole K: lane (i, h)
                                                         // holds bw4[u] = W[k = 8u + 4h + 0..3][n = 32 wid + i]
  const float4* wpk = (grp ? g.wr_pk : g.wl_pk) + wid * 16 * 64 + lane;
  auto load_ids = [&]() {                                // the WHOLE table row (one 64-byte line) at once
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int64_t row = m0 + 16 * p + rsub;
      const bool rok = row < g.rows;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int4 v = make_int4(-1, -1, -1, -1);
        if (rok && 4 * q < g.ell_w) v = *reinterpret_cast<const int4*>(g.ell + row * g.ell_w + 4 * q);
        ids[p][4 * q] = v.x; ids[p][4 * q + 1] = v.y; ids[p][4 * q + 2] = v.z; ids[p][4 * q + 3] = v.w;
      }
      dsc[p] = (g.dst_scale && rok) ? g.dst_scale[row] : 1.f;
    }
  };
  auto load_w = [&]() {
#pragma unroll
    for (int u = 0; u < 16; ++u) bw4[u] = (8 * u < K) ? wpk[u * 64] : sc_zero4();
  };
  float4 win[2][8];                                      // neighbour rows in flight: entries 0-7 of both passes, then 8-15
  auto issue_rows = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = ids[p][k0 + k];
        win[p][k] = (j >= 0 && nvc > 0) ? sc_ld4(g.xg + (int64_t)j * g.ldxg + 4 * c4) : sc_zero4();
      }
  };
  float4 va[2];
  auto accumulate = [&]() {                              // (empty entries: zero rows — the sum runs in table order)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int k = 0; k < 8; ++k) sc_add4(va[p], win[p][k]);
  };
  auto commit_agg = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int pr = 16 * p + rsub;
      const int64_t row = m0 + pr;
      float4 v = va[p];
      if (g.tail_ptr && row < g.rows && ids[p][g.ell_w - 1] >= 0) {     // a FULL table may continue in the CSR tail (rare)
        for (int e = g.tail_ptr[row]; e < g.tail_ptr[row + 1]; ++e) {
          const int j = g.tail_col[e];
          const float4 t = nvc > 0 ? sc_ld4(g.xg + (int64_t)j * g.ldxg + 4 * c4) : sc_zero4();
          sc_add4(v, t);
        }
      }
      v.x *= dsc[p]; v.y *= dsc[p]; v.z *= dsc[p]; v.w *= dsc[p];
      if (nvc < 4) v.w = 0.f;
      if (nvc < 3) v.z = 0.f;
      if (nvc < 2) v.y = 0.f;
      if (nvc < 1) v.x = 0.f;
      *reinterpret_cast<float4*>(P + pr * SC_LDP + 4 * (c4 ^ (pr & 15))) = v;
      if (g.zout && nvc > 0 && row < g.rows) st_out(reinterpret_cast<float4*>(g.zout + row * g.ldz + 4 * c4), v);
    }
  };
  va[0] = sc_zero4(); va[1] = sc_zero4();

  sc_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool wave_on = 32 * wid < N;                      // uniform: this wave's column tile exists
  // steps u0 .. u1 - 1 (8 k each) of one half's chain, in pairs; the next pair's A fragments are read under this pair's MFMAs
  auto kloop = [&](int half, int u0, int u1) {
    // A fragments: lane (i, h), step u: the float4 at float column 8u + 4h of row i (= float4 column 2u + h)
    const float* prow = P + i * SC_LDP + half * 128;
    float4 af[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) af[(u0 >> 1) & 1][q] = *reinterpret_cast<const float4*>(prow + 4 * ((2 * (u0 + q) + h) ^ (i & 15)));
#pragma unroll
    for (int up = 0; up < 8; ++up) {                      // pair up = steps 2 up, 2 up + 1
      if (2 * up >= u0 && 2 * up < u1 && 16 * up < K) {
        if (2 * (up + 1) < u1 && 16 * (up + 1) < K) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
            af[(up + 1) & 1][q] = *reinterpret_cast<const float4*>(prow + 4 * ((2 * (2 * (up + 1) + q) + h) ^ (i & 15)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int u = 2 * up + q;
          if (8 * u < K) {
            const float4 a = af[up & 1][q];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw4[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw4[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw4[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw4[u].w, acc, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (grp == 1) {
    // ---- group 1: self rows (all 32: 8 rows per pass, four passes), its neighbour ids, W_r — in the order they are needed ----
    const int c4s = tid & 31, rs = tid >> 5;
    const int nvs = min(4, max(0, K - 4 * c4s));
    float4 selfv[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t row = m0 + 8 * p + rs;
      selfv[p] = (row < g.rows && nvs > 0) ? sc_ld4(g.xs + row * g.ldxs + 4 * c4s) : sc_zero4();
    }
    load_ids();
    load_w();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int pr = 8 * p + rs;
      float4 v = selfv[p];
      if (nvs < 4) v.w = 0.f;
      if (nvs < 3) v.z = 0.f;
      if (nvs < 2) v.y = 0.f;
      if (nvs < 1) v.x = 0.f;
      *reinterpret_cast<float4*>(P + pr * SC_LDP + 4 * (32 + (c4s ^ (pr & 15)))) = v;
    }
    TR_AFTER(__float_as_int(selfv[3].x), 7);
    __syncthreads();                                      // #1: the self half is complete (group 0 arrived long ago)
    TR(1);
    TR_AFTER(ids[0][0], 6);
    issue_rows(0);
    TR_AFTER(__float_as_int(bw4[15].w), 8);
    // the self half's chain runs while the neighbour rows fly; half way through, the first eight neighbours of every row have
    // arrived: they are summed and the requests for neighbours 9-16 take their registers — back by the end of the chain
    if (wave_on) kloop(1, 0, 8);
    TR_AFTER(__float_as_int(win[1][7].x), 9);
    accumulate();
    if (wide) issue_rows(8);
    if (wave_on) kloop(1, 8, 16);
    TR(2);
    if (wide) accumulate();
    commit_agg();
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[(wid * 16 + r) * 64 + lane] = acc[r];     // the self half's partial sums, for group 0
    __syncthreads();                                      // #2
    TR(3);
    TR_END();
    return;
  }
  // ---- group 0: neighbour ids first; it never reads the self half, so barrier #1 is only an arrival, made while the ids travel ----
  load_ids();
  const float bias_v = (g.bias && (32 * wid + i) < N) ? g.bias[32 * wid + i] : 0.f;
  int ro_gf = 0, ro_gl = -1;
  if (g.ro_packed || g.post_h) {
    ro_gf = g.ro_row_graph[m0];
    ro_gl = g.ro_row_graph[min(m0 + 31, g.rows - 1)];
  }
  __syncthreads();                                        // #1
  TR(1);
  TR_AFTER(ids[0][0], 6);
  issue_rows(0);
  load_w();                                               // W_l is needed last: requested behind the neighbour rows
  TR_AFTER(__float_as_int(win[1][7].x), 9);
  accumulate();
  if (wide) { issue_rows(8); accumulate(); }
  commit_agg();
  TR_AFTER(__float_as_int(bw4[15].w), 8);
  __syncthreads();                                        // #2: the aggregated half and group 1's partial sums are complete
  TR(3);
  const int cn = 32 * wid + i;
  const bool okc = cn < N;
  float hpost[16], rsc[16];                               // post epilogue operands, requested before the chain (its registers: the
  constexpr int PG = 3;                                   // neighbour rows' that are dead by now); the first PG graphs of the panel
  float pdm[PG], pds[PG]; int pa[PG], plo[PG], phi[PG];
  if (g.post_h) {
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int b = min(ro_gf + q, ro_gl);                // (clamped: always a valid graph)
      plo[q] = g.ro_graph_ptr[b]; phi[q] = g.ro_graph_ptr[b + 1];
      pdm[q] = okc ? g.post_dread[(int64_t)b * g.post_lddr + cn] : 0.f;
      pds[q] = okc ? g.post_dread[(int64_t)b * g.post_lddr + N + cn] : 0.f;
      pa[q] = okc ? g.post_arg[(int64_t)b * N + cn] : -1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const bool ok = gm < g.rows && okc;
      hpost[r] = ok ? g.post_h[gm * g.post_ldh + cn] : 0.f;
      rsc[r] = (g.out2 && gm < g.rows) ? g.post_row_scale[gm] : 0.f;
    }
  }
  if (wave_on) kloop(0, 0, 16);
  TR(4);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += xch[(wid * 16 + r) * 64 + lane];

  // ---- epilogue in registers: lane (i, h) of wave wid holds C[(r & 3) + 8 (r >> 2) + 4 h][32 wid + i] in acc[r] ---------
  float scale[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = okc ? acc[r] + bias_v : 0.f;
    if (g.relu_out) v = fmaxf(v, 0.f);
    acc[r] = v;
    scale[r] = 1.f;
  }
  if (g.post_h) {
    // the readout gradients of the rows' graphs (a panel touches one to a few graphs), then the ReLU mask of the layer's output
    const int64_t last = min(m0 + 31, g.rows - 1);
#pragma unroll
    for (int q = 0; q < PG; ++q) {                        // the panel's first PG graphs: their operands arrived during the chain
      if (ro_gf + q <= ro_gl) {
        const int64_t lo = max(m0, (int64_t)plo[q]), hi = min(last + 1, (int64_t)phi[q]);
        const float ds = pds[q] * (1.0f / (float)max(phi[q] - plo[q], 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (gm >= lo && gm < hi) acc[r] += ((int64_t)pa[q] == gm ? pdm[q] : 0.f) + ds;
        }
      }
    }
    for (int b = ro_gf + PG; b <= ro_gl; ++b) {           // panels of many small graphs: the rest, one trip per graph
      const int64_t glo = g.ro_graph_ptr[b], ghi = g.ro_graph_ptr[b + 1];
      const int64_t lo = max(m0, glo), hi = min(last + 1, ghi);
      const float inv_n = 1.0f / (float)max((int)(ghi - glo), 1);
      const float dm = okc ? g.post_dread[(int64_t)b * g.post_lddr + cn] : 0.f;
      const float ds = okc ? g.post_dread[(int64_t)b * g.post_lddr + N + cn] * inv_n : 0.f;
      const int a = okc ? g.post_arg[(int64_t)b * N + cn] : -1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (gm >= lo && gm < hi) acc[r] += ((int64_t)a == gm ? dm : 0.f) + ds;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = hpost[r] > 0.f ? acc[r] : 0.f;
    if (g.out2 && wave_on) {
      float* cp2 = g.out2 + (m0 + 4 * h) * g.ldo2 + cn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (gm < g.rows && okc) cp2[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldo2] = acc[r] * rsc[r];
      }
    }
  }
  if (g.normalize) {                                      // F.normalize(out, p = 2, dim = -1) (SAGEConv(normalize=True))
    float ss[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ss[r] = acc[r] * acc[r];
    float* red = scratch;                                 // [32 rows][4 waves]
    float* inv = scratch + 128;                           // [32 rows]
    float tot = row16_sum_transpose(ss);
    tot += __shfl_xor(tot, 16, 64);
    if ((lane & 16) == 0) {
      const int r = lane & 15, row = (r & 3) + 8 * (r >> 2) + 4 * h;
      red[row * 4 + wid] = tot;
    }
    __syncthreads();                                      // (group 1 has retired: the barrier counts the live waves)
    if (tid < 32) {
      const float4 p = *reinterpret_cast<const float4*>(red + tid * 4);
      const float rs = fminf(__builtin_amdgcn_rsqf((p.x + p.y) + (p.z + p.w)), 1.0f / SC_NORM_EPS);
      inv[tid] = rs;
      if (g.rinv && (m0 + tid) < g.rows) g.rinv[m0 + tid] = rs;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(inv + 8 * q + 4 * h);
      scale[4 * q] = v.x; scale[4 * q + 1] = v.y; scale[4 * q + 2] = v.z; scale[4 * q + 3] = v.w;
    }
  }
  if (wave_on) {
    float* cp = g.out + (m0 + 4 * h) * g.ldo + cn;
    if ((m0 + 32) <= g.rows && okc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st_out(cp + (int64_t)((r & 3) + 8 * (r >> 2)) * g.ldo, acc[r] * scale[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (gm < g.rows && okc) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldo] = acc[r] * scale[r];
      }
    }
    if (g.ro_packed) {
      // readouts of the OUTPUT rows, per graph of the panel: column max as one packed atomicMax, column sum as one 64-bit
      // fixed-point atomicAdd (integer addition is associative: the totals do not depend on the order the panels finish in)
      const int64_t last = min(m0 + 31, g.rows - 1);
      for (int b = ro_gf; b <= ro_gl; ++b) {
        int64_t lo = m0, hi = last + 1;
        if (ro_gf != ro_gl) { lo = max(lo, (int64_t)g.ro_graph_ptr[b]); hi = min(hi, (int64_t)g.ro_graph_ptr[b + 1]); }
        unsigned long long best = 0ull;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (gm >= lo && gm < hi) {
            const float v = acc[r] * scale[r];
            const unsigned long long pk = sc_pack_max(v, (unsigned)gm);
            best = pk > best ? pk : best;
            sum += v;
          }
        }
        const unsigned olo = __shfl_xor((unsigned)(best & 0xFFFFFFFFull), 32, 64), ohi = __shfl_xor((unsigned)(best >> 32), 32, 64);
        const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
        best = other > best ? other : best;
        const long long q = __double2ll_rn((double)sum * SC_RO_FIX);
        const long long qo = ((long long)__shfl_xor((int)(q >> 32), 32, 64) << 32) | (unsigned)__shfl_xor((int)(q & 0xFFFFFFFFll), 32, 64);
        if (h == 0 && okc) {
          if (best) atomicMax(&g.ro_packed[(int64_t)b * N + cn], best);
          atomicAdd(&g.ro_sums[(int64_t)b * N + cn], (unsigned long long)(q + qo));
        }
      }
    }
  }
  TR(5);
  TR_END();
}

}  // namespace
