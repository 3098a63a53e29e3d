// One GAT layer (all heads) in two launches forward and three backward (encoders_GAT.py:29-49, 68-84).
//
// Operand layout.  The heads' parameters are packed into ONE projection matrix with 2H extra columns
//     W' = [ W_0 | W_1 | ... | W_{H-1} | W_0 a1_0 ... W_{H-1} a1_{H-1} | W_0 a2_0 ... W_{H-1} a2_{H-1} | 0 pad ]      [Fin, Ns]
// so that hp = x . W' carries, per node, its projected features (C = H*Fh columns) AND the two attention scalars of every head
// (encoders_GAT.py:35-36: e_ij = LeakyReLU(a1 . h_i + a2 . h_j)): s_row = hp[:, C + h], s_col = hp[:, C + H + h].  The reference's
// [N, N, 2F] pair tensor, the dense [N, N] attention matrix and even the per-edge alpha array never exist here; neither do
// the separate score / score-gradient passes: d(s_row), d(s_col) are two more columns of dhp and flow into dW' and dx through the
// same two products as the features; the chain rule back to (W_h, a_h) is a [Fin, Ns]-sized epilogue (gat_unpack).
//
//   forward   gat_attn_fwd     one wave per ROW i:  alpha_ij = softmax over the COLUMN j's entries (dim=1, trap T3) recomputed on the
//                              fly from the scalars (a column has ~5 entries), out_i = sum_j alpha_ij h_j + uniform term of the
//                              edge-less columns (their all-masked softmax is 1/N, :38-41), mean over heads / ELU in registers.
//   backward  gat_attn_bwd     one wave per COLUMN j: dalpha_ij = <dpre_i, h_j>, softmax + LeakyReLU backward, dh_j = sum_i alpha_ij
//                              dpre_i, d(s_col)_j; per-entry terms of d(s_row) in A^T order.  dpre = dy * ELU'(y) is formed on the fly.
//                              One extra workgroup per graph sums dpre over the graph's rows for the uniform term's gradient.
//             gat_score_rowsum d(s_row)_i = sum over row i's entries (tiny).
// Attention dropout (encoders_GAT.py:42, F.dropout on the attention MATRIX, edge-less columns' 1/N entries included) is a
// counter-based Philox4x32-10 mask keyed on (seed; row i, column j, head): forward and backward regenerate the same bits.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

__device__ __forceinline__ float lrelu_(float t, float slope) { return t > 0.f ? t : slope * t; }
__device__ __forceinline__ float4 ldg4_(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float elu_grad_y(float y) { return y > 0.f ? 1.f : y + 1.f; }      // d/dx elu(x) = exp(x) = elu(x) + 1 for x <= 0

struct Drop {
  unsigned thresh;     // an element is dropped when its 32 random bits are < thresh (= p * 2^32); 0: no dropout
  float scale;         // 1 / (1 - p)
  unsigned seed_lo, seed_hi;
  const unsigned long long* ctr;   // nullable: a DEVICE counter added to the key when the kernel starts (a step replayed from a
};                                 // hipGraph advances it, so every replay draws a new mask; the backward reads the same value)
__device__ __forceinline__ Drop drop_key(Drop d) {
  if (d.thresh != 0u && d.ctr) {
    const unsigned long long c = *d.ctr;
    d.seed_lo += (unsigned)(c & 0xffffffffull);
    d.seed_hi += (unsigned)(c >> 32);
  }
  return d;
}
// multiplier of attention element (i, j) of head h: 0 or 1 / (1 - p)
__device__ __forceinline__ float drop_mult(const Drop& d, unsigned i, unsigned j, int h) {
  if (d.thresh == 0u) return 1.f;
  const uint4 r = philox4x32_10(make_uint4(i, j, (unsigned)(h >> 2), 0x47415431u), make_uint2(d.seed_lo, d.seed_hi));
  const unsigned v = (h & 3) == 0 ? r.x : (h & 3) == 1 ? r.y : (h & 3) == 2 ? r.z : r.w;
  return v < d.thresh ? 0.f : d.scale;
}

// resident waves per SIMD the two attention kernels are compiled for (<= 80 VGPRs at 6): a DD batch is ~8,500 one-wave rows,
// each a chain of 4-5 dependent round trips, so the kernel time is (rounds of resident waves) x (chain latency)
#ifndef GAT_WAVES_PER_SIMD
#define GAT_WAVES_PER_SIMD 6
#endif
#ifndef GAT_BWD_WAVES_PER_SIMD
#define GAT_BWD_WAVES_PER_SIMD 5          /* four entries' dy and y rows in flight: 6 spills nine registers */
#endif

// ---------------------------------------------------------------- column statistics
// stat[j, h] = (m, 1 / Z) of column j: m = max_i e_ij, Z = sum_i exp(e_ij - m) over the column's entries i (rows of A^T),
// e_ij = LeakyReLU(s_row[i] + s_col[j]).  Eight lanes per (column, head): entry ids, then their scalars — two dependent round
// trips per column; the first entry of every lane stays in registers.  Edge-less columns get (0, 0).
__global__ __launch_bounds__(256) void gat_col_stats_kernel(const float* __restrict__ hp, int64_t ldh, const int* __restrict__ rp_t,
                                                            const int* __restrict__ col_t, int64_t rows, int H, int C, float slope,
                                                            float2* __restrict__ stat) {
  const int sub = threadIdx.x & 7;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 8;
  if (i >= rows * H) return;
  const int64_t j = i / H;
  const int h = (int)(i % H);
  const int t0 = rp_t[j], t1 = rp_t[j + 1];
  if (t0 == t1) {
    if (sub == 0) stat[i] = make_float2(0.f, 0.f);
    return;
  }
  const float scol = hp[j * ldh + C + H + h];
  const float* srow = hp + C + h;
  const int tf = t0 + sub;
  const bool has = tf < t1;
  const float ef = has ? lrelu_(srow[(int64_t)col_t[tf] * ldh] + scol, slope) : -INFINITY;
  float m = ef;
  for (int t = tf + 8; t < t1; t += 8) m = fmaxf(m, lrelu_(srow[(int64_t)col_t[t] * ldh] + scol, slope));
  m = group_max<8>(m);
  float z = has ? __expf(ef - m) : 0.f;
  for (int t = tf + 8; t < t1; t += 8) z += __expf(lrelu_(srow[(int64_t)col_t[t] * ldh] + scol, slope) - m);
  z = group_sum<8>(z);
  if (sub == 0) stat[i] = make_float2(m, 1.f / z);
}

// ---------------------------------------------------------------- forward: one wave per row
struct GatFwd {
  const float* hp; int64_t ldh;
  const int* rowptr; const int* col;          // A   (row i: its columns j)
  const float2* stat;                         // [rows, H] column statistics (gat_col_stats_kernel)
  int64_t rows; int H, Fh; float slope;
  const int* row_graph; int nmax;             // graph of row r: row_graph[r] (ragged batches) or r / nmax
  const int* iso_idx; const float* iso_w; const int* iso_ptr;   // edge-less columns of every graph (ids, multiplicities); nullable
  float uscale;                               // 1 / N
  int mean_heads, apply_elu;
  Drop drop;
  float* y; int64_t ldy;
};

template <int LPH, bool DROP>   // lanes per head = Fh / 4; DROP: attention dropout compiled in
__global__ __launch_bounds__(256, GAT_WAVES_PER_SIMD) void gat_attn_fwd_kernel(GatFwd a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t r = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wid;     // neighbouring rows share gathered rows: one L2
  if (r >= a.rows) return;
  const int H = a.H, C = H * a.Fh, G = H * LPH;
  const bool live = lane < G;
  const int h = live ? lane / LPH : 0;
  const int co = live ? 4 * lane : 0;
  const float* __restrict__ hp = a.hp;
  const int64_t ldh = a.ldh;
  if (DROP) a.drop = drop_key(a.drop);
  TR(0);
  const int e0 = a.rowptr[r], e1 = a.rowptr[r + 1];
  const int EB = min(8, 64 / H);              // entries per batch: lane p = (entry p / H, head p % H) computes one alpha
  const int pk = lane / H, ph = lane - pk * H;
  const float srow_i = hp[r * ldh + C + ph];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

  int q0 = 0, q1 = 0;
  if (a.iso_ptr) {
    const int b = a.row_graph ? a.row_graph[r] : (int)(r / a.nmax);
    q0 = a.iso_ptr[b]; q1 = a.iso_ptr[b + 1];
  }
  int jq = 0;
  float wq = 0.f;
  if (q0 < q1) { jq = a.iso_idx[q0]; wq = a.iso_w[q0]; }    // (the first listed column: requested with the edge chain)
  TR(1);
  for (int eb = e0; eb < e1; eb += EB) {
    const int cnt = min(EB, e1 - eb);                       // uniform over the wave
    const bool has = pk < cnt;
    const int j = a.col[has ? eb + pk : eb];
    TR_AFTER(j, 2);                                         // column ids arrived
    float4 v[8];
    // the batch's feature rows, requested before the statistics chain.  All eight requests are unconditional (entries beyond the
    // row's end repeat its first column; their alpha is 0): behind `if (k < cnt)` every request sat in its own block behind an
    // s_waitcnt vmcnt(0) — eight DEPENDENT round trips per row
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int jk = __shfl(j, (k < cnt ? k : 0) * H, 64);
      v[k] = ldg4_(hp + (int64_t)jk * ldh + co);
    }
    const float scol_j = hp[(int64_t)j * ldh + C + H + ph];
    const float2 st = a.stat[(int64_t)j * H + ph];           // (m, 1 / Z) of column j
    float alpha = 0.f;
    if (has) {
      alpha = __expf(lrelu_(srow_i + scol_j, a.slope) - st.x) * st.y;
      if (DROP) alpha *= drop_mult(a.drop, (unsigned)r, (unsigned)j, ph);
    }
    TR_AFTER(__float_as_int(alpha), 3);                     // statistics and scalars arrived
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < cnt) {                                        // (arithmetic only: uniform, no memory operation inside)
        const float al = __shfl(alpha, k * H + h, 64);
        acc.x = fmaf(al, v[k].x, acc.x); acc.y = fmaf(al, v[k].y, acc.y);
        acc.z = fmaf(al, v[k].z, acc.z); acc.w = fmaf(al, v[k].w, acc.w);
      }
    }
  }

  // uniform term: the edge-less columns j of this row's graph, alpha_ij = 1/N each (a ghost representative counts Nmax - n_b times).
  // Its list pointers are requested before the edge loop (q0, q1 above), its rows after it: the two chains overlap.
  if (q0 < q1) {                                            // the first listed column (its id came in with the edge chain)
    const float4 vq = ldg4_(hp + (int64_t)jq * ldh + co);
    const float wk = (DROP ? drop_mult(a.drop, (unsigned)r, (unsigned)jq, h) : 1.f) * (wq * a.uscale);
    acc.x = fmaf(wk, vq.x, acc.x); acc.y = fmaf(wk, vq.y, acc.y); acc.z = fmaf(wk, vq.z, acc.z); acc.w = fmaf(wk, vq.w, acc.w);
  }
  if (q0 + 1 < q1) {
    for (int q = q0 + 1; q < q1; q += 4) {
      int jj[4];
      float w[4];
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int qq = min(q + k, q1 - 1);
        jj[k] = a.iso_idx[qq];
        w[k] = (q + k < q1) ? a.iso_w[qq] * a.uscale : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = ldg4_(hp + (int64_t)jj[k] * ldh + co);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float wk = DROP ? w[k] * drop_mult(a.drop, (unsigned)r, (unsigned)jj[k], h) : w[k];
        acc.x = fmaf(wk, v[k].x, acc.x); acc.y = fmaf(wk, v[k].y, acc.y);
        acc.z = fmaf(wk, v[k].z, acc.z); acc.w = fmaf(wk, v[k].w, acc.w);
      }
    }
  }

  TR_AFTER(__float_as_int(acc.x), 4);                       // feature rows arrived, sums done
  if (a.mean_heads) {                                       // (h_0 + h_1 + ...) / H, then ELU (encoders_GAT.py:78-83)
    float4 s = acc;
    for (int k = 1; k < H; ++k) {
      const int src = lane + k * LPH < G ? lane + k * LPH : lane;
      s.x += __shfl(acc.x, src, 64); s.y += __shfl(acc.y, src, 64); s.z += __shfl(acc.z, src, 64); s.w += __shfl(acc.w, src, 64);
    }
    const float Hf = (float)H;
    s.x /= Hf; s.y /= Hf; s.z /= Hf; s.w /= Hf;
    if (a.apply_elu) {
      s.x = s.x <= 0.f ? expm1f(s.x) : s.x; s.y = s.y <= 0.f ? expm1f(s.y) : s.y;
      s.z = s.z <= 0.f ? expm1f(s.z) : s.z; s.w = s.w <= 0.f ? expm1f(s.w) : s.w;
    }
    if (lane < LPH) *reinterpret_cast<float4*>(a.y + r * a.ldy + 4 * lane) = s;
  } else {
    if (a.apply_elu) {
      acc.x = acc.x <= 0.f ? expm1f(acc.x) : acc.x; acc.y = acc.y <= 0.f ? expm1f(acc.y) : acc.y;
      acc.z = acc.z <= 0.f ? expm1f(acc.z) : acc.z; acc.w = acc.w <= 0.f ? expm1f(acc.w) : acc.w;
    }
    if (live) *reinterpret_cast<float4*>(a.y + r * a.ldy + co) = acc;
  }
  TR(5);
  TR_END();
}

// ---------------------------------------------------------------- backward: one wave per column (+ workgroups per graph)
struct GatBwd {
  const float* hp; int64_t ldh;
  const float* y; int64_t ldy;                // the layer's output (ELU' from it)
  const float* dy; int64_t lddy;
  const int* rp_t; const int* col_t;
  const float2* stat;
  int64_t rows; int H, Fh; float slope;
  int mean_heads, apply_elu;
  const int* graph_ptr; int B;                // rows of graph b: [graph_ptr[b], graph_ptr[b+1])
  const int* iso_idx; const float* iso_w; const int* iso_ptr; const float* iso_row; int iso_row_ld;   // nullable together
  float uscale;
  Drop drop;
  float* dhp; int Ns;                         // [rows, ldh]: dh | d s_row (gat_score_rowsum) | d s_col | zero pad up to Ns
  float* t1; float* t2;                       // [nnz, H] in A^T entry order: lrelu' alpha dalpha, lrelu' alpha
  float* S;                                   // [rows, H]: sum_i alpha_ij dalpha_ij of column j
  float* dupart; int P;                       // [B, P, C]: partial sums of dpre over the P row ranges of every graph
  unsigned ngraph_blocks;                     // B * P (0 without listed edge-less columns)
  // nullable: the layer's output feeds ONLY the max readout over each graph's rows (the LAST layer, encoders_GAT.py:189): dy is not
  // a tensor then but dy[i, c] = (ro_arg[b, c] == i) ? ro_dout[b, c] : 0 with b = row_graph[i] — formed on the fly, so the pass that
  // would scatter the readout's gradient into a [rows, C] tensor and the gathers of its rows here both disappear
  const float* ro_dout; int64_t ro_ldo; const int* ro_arg; const int* row_graph;
};
// the readout's (winning row, gradient) of graph b for this lane's four output columns
struct RoPair { int4 win; float4 g; };
template <int LPH>
__device__ __forceinline__ RoPair ro_pair(const GatBwd& a, int b, int lane, int co) {
  const int c = a.mean_heads ? 4 * (lane % LPH) : co;
  const int Co = a.mean_heads ? a.Fh : a.H * a.Fh;
  RoPair r;
  r.win = *reinterpret_cast<const int4*>(a.ro_arg + (int64_t)b * Co + c);
  r.g = ldg4_(a.ro_dout + (int64_t)b * a.ro_ldo + c);
  return r;
}

// dpre of row i for this lane's four features: dy * ELU'(y) (concat) or (dy * ELU'(y)) / H of the lane's slot (mean over heads).
// Split in two so that a caller can issue the requests of SEVERAL rows before the arithmetic of the first one waits for them
// (requests and arithmetic interleaved row by row were one dependent round trip per row).  Requests are unconditional — a lane
// beyond the row's width reads column 0 and is masked at the end.
template <int LPH>
__device__ __forceinline__ void dpre_load(const GatBwd& a, int64_t i, int lane, int co, float4& d, float4& yv, const RoPair& ro) {
  const int c = a.mean_heads ? 4 * (lane % LPH) : co;
  if (a.ro_arg) {                                            // (uniform) the readout's gradient: no request at all
    const int ii = (int)i;
    d = make_float4(ro.win.x == ii ? ro.g.x : 0.f, ro.win.y == ii ? ro.g.y : 0.f, ro.win.z == ii ? ro.g.z : 0.f, ro.win.w == ii ? ro.g.w : 0.f);
  } else {
    d = ldg4_(a.dy + i * a.lddy + c);
  }
  yv = ldg4_(a.y + i * a.ldy + c);
}
__device__ __forceinline__ float4 dpre_finish(const GatBwd& a, float4 d, const float4& yv, bool live) {
  if (a.apply_elu) {
    d.x *= elu_grad_y(yv.x); d.y *= elu_grad_y(yv.y); d.z *= elu_grad_y(yv.z); d.w *= elu_grad_y(yv.w);
  }
  if (!live) d = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.mean_heads) {
    if ((a.H & (a.H - 1)) == 0) {                            // 1 / H is exact: the product equals the quotient
      const float r = 1.f / (float)a.H;
      d.x *= r; d.y *= r; d.z *= r; d.w *= r;
    } else {
      const float Hf = (float)a.H;
      d.x /= Hf; d.y /= Hf; d.z /= Hf; d.w /= Hf;
    }
  }
  return d;
}
template <int LPH>
__device__ __forceinline__ float4 dpre_row(const GatBwd& a, int64_t i, int lane, int co, bool live, const RoPair& ro) {
  float4 d, yv;
  dpre_load<LPH>(a, i, lane, co, d, yv, ro);
  return dpre_finish(a, d, yv, live);
}

// per-column accumulators of the backward
struct ColAcc {
  float4 dh; float S, P1, P2;
};
// one entry (i, j): its share of dh_j, S_j, d s_col[j] and the per-entry terms of d s_row[i]
template <int LPH, bool DROP>
__device__ __forceinline__ void bwd_entry(const GatBwd& a, ColAcc& c, const float4& hj, const float4& dp, float sr, float scol, float m,
                                          float rZ, int i, int64_t j, int t, int h, bool writer) {
  const float tt = sr + scol;
  const float al = __expf(lrelu_(tt, a.slope) - m) * rZ;
  float dal = group_sum<LPH>((dp.x * hj.x + dp.y * hj.y) + (dp.z * hj.z + dp.w * hj.w));
  float ae = al;
  if (DROP) {
    const float mlt = drop_mult(a.drop, (unsigned)i, (unsigned)j, h);
    dal *= mlt; ae *= mlt;
  }
  c.dh.x = fmaf(ae, dp.x, c.dh.x); c.dh.y = fmaf(ae, dp.y, c.dh.y); c.dh.z = fmaf(ae, dp.z, c.dh.z); c.dh.w = fmaf(ae, dp.w, c.dh.w);
  const float lr = tt > 0.f ? 1.f : a.slope;
  const float w1 = lr * al * dal, w2 = lr * al;
  c.S = fmaf(al, dal, c.S);
  c.P1 += w1; c.P2 += w2;
  if (writer) {
    a.t1[(int64_t)t * a.H + h] = w1;
    a.t2[(int64_t)t * a.H + h] = w2;
  }
}

template <int LPH, bool DROP>
__global__ __launch_bounds__(256, GAT_BWD_WAVES_PER_SIMD) void gat_attn_bwd_kernel(GatBwd a) {
  __shared__ __attribute__((aligned(16))) float red[4][256];
  if (DROP) a.drop = drop_key(a.drop);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int H = a.H, C = H * a.Fh, G = H * LPH;
  const bool live = lane < G;
  const int h = live ? lane / LPH : 0;
  const int co = live ? 4 * lane : 0;
  const int64_t ldh = a.ldh;

  if (blockIdx.x < a.ngraph_blocks) {
    // gradient of the uniform term: every edge-less column j of graph b receives (w_j / N) * sum_i dpre_i over the graph's rows.
    // Block (b, part) sums its share of the rows; gat_score_rowsum adds the parts up and writes the columns.
    const int b = (int)blockIdx.x / a.P, part = (int)blockIdx.x % a.P;
    const int64_t g0 = a.graph_ptr[b], g1 = a.graph_ptr[b + 1];
    const int q0 = a.iso_ptr[b], q1 = a.iso_ptr[b + 1];
    if (q0 == q1) return;
    RoPair ro{};
    if (a.ro_arg) ro = ro_pair<LPH>(a, b, lane, co);
    if (DROP) {
      // with dropout every (i, j) element has its own mask: column by column (the reference's dense attention^T . dpre)
      if (part != 0) return;
      for (int q = q0 + wid; q < q1; q += 4) {
        const int jj = a.iso_idx[q];
        const float w = a.iso_w[q] * a.uscale;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t i = g0; i < g1; ++i) {
          const float mlt = drop_mult(a.drop, (unsigned)i, (unsigned)jj, h);
          const float4 d = dpre_row<LPH>(a, i, lane, co, live, ro);
          s.x = fmaf(mlt, d.x, s.x); s.y = fmaf(mlt, d.y, s.y); s.z = fmaf(mlt, d.z, s.z); s.w = fmaf(mlt, d.w, s.w);
        }
        if (live) *reinterpret_cast<float4*>(a.dhp + (int64_t)jj * ldh + co) = make_float4(w * s.x, w * s.y, w * s.z, w * s.w);
      }
      return;
    }
    const int64_t per = (g1 - g0 + a.P - 1) / a.P;
    const int64_t r0 = g0 + part * per, r1 = min(g1, r0 + per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = r0 + wid; i < r1; i += 16) {           // four rows of this wave in flight
      float4 d[4];
      float4 yv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t ii = i + 4 * k;
        dpre_load<LPH>(a, ii < r1 ? ii : r0, lane, co, d[k], yv[k], ro);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = dpre_finish(a, d[k], yv[k], live && i + 4 * k < r1);
#pragma unroll
      for (int k = 0; k < 4; ++k) { s.x += d[k].x; s.y += d[k].y; s.z += d[k].z; s.w += d[k].w; }
    }
    *reinterpret_cast<float4*>(&red[wid][4 * lane]) = s;
    __syncthreads();
    if (wid == 0) {
      const float4 s0 = *reinterpret_cast<const float4*>(&red[0][4 * lane]), s1 = *reinterpret_cast<const float4*>(&red[1][4 * lane]);
      const float4 s2 = *reinterpret_cast<const float4*>(&red[2][4 * lane]), s3 = *reinterpret_cast<const float4*>(&red[3][4 * lane]);
      if (live)
        *reinterpret_cast<float4*>(a.dupart + (int64_t)blockIdx.x * C + co) =
            make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                        (s0.w + s1.w) + (s2.w + s3.w));
    }
    return;
  }

  const unsigned nreg = gridDim.x - a.ngraph_blocks;
  const int64_t j = (int64_t)xcd_remap(blockIdx.x - a.ngraph_blocks, nreg) * 4 + wid;
  if (j >= a.rows) return;
  const float* __restrict__ hp = a.hp;
  TR(0);
  const int t0 = a.rp_t[j], t1 = a.rp_t[j + 1];
  RoPair ro{};
  if (a.ro_arg) ro = ro_pair<LPH>(a, a.row_graph[j], lane, co);      // column j's entries are rows of j's own graph
  float4 hj = ldg4_(hp + j * ldh + co);                       // (unconditional; lanes beyond the row's width are zeroed)
  if (!live) hj = make_float4(0.f, 0.f, 0.f, 0.f);
  const float scol = hp[j * ldh + C + H + h];
  const float* srow = hp + C + h;
  const bool writer = live && (lane % LPH) == 0;
  ColAcc c{make_float4(0.f, 0.f, 0.f, 0.f), 0.f, 0.f, 0.f};
  const int deg = t1 - t0;                                   // uniform over the wave
  if (deg > 0) {
    const float2 st = a.stat[j * H + h];                     // (m, 1 / Z) of this column
    int ii[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ii[k] = a.col_t[min(t0 + k, t1 - 1)];        // the usual column: every entry id in one round trip
    TR_AFTER(ii[0], 1);                                      // entry ids arrived
    for (int tb = 0; tb < deg; tb += 4) {
      int i4[4];
      float sr[4];
      float4 dp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) i4[k] = tb < 8 ? (tb == 0 ? ii[k] : ii[4 + k]) : a.col_t[min(t0 + tb + k, t1 - 1)];
      float4 yv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {                            // twelve requests, then the arithmetic
        sr[k] = srow[(int64_t)i4[k] * ldh];
        dpre_load<LPH>(a, i4[k], lane, co, dp[k], yv[k], ro);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) dp[k] = dpre_finish(a, dp[k], yv[k], live);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (tb + k < deg) bwd_entry<LPH, DROP>(a, c, hj, dp[k], sr[k], scol, st.x, st.y, i4[k], j, t0 + tb + k, h, writer);
      TR_AFTER(__float_as_int(c.S), 2 + min(tb / 4, 2));     // a group of four entries done
    }
  }
  const bool listed = a.iso_row && deg == 0 && a.iso_row[j * a.iso_row_ld] != 0.f;     // its dh comes from the graph's sums
  if (live && !listed) *reinterpret_cast<float4*>(a.dhp + j * ldh + co) = c.dh;
  if (writer) {
    a.S[j * H + h] = c.S;
    a.dhp[j * ldh + C + H + h] = c.P1 - c.S * c.P2;          // d s_col[j] = sum_i lrelu' alpha (dalpha - S)
  }
  if (lane < a.Ns - C - 2 * H) a.dhp[j * ldh + C + 2 * H + lane] = 0.f;
  TR(5);
  TR_END();
}

// d s_row[i, h] = sum over row i's entries (i, j) of  t1[e] - t2[e] * S[j]   (the per-entry terms live in A^T entry order: eperm)
// + the last nfin blocks (one per graph): dh of the listed edge-less columns from the partial sums of dpre
struct GatFin {
  const float* dupart; int P, C; const int* iso_idx; const float* iso_w; const int* iso_ptr; float uscale; unsigned nfin;
};
__global__ __launch_bounds__(256) void gat_score_rowsum_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                               const int* __restrict__ eperm, const float* __restrict__ t1,
                                                               const float* __restrict__ t2, const float* __restrict__ S, int64_t rows,
                                                               int H, float* __restrict__ dhp, int64_t ldh, int C, GatFin f) {
  if (blockIdx.x >= gridDim.x - f.nfin) {
    const int b = (int)(blockIdx.x - (gridDim.x - f.nfin));
    const int q0 = f.iso_ptr[b], q1 = f.iso_ptr[b + 1];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (q0 == q1 || 4 * lane >= f.C) return;
    float4 du = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < f.P; ++p) {                           // fixed order: bitwise reproducible
      const float4 v = ldg4_(f.dupart + ((int64_t)b * f.P + p) * f.C + 4 * lane);
      du.x += v.x; du.y += v.y; du.z += v.z; du.w += v.w;
    }
    for (int q = q0 + wid; q < q1; q += 4) {
      const float w = f.iso_w[q] * f.uscale;
      *reinterpret_cast<float4*>(dhp + (int64_t)f.iso_idx[q] * ldh + 4 * lane) = make_float4(w * du.x, w * du.y, w * du.z, w * du.w);
    }
    return;
  }
  const int sub = threadIdx.x & 7;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 8;
  if (i >= rows * H) return;
  const int64_t r = i / H;
  const int h = (int)(i % H);
  const int e1 = rowptr[r + 1];
  float acc = 0.f;
  for (int e = rowptr[r] + sub; e < e1; e += 8) {
    const int64_t p = eperm[e];
    acc += t1[p * H + h] - t2[p * H + h] * S[(int64_t)col[e] * H + h];
  }
  acc = group_sum<8>(acc);
  if (sub == 0) dhp[r * ldh + C + h] = acc;
}

// the multipliers themselves, dense, for a block of the attention matrix (tests: the oracle applies exactly these)
__global__ void gat_dropout_mult_kernel(Drop d, int64_t i0, int64_t ni, int64_t j0, int64_t nj, int H, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ni * nj * H) return;
  const int h = (int)(t % H);
  const int64_t j = (t / H) % nj, i = t / H / nj;
  d = drop_key(d);
  out[t] = drop_mult(d, (unsigned)(i0 + i), (unsigned)(j0 + j), h);
}

// ---------------------------------------------------------------- parameters of the heads <-> W'
constexpr int GAT_LMAX = 4, GAT_HMAX = 8;
struct GatPackLayer {
  const float* w[GAT_HMAX];     // [Fin, Fo] each
  const float* a[GAT_HMAX];     // [2 Fo] each: a1 (dotted with the row node) | a2 (column node)
  float* wp;                    // [Fin, Ns] packed operand (pack) — or its gradient (unpack)
  float* gw;                    // unpack: [H, Fin, Fo]
  float* ga;                    // unpack: [H, 2 Fo]
  int H, Fin, Fo, Ns;
  int blk0;                     // first block of this layer in the launch
};
struct GatPack {
  GatPackLayer l[GAT_LMAX];
  int L;
};

// one block per (layer, input row k): copies the row of every head and appends the 2H dots W_h[k, :] . a_h
__global__ __launch_bounds__(256) void gat_pack_kernel(GatPack p) {
  __shared__ float prod[2][GAT_HMAX * 256];
  int li = 0;
#pragma unroll
  for (int t = 1; t < GAT_LMAX; ++t) li = (t < p.L && (int)blockIdx.x >= p.l[t].blk0) ? t : li;
  const GatPackLayer& y = p.l[li];
  const int k = (int)blockIdx.x - y.blk0;
  const int C = y.H * y.Fo;
  float* out = y.wp + (int64_t)k * y.Ns;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int h = c / y.Fo, f = c - h * y.Fo;
    const float wv = y.w[h][(int64_t)k * y.Fo + f];
    out[c] = wv;
    prod[0][c] = wv * y.a[h][f];
    prod[1][c] = wv * y.a[h][y.Fo + f];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < y.Ns - C; t += 256) {
    float s = 0.f;
    if (t < 2 * y.H) {
      const int which = t / y.H, h = t - which * y.H;
      for (int f = 0; f < y.Fo; ++f) s += prod[which][h * y.Fo + f];
    }
    out[C + t] = s;
  }
}

// gradients: blocks [blk0, blk0 + Fin): gw[h][k, f] = dW'[k, hFo+f] + dW'[k, C+h] a1_h[f] + dW'[k, C+H+h] a2_h[f];
//            blocks [blk0 + Fin, blk0 + Fin + H): ga[h][f] = sum_k W_h[k, f] dW'[k, C+h],  ga[h][Fo+f] = sum_k W_h[k, f] dW'[k, C+H+h]
__global__ __launch_bounds__(256) void gat_unpack_kernel(GatPack p) {
  __shared__ __attribute__((aligned(16))) float part[2][1024];
  int li = 0;
#pragma unroll
  for (int t = 1; t < GAT_LMAX; ++t) li = (t < p.L && (int)blockIdx.x >= p.l[t].blk0) ? t : li;
  const GatPackLayer& y = p.l[li];
  const int kb = (int)blockIdx.x - y.blk0;
  const int C = y.H * y.Fo;
  if (kb < y.Fin) {
    const float* dw = y.wp + (int64_t)kb * y.Ns;
    for (int c = threadIdx.x; c < C; c += 256) {
      const int h = c / y.Fo, f = c - h * y.Fo;
      y.gw[((int64_t)h * y.Fin + kb) * y.Fo + f] = dw[c] + dw[C + h] * y.a[h][f] + dw[C + y.H + h] * y.a[h][y.Fo + f];
    }
    return;
  }
  const int h = kb - y.Fin;
  // ga[h]: a [Fo x Fin] matrix-vector product per vector.  Thread (f4, kq): four features, rows k = kq, kq + KQ, ...; the KQ
  // row groups meet in LDS in a fixed order.  (Fo % 4 == 0 and Fo <= 64 on this path: the fused kernels' head shapes.)
  const int F4 = y.Fo / 4, KQ = 256 / F4;
  const int f4 = (int)threadIdx.x % F4, kq = (int)threadIdx.x / F4;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int k = kq; k < y.Fin; k += 4 * KQ) {
    float4 wv[4];
    float d1[4], d2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = min(k + u * KQ, y.Fin - 1);
      wv[u] = ldg4_(y.w[h] + (int64_t)kk * y.Fo + 4 * f4);
      d1[u] = (k + u * KQ < y.Fin) ? y.wp[(int64_t)kk * y.Ns + C + h] : 0.f;
      d2[u] = (k + u * KQ < y.Fin) ? y.wp[(int64_t)kk * y.Ns + C + y.H + h] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s1.x = fmaf(wv[u].x, d1[u], s1.x); s1.y = fmaf(wv[u].y, d1[u], s1.y); s1.z = fmaf(wv[u].z, d1[u], s1.z); s1.w = fmaf(wv[u].w, d1[u], s1.w);
      s2.x = fmaf(wv[u].x, d2[u], s2.x); s2.y = fmaf(wv[u].y, d2[u], s2.y); s2.z = fmaf(wv[u].z, d2[u], s2.z); s2.w = fmaf(wv[u].w, d2[u], s2.w);
    }
  }
  *reinterpret_cast<float4*>(&part[0][4 * threadIdx.x]) = s1;
  *reinterpret_cast<float4*>(&part[1][4 * threadIdx.x]) = s2;
  __syncthreads();
  if ((int)threadIdx.x < 2 * y.Fo) {
    const int which = (int)threadIdx.x / y.Fo, f = (int)threadIdx.x % y.Fo;
    float t = 0.f;
    for (int q = 0; q < KQ; ++q) t += part[which][4 * (q * F4 + f / 4) + (f & 3)];
    y.ga[(int64_t)h * 2 * y.Fo + which * y.Fo + f] = t;
  }
}

inline int lph_of(int Fh) { return Fh / 4; }
inline bool fused_ok(int H, int Fh) {
  const int lph = Fh / 4;
  return H >= 1 && H <= GAT_HMAX && Fh > 0 && Fh % 4 == 0 && (lph == 1 || lph == 2 || lph == 4 || lph == 8 || lph == 16) &&
         H * lph <= 64;
}

inline Drop make_drop(float p, uint64_t seed, const unsigned long long* ctr) {
  Drop d{0u, 1.f, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), ctr};
  if (p > 0.f) {
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
    if (d.thresh == 0u) d.thresh = 1u;
    d.scale = 1.f / (1.f - p);
  }
  return d;
}

inline bool unpack_desc(const int64_t* desc, GatPack& p, bool grads) {
  // desc: [L, then per layer: H, Fin, Fo, Ns, wp, gw, ga, w_0..w_7, a_0..a_7]
  if (!desc) return false;
  p.L = (int)desc[0];
  if (p.L < 1 || p.L > GAT_LMAX) return false;
  int blk = 0;
  for (int t = 0; t < p.L; ++t) {
    const int64_t* d = desc + 1 + t * (7 + 2 * GAT_HMAX);
    GatPackLayer& y = p.l[t];
    y.H = (int)d[0]; y.Fin = (int)d[1]; y.Fo = (int)d[2]; y.Ns = (int)d[3];
    y.wp = reinterpret_cast<float*>(d[4]); y.gw = reinterpret_cast<float*>(d[5]); y.ga = reinterpret_cast<float*>(d[6]);
    if (y.H < 1 || y.H > GAT_HMAX || y.Fin < 1 || y.Fo < 1 || y.H * y.Fo > GAT_HMAX * 256 || y.Ns < y.H * y.Fo + 2 * y.H || !y.wp)
      return false;
    if (grads && (!y.gw || !y.ga || (y.Fo % 4) || y.Fo > 64)) return false;
    for (int h = 0; h < GAT_HMAX; ++h) {
      y.w[h] = reinterpret_cast<const float*>(d[7 + h]);
      y.a[h] = reinterpret_cast<const float*>(d[7 + GAT_HMAX + h]);
      if (h < y.H && (!y.w[h] || !y.a[h])) return false;
    }
    y.blk0 = blk;
    blk += y.Fin + (grads ? y.H : 0);
  }
  for (int t = p.L; t < GAT_LMAX; ++t) p.l[t] = p.l[0];
  return true;
}

}  // namespace

extern "C" {

/* 1 if the fused GAT layer kernels take (H heads, Fh features per head): Fh / 4 a power of two <= 16, H * Fh <= 256, H <= 8 */
int tsgnn_gat_fused_supported(int H, int Fh) { return fused_ok(H, Fh) ? 1 : 0; }

int tsgnn_gat_attn_fwd_f32(const float* hp, int64_t ldh, const int* rowptr, const int* col, const int* rp_t, const int* col_t,
                           int64_t rows, int H, int Fh, float slope, const int* row_graph, int nmax, const int* iso_idx,
                           const float* iso_w, const int* iso_ptr, float uscale, int mean_heads, int apply_elu, float drop_p,
                           uint64_t seed, const unsigned long long* drop_ctr, float* stat, float* y, int64_t ldy, tsgnn_stream_t stream) {
  if (!hp || !rowptr || !col || !rp_t || !col_t || !y || !stat || (reinterpret_cast<uintptr_t>(stat) & 7) || rows < 0 || nmax <= 0 || drop_p < 0.f || drop_p >= 1.f) return TSGNN_EINVAL;
  if (!fused_ok(H, Fh)) return TSGNN_EUNSUPPORTED;
  const int C = H * Fh;
  if (ldh < C + 2 * H || (ldh % 4) || (ldy % 4) || ldy < (mean_heads ? Fh : C) ||
      ((reinterpret_cast<uintptr_t>(hp) | reinterpret_cast<uintptr_t>(y)) & 15))
    return TSGNN_EUNSUPPORTED;
  if ((iso_idx == nullptr) != (iso_ptr == nullptr) || (iso_idx == nullptr) != (iso_w == nullptr)) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  gat_col_stats_kernel<<<(unsigned)ceil_div64(rows * H * 8, 256), 256, 0, stream>>>(hp, ldh, rp_t, col_t, rows, H, C, slope,
                                                                                       reinterpret_cast<float2*>(stat));
  GatFwd a{hp, ldh, rowptr, col, reinterpret_cast<const float2*>(stat), rows, H, Fh, slope, row_graph, nmax, iso_idx, iso_w, iso_ptr, uscale, mean_heads,
           apply_elu, make_drop(drop_p, seed, drop_ctr), y, ldy};
  const unsigned grid = (unsigned)ceil_div64(rows, 4);
  const bool drop = a.drop.thresh != 0u;
  TSGNN_KNAME("gat_attn_fwd_kernel<%d,%s>", Fh / 4, drop ? "true" : "false");
#define GAT_FWD(L_) do { if (drop) gat_attn_fwd_kernel<L_, true><<<grid, 256, 0, stream>>>(a); \
                         else gat_attn_fwd_kernel<L_, false><<<grid, 256, 0, stream>>>(a); } while (0)
  switch (Fh / 4) {
    case 1: GAT_FWD(1); break;
    case 2: GAT_FWD(2); break;
    case 4: GAT_FWD(4); break;
    case 8: GAT_FWD(8); break;
    default: GAT_FWD(16); break;
  }
#undef GAT_FWD
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* number of row ranges per graph the backward sums dpre over (dupart holds B * P * H * Fh floats) */
int tsgnn_gat_bwd_parts(int B) {
  if (B <= 0) return 1;
  const int p = 256 / B;
  return p < 1 ? 1 : (p > 8 ? 8 : p);
}

/* backward of tsgnn_gat_attn_fwd_f32 up to d s_row and the listed edge-less columns' dh (tsgnn_gat_score_rowsum_f32 completes
 * dhp).  iso_row[rows * iso_row_ld]: the edge-less columns' weights per row (nullable with the list).  t1, t2: [nnz, H];
 * S: [rows, H]; dupart: [B, tsgnn_gat_bwd_parts(B), H * Fh] (nullable with the list). */
int tsgnn_gat_attn_bwd_f32(const float* hp, int64_t ldh, const float* y, int64_t ldy, const float* dy, int64_t lddy, const int* rp_t,
                           const int* col_t, int64_t rows, int H, int Fh, float slope, int mean_heads, int apply_elu,
                           const int* graph_ptr, int B, const int* iso_idx, const float* iso_w, const int* iso_ptr,
                           const float* iso_row, int iso_row_ld, float uscale, float drop_p, uint64_t seed,
                           const unsigned long long* drop_ctr, const float* stat, float* dhp, int Ns, float* t1, float* t2, float* S,
                           float* dupart, tsgnn_stream_t stream) {
  if (!dy) return TSGNN_EINVAL;
  return tsgnn_gat_attn_bwd_ro_f32(hp, ldh, y, ldy, dy, lddy, rp_t, col_t, rows, H, Fh, slope, mean_heads, apply_elu, graph_ptr, B, iso_idx,
                                   iso_w, iso_ptr, iso_row, iso_row_ld, uscale, drop_p, seed, drop_ctr, stat, dhp, Ns, t1, t2, S, dupart,
                                   nullptr, 0, nullptr, nullptr, stream);
}

/* the same for a layer whose output feeds ONLY the max readout over each graph's rows (the last layer, encoders_GAT.py:189): dy NULL,
 * and instead the readout's gradient ro_dout [B, Co] (leading dimension ro_ldo), its winners ro_arg [B, Co] and row_graph [rows]:
 * dy[i, c] = (ro_arg[b, c] == i) ? ro_dout[b, c] : 0 with b = row_graph[i] is formed on the fly (Co = the layer's output width). */
int tsgnn_gat_attn_bwd_ro_f32(const float* hp, int64_t ldh, const float* y, int64_t ldy, const float* dy, int64_t lddy, const int* rp_t,
                              const int* col_t, int64_t rows, int H, int Fh, float slope, int mean_heads, int apply_elu,
                              const int* graph_ptr, int B, const int* iso_idx, const float* iso_w, const int* iso_ptr,
                              const float* iso_row, int iso_row_ld, float uscale, float drop_p, uint64_t seed,
                              const unsigned long long* drop_ctr, const float* stat, float* dhp, int Ns, float* t1, float* t2, float* S,
                              float* dupart, const float* ro_dout, int64_t ro_ldo, const int* ro_arg, const int* row_graph,
                              tsgnn_stream_t stream) {
  if (!hp || !y || !rp_t || !col_t || !dhp || !stat || !t1 || !t2 || !S || rows < 0 || drop_p < 0.f || drop_p >= 1.f) return TSGNN_EINVAL;
  if ((dy == nullptr) == (ro_arg == nullptr)) return TSGNN_EINVAL;            // exactly one source of the output's gradient
  if (ro_arg && (!ro_dout || !row_graph)) return TSGNN_EINVAL;
  if (!fused_ok(H, Fh)) return TSGNN_EUNSUPPORTED;
  const int C = H * Fh, Co = mean_heads ? Fh : C;
  if (ro_arg) {
    if (ro_ldo < Co) return TSGNN_EINVAL;
    if ((ro_ldo % 4) || (Co % 4) || ((reinterpret_cast<uintptr_t>(ro_dout) | reinterpret_cast<uintptr_t>(ro_arg)) & 15)) return TSGNN_EUNSUPPORTED;
    dy = y; lddy = ldy;                                                        // (never read; keeps the checks below meaningful)
  }
  if (Ns < C + 2 * H || ldh < Ns || Ns - C - 2 * H > 64 || (ldh % 4) || (ldy % 4) || (lddy % 4) || ldy < Co || lddy < Co ||
      ((reinterpret_cast<uintptr_t>(hp) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dhp) |
        reinterpret_cast<uintptr_t>(dupart)) & 15))
    return TSGNN_EUNSUPPORTED;
  const bool lst = iso_idx != nullptr;
  if (lst && (!iso_w || !iso_ptr || !iso_row || !graph_ptr || !dupart || B <= 0 || iso_row_ld <= 0)) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  const int P = tsgnn_gat_bwd_parts(B);
  GatBwd a{hp, ldh, y, ldy, dy, lddy, rp_t, col_t, reinterpret_cast<const float2*>(stat), rows, H, Fh, slope, mean_heads, apply_elu, graph_ptr, B,
           lst ? iso_idx : nullptr, lst ? iso_w : nullptr, lst ? iso_ptr : nullptr, lst ? iso_row : nullptr, iso_row_ld, uscale,
           make_drop(drop_p, seed, drop_ctr), dhp, Ns, t1, t2, S, dupart, P, lst ? (unsigned)(B * P) : 0u,
           ro_arg ? ro_dout : nullptr, ro_ldo, ro_arg, row_graph};
  const unsigned grid = (unsigned)ceil_div64(rows, 4) + a.ngraph_blocks;
  const bool drop = a.drop.thresh != 0u;
  TSGNN_KNAME("gat_attn_bwd_kernel<%d,%s>", Fh / 4, drop ? "true" : "false");
#define GAT_BWD(L_) do { if (drop) gat_attn_bwd_kernel<L_, true><<<grid, 256, 0, stream>>>(a); \
                         else gat_attn_bwd_kernel<L_, false><<<grid, 256, 0, stream>>>(a); } while (0)
  switch (Fh / 4) {
    case 1: GAT_BWD(1); break;
    case 2: GAT_BWD(2); break;
    case 4: GAT_BWD(4); break;
    case 8: GAT_BWD(8); break;
    default: GAT_BWD(16); break;
  }
#undef GAT_BWD
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* d s_row into dhp[:, C + h]; with a list (iso_* and dupart of the backward, drop_p == 0): also dh of the listed columns */
int tsgnn_gat_score_rowsum_f32(const int* rowptr, const int* col, const int* eperm, const float* t1, const float* t2, const float* S,
                               int64_t rows, int H, float* dhp, int64_t ldh, int C, const float* dupart, int B, const int* iso_idx,
                               const float* iso_w, const int* iso_ptr, float uscale, tsgnn_stream_t stream) {
  if (!rowptr || !col || !eperm || !t1 || !t2 || !S || !dhp || rows < 0 || H <= 0 || C <= 0 || ldh < C + H) return TSGNN_EINVAL;
  if (dupart && (!iso_idx || !iso_w || !iso_ptr || B <= 0 || (C % 4) || C > 256)) return TSGNN_EINVAL;
  if (rows == 0) return TSGNN_OK;
  GatFin f{dupart, tsgnn_gat_bwd_parts(B), C, iso_idx, iso_w, iso_ptr, uscale, dupart ? (unsigned)B : 0u};
  gat_score_rowsum_kernel<<<(unsigned)ceil_div64(rows * H * 8, 256) + f.nfin, 256, 0, stream>>>(rowptr, col, eperm, t1, t2, S, rows, H, dhp,
                                                                                                 ldh, C, f);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gat_dropout_mult_f32(float drop_p, uint64_t seed, const unsigned long long* drop_ctr, int64_t i0, int64_t ni, int64_t j0,
                               int64_t nj, int H, float* out, tsgnn_stream_t stream) {
  if (!out || ni < 0 || nj < 0 || H <= 0 || drop_p < 0.f || drop_p >= 1.f) return TSGNN_EINVAL;
  if (ni * nj == 0) return TSGNN_OK;
  gat_dropout_mult_kernel<<<(unsigned)ceil_div64(ni * nj * H, 256), 256, 0, stream>>>(make_drop(drop_p, seed, drop_ctr), i0, ni, j0, nj, H, out);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* number of int64 words per layer of the pack descriptor, after the leading layer count */
int tsgnn_gat_pack_desc_words(void) { return 7 + 2 * GAT_HMAX; }

/* W' of up to 4 layers in one launch.  desc (HOST memory): [L, then per layer H, Fin, Fo, Ns, W' ptr, 0, 0, w_0..w_7, a_0..a_7] */
int tsgnn_gat_pack_f32(const int64_t* desc, tsgnn_stream_t stream) {
  GatPack p;
  if (!unpack_desc(desc, p, false)) return TSGNN_EINVAL;
  const GatPackLayer& last = p.l[p.L - 1];
  gat_pack_kernel<<<(unsigned)(last.blk0 + last.Fin), 256, 0, stream>>>(p);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* the heads' gradients from dW' of up to 4 layers in one launch.  desc as for the pack, with W' ptr = dW' and the gw [H, Fin, Fo],
 * ga [H, 2 Fo] outputs filled in */
int tsgnn_gat_unpack_f32(const int64_t* desc, tsgnn_stream_t stream) {
  GatPack p;
  if (!unpack_desc(desc, p, true)) return TSGNN_EINVAL;
  const GatPackLayer& last = p.l[p.L - 1];
  gat_unpack_kernel<<<(unsigned)(last.blk0 + last.Fin + last.H), 256, 0, stream>>>(p);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
