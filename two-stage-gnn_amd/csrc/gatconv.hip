// torch_geometric GATConv as fused launches: the PER-TARGET edge softmax (standard GAT; BASELINE.json north_star: "edge-softmax
// attention ... written for CDNA4"; SURVEY §8 a15 — no call site in the reference, PARITY UNPINNED).  Sibling of gat_fused.hip, which
// implements the reference's own DGATHead (softmax over the ROW index for every column, encoders_GAT.py:41).
//
// Operand layout (as gat_fused.hip): the layer's parameters are packed into ONE projection matrix with 2H extra columns
//     W' = [ W^T | W_h^T att_r_h ... | W_h^T att_l_h ... | 0 pad ]                                   [Fin, Ns],  C = H * Co
// so that hp = x W' carries per node its projected features AND both attention scalars of every head:
//     s_dst[i, h] = hp[i, C + h] = att_r_h . h_i   (the node as TARGET),   s_src[j, h] = hp[j, C + H + h] = att_l_h . h_j   (as SOURCE)
// (PyG: alpha_ij = softmax_j LeakyReLU(att_l . h_j + att_r . h_i) over the sources j of target i, self loops included).
// Their gradients are two more columns of dhp and return through the same two products as the features.
//
//   forward   gatconv_row_stats   (m, 1 / Z) of every (target row, head): 8 lanes per pair, two dependent round trips
//             gatconv_fwd         one wave per target row: alpha from the scalars + the row's statistics, the sources' feature rows
//                                 gathered 8 at a time (16 B per lane), mean over heads / + bias / ELU in registers
//   backward  gatconv_bwd_rows    one wave per target row: dpre_i (written out: the transposed pass gathers it), dalpha_ij =
//                                 <dpre_i, h_j>, S_i, d s_dst[i]; per-entry alpha and LeakyReLU terms for the transposed pass
//             (tsgnn_csr_spmm_heads_epi_f32 over A^T: dh_j = sum_i alpha_ij dpre_i ; tsgnn_gat_score_rowsum_f32 over A^T: d s_src[j])
//             gatconv_pack / _unpack   nn.Linear-layout weight + att_l / att_r <-> W', all layers in one launch each way
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

__device__ __forceinline__ float gc_lrelu(float t, float slope) { return t > 0.f ? t : slope * t; }
__device__ __forceinline__ float4 gc_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float gc_elu_grad_y(float y) { return y > 0.f ? 1.f : y + 1.f; }      // d/dx elu(x) = elu(x) + 1 for x <= 0

inline bool gatconv_ok(int H, int Co) {
  return (H == 1 || H == 2 || H == 4 || H == 8) && (Co == 4 || Co == 8 || Co == 16 || Co == 32 || Co == 64) && H * Co <= 256;
}

// ---------------------------------------------------------------- row statistics
__global__ __launch_bounds__(256) void gatconv_row_stats_kernel(const float* __restrict__ hp, int64_t ldh, const int* __restrict__ rowptr,
                                                                const int* __restrict__ col, int64_t rows, int H, int C, float slope,
                                                                float2* __restrict__ stat) {
  const int sub = threadIdx.x & 7;
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 8;
  if (q >= rows * H) return;
  const int64_t i = q / H;
  const int h = (int)(q % H);
  const int t0 = rowptr[i], t1 = rowptr[i + 1];
  if (t0 == t1) {
    if (sub == 0) stat[q] = make_float2(0.f, 0.f);
    return;
  }
  const float sdst = hp[i * ldh + C + h];
  const float* ssrc = hp + C + H + h;
  const int tf = t0 + sub;
  const bool has = tf < t1;
  const float ef = has ? gc_lrelu(ssrc[(int64_t)col[tf] * ldh] + sdst, slope) : -INFINITY;
  float m = ef;
  for (int t = tf + 8; t < t1; t += 8) m = fmaxf(m, gc_lrelu(ssrc[(int64_t)col[t] * ldh] + sdst, slope));
  m = group_max<8>(m);
  float z = has ? __expf(ef - m) : 0.f;
  for (int t = tf + 8; t < t1; t += 8) z += __expf(gc_lrelu(ssrc[(int64_t)col[t] * ldh] + sdst, slope) - m);
  z = group_sum<8>(z);
  if (sub == 0) stat[q] = make_float2(m, 1.f / z);
}

// ---------------------------------------------------------------- forward: one wave per target row
struct GcFwd {
  const float* hp; int64_t ldh;
  const int* rowptr; const int* col;
  const float2* stat;
  int64_t rows; int H, Co; float slope;
  int mean_heads, apply_elu;
  const float* bias;                          // nullable: [C] (concat) or [Co] (mean over heads)
  float* y; int64_t ldy;
};

template <int LPH>   // lanes per head = Co / 4
__global__ __launch_bounds__(256, 6) void gatconv_fwd_kernel(GcFwd a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t r = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wid;
  if (r >= a.rows) return;
  const int H = a.H, C = H * a.Co, G = H * LPH;
  const bool live = lane < G;
  const int h = live ? lane / LPH : 0;
  const int co = live ? 4 * lane : 0;
  const float* __restrict__ hp = a.hp;
  const int64_t ldh = a.ldh;
  const int e0 = a.rowptr[r], e1 = a.rowptr[r + 1];
  const int EB = min(8, 64 / H);              // entries per batch: lane p = (entry p / H, head p % H) computes one alpha
  const int pk = lane / H, ph = lane - pk * H;
  const float sdst_i = hp[r * ldh + C + ph];
  const float2 st = a.stat[r * H + ph];       // (m, 1 / Z) of this row
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bv = gc_ld4(a.bias + (a.mean_heads ? 4 * (lane % LPH) : co));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int eb = e0; eb < e1; eb += EB) {
    const int cnt = min(EB, e1 - eb);                       // uniform over the wave
    const bool has = pk < cnt;
    const int j = a.col[has ? eb + pk : eb];
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                           // unconditional requests (entries beyond the end repeat the first: alpha 0)
      const int jk = __shfl(j, (k < cnt ? k : 0) * H, 64);
      v[k] = gc_ld4(hp + (int64_t)jk * ldh + co);
    }
    const float ssrc_j = hp[(int64_t)j * ldh + C + H + ph];
    const float alpha = has ? __expf(gc_lrelu(sdst_i + ssrc_j, a.slope) - st.x) * st.y : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < cnt) {
        const float al = __shfl(alpha, k * H + h, 64);
        acc.x = fmaf(al, v[k].x, acc.x); acc.y = fmaf(al, v[k].y, acc.y);
        acc.z = fmaf(al, v[k].z, acc.z); acc.w = fmaf(al, v[k].w, acc.w);
      }
    }
  }
  if (a.mean_heads) {                                       // out.mean(dim = heads) + bias [, ELU]
    float4 s = acc;
    for (int k = 1; k < H; ++k) {
      const int src = lane + k * LPH < G ? lane + k * LPH : lane;
      s.x += __shfl(acc.x, src, 64); s.y += __shfl(acc.y, src, 64); s.z += __shfl(acc.z, src, 64); s.w += __shfl(acc.w, src, 64);
    }
    const float rH = 1.f / (float)H;                        // H is a power of two: exact
    s.x = fmaf(s.x, rH, bv.x); s.y = fmaf(s.y, rH, bv.y); s.z = fmaf(s.z, rH, bv.z); s.w = fmaf(s.w, rH, bv.w);
    if (a.apply_elu) {
      s.x = s.x <= 0.f ? expm1f(s.x) : s.x; s.y = s.y <= 0.f ? expm1f(s.y) : s.y;
      s.z = s.z <= 0.f ? expm1f(s.z) : s.z; s.w = s.w <= 0.f ? expm1f(s.w) : s.w;
    }
    if (lane < LPH) *reinterpret_cast<float4*>(a.y + r * a.ldy + 4 * lane) = s;
  } else {
    acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
    if (a.apply_elu) {
      acc.x = acc.x <= 0.f ? expm1f(acc.x) : acc.x; acc.y = acc.y <= 0.f ? expm1f(acc.y) : acc.y;
      acc.z = acc.z <= 0.f ? expm1f(acc.z) : acc.z; acc.w = acc.w <= 0.f ? expm1f(acc.w) : acc.w;
    }
    if (live) *reinterpret_cast<float4*>(a.y + r * a.ldy + co) = acc;
  }
}

// ---------------------------------------------------------------- backward, target side: one wave per row
struct GcBwd {
  const float* hp; int64_t ldh;
  const float* y; int64_t ldy;                // the layer's output (ELU' from it)
  const float* dy; int64_t lddy;
  const int* rowptr; const int* col;
  const float2* stat;
  int64_t rows; int H, Co; float slope;
  int mean_heads, apply_elu;
  float* dpre; int64_t lddp;                  // [rows, C] out: the gradient of the pre-activation aggregate (per head)
  float* dhp; int Ns;                         // [rows, ldh]: this kernel writes column C + h (d s_dst) and the zero pad
  float* alpha; float* t1; float* t2;         // [nnz, H] in A's entry order: alpha, lrelu' alpha dalpha, lrelu' alpha
  float* S;                                   // [rows, H]: sum_j alpha_ij dalpha_ij
};

template <int LPH>
__global__ __launch_bounds__(256, 5) void gatconv_bwd_rows_kernel(GcBwd a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t r = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wid;
  if (r >= a.rows) return;
  const int H = a.H, C = H * a.Co, G = H * LPH;
  const bool live = lane < G;
  const int h = live ? lane / LPH : 0;
  const int co = live ? 4 * lane : 0;
  const float* __restrict__ hp = a.hp;
  const int64_t ldh = a.ldh;
  // dpre of this row for the lane's four features: dy * ELU'(y) (concat) or (dy * ELU'(y)) / H of the lane's slot (mean over heads)
  const int cy = a.mean_heads ? 4 * (lane % LPH) : co;
  float4 dp = gc_ld4(a.dy + r * a.lddy + cy);
  const float4 yv = gc_ld4(a.y + r * a.ldy + cy);
  const int e0 = a.rowptr[r], e1 = a.rowptr[r + 1];
  const int EB = min(8, 64 / H);
  const int pk = lane / H, ph = lane - pk * H;
  const float sdst_i = hp[r * ldh + C + ph];
  const float2 st = a.stat[r * H + ph];
  if (a.apply_elu) { dp.x *= gc_elu_grad_y(yv.x); dp.y *= gc_elu_grad_y(yv.y); dp.z *= gc_elu_grad_y(yv.z); dp.w *= gc_elu_grad_y(yv.w); }
  if (!live) dp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.mean_heads) {
    const float rH = 1.f / (float)H;
    dp.x *= rH; dp.y *= rH; dp.z *= rH; dp.w *= rH;
  }
  if (live) *reinterpret_cast<float4*>(a.dpre + r * a.lddp + co) = dp;
  float S = 0.f, P1 = 0.f, P2 = 0.f;
  for (int eb = e0; eb < e1; eb += EB) {
    const int cnt = min(EB, e1 - eb);
    const bool has = pk < cnt;
    const int j = a.col[has ? eb + pk : eb];
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int jk = __shfl(j, (k < cnt ? k : 0) * H, 64);
      v[k] = gc_ld4(hp + (int64_t)jk * ldh + co);
    }
    const float ssrc_j = hp[(int64_t)j * ldh + C + H + ph];
    const float tt = sdst_i + ssrc_j;
    const float alpha = has ? __expf(gc_lrelu(tt, a.slope) - st.x) * st.y : 0.f;
    float dal = 0.f;                                        // <dpre_i, h_j> of THIS lane's (entry pk, head ph)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < cnt) {
        const float d = group_sum<LPH>((dp.x * v[k].x + dp.y * v[k].y) + (dp.z * v[k].z + dp.w * v[k].w));   // every lane of head h: dal(k, h)
        const float dsel = __shfl(d, ph * LPH, 64);
        if (pk == k) dal = dsel;
      }
    }
    if (has) {
      const float lr = tt > 0.f ? 1.f : a.slope;
      const float w1 = lr * alpha * dal, w2 = lr * alpha;
      S = fmaf(alpha, dal, S); P1 += w1; P2 += w2;
      const int64_t e = (int64_t)(eb + pk) * H + ph;
      a.alpha[e] = alpha; a.t1[e] = w1; a.t2[e] = w2;
    }
  }
  // sums over the entry slots pk of the same head (lanes pk * H + ph): xor strides H, 2H, ...
  for (int s = H; s < EB * H; s <<= 1) {
    S += __shfl_xor(S, s, 64); P1 += __shfl_xor(P1, s, 64); P2 += __shfl_xor(P2, s, 64);
  }
  if (lane < H) {
    a.S[r * H + lane] = S;
    a.dhp[r * ldh + C + lane] = P1 - S * P2;                // d s_dst[i] = sum_j lrelu' alpha (dalpha - S)
  }
  if (lane < a.Ns - C - 2 * H) a.dhp[r * ldh + C + 2 * H + lane] = 0.f;
}

// ---------------------------------------------------------------- parameters <-> W'
constexpr int GC_LMAX = 4;
struct GcLayer {
  int H, Fin, Co, Ns;
  const float* w; int64_t ldw;                // lin_l.weight [H * Co, Fin] (nn.Linear's [out, in])
  const float* att_r; const float* att_l;     // [H * Co]: dotted with the TARGET / SOURCE node
  float* wp;                                  // pack: W' [Fin, Ns] out; unpack: dW' in
  float* gw; float* gar; float* gal;          // unpack: gradients out
  int blk0;
};
struct GcPack { GcLayer l[GC_LMAX]; int L; };

// one block per (layer, input row k): W'[k, c] = W[c, k]; W'[k, C + h] = sum_f W[h Co + f, k] att_r[h Co + f]; ... att_l
__global__ __launch_bounds__(256) void gatconv_pack_kernel(GcPack p) {
  __shared__ float red[2][256];
  int li = 0;
#pragma unroll
  for (int t = 1; t < GC_LMAX; ++t) if (t < p.L && (int)blockIdx.x >= p.l[t].blk0) li = t;
  const GcLayer& L = p.l[li];
  const int k = (int)blockIdx.x - L.blk0, tid = threadIdx.x;
  const int C = L.H * L.Co;
  float pr = 0.f, pl = 0.f;
  if (tid < C) {
    const float w = L.w[(int64_t)tid * L.ldw + k];
    L.wp[(int64_t)k * L.Ns + tid] = w;
    pr = w * L.att_r[tid]; pl = w * L.att_l[tid];
  }
  red[0][tid] = pr; red[1][tid] = pl;
  __syncthreads();
  if (tid < 2 * L.H) {                                      // 2H dots of Co terms each, in index order
    const int which = tid / L.H, h = tid % L.H;
    float s = 0.f;
    for (int f = 0; f < L.Co; ++f) s += red[which][h * L.Co + f];
    L.wp[(int64_t)k * L.Ns + C + which * L.H + h] = s;
  }
  if (tid < L.Ns - C - 2 * L.H) L.wp[(int64_t)k * L.Ns + C + 2 * L.H + tid] = 0.f;
}

// gradients.  Blocks [blk0, blk0 + Fin): gw[c, k] = dW'[k, c] + dW'[k, C + h] att_r[c] + dW'[k, C + H + h] att_l[c]  (thread c, h = c / Co);
// blocks [blk0 + Fin, blk0 + Fin + ceil(C / 4)): gar[c] = sum_k W[c, k] dW'[k, C + h], gal[c] = sum_k W[c, k] dW'[k, C + H + h] — one wave
// per c, lanes stride over k (W's row is contiguous in k), lane sums added by DPP
__global__ __launch_bounds__(256) void gatconv_unpack_kernel(GcPack p) {
  int li = 0;
#pragma unroll
  for (int t = 1; t < GC_LMAX; ++t) if (t < p.L && (int)blockIdx.x >= p.l[t].blk0) li = t;
  const GcLayer& L = p.l[li];
  const int b = (int)blockIdx.x - L.blk0;
  const int C = L.H * L.Co;
  if (b < L.Fin) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const int h = c / L.Co;
    const float* d = L.wp + (int64_t)b * L.Ns;
    L.gw[(int64_t)c * L.ldw + b] = d[c] + d[C + h] * L.att_r[c] + d[C + L.H + h] * L.att_l[c];
    return;
  }
  const int c = 4 * (b - L.Fin) + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const int h = c / L.Co;
  float sr = 0.f, sl = 0.f;
  for (int k = lane; k < L.Fin; k += 64) {
    const float w = L.w[(int64_t)c * L.ldw + k];
    sr = fmaf(w, L.wp[(int64_t)k * L.Ns + C + h], sr);
    sl = fmaf(w, L.wp[(int64_t)k * L.Ns + C + L.H + h], sl);
  }
  sr = wave_sum(sr); sl = wave_sum(sl);
  if (lane == 0) { L.gar[c] = sr; L.gal[c] = sl; }
}

int fill_pack(const int64_t* desc, GcPack& p, bool unpack) {
  if (!desc) return TSGNN_EINVAL;
  p.L = (int)desc[0];
  if (p.L <= 0 || p.L > GC_LMAX) return TSGNN_EINVAL;
  int blk = 0;
  const int64_t* d = desc + 1;
  for (int t = 0; t < p.L; ++t, d += 12) {
    GcLayer& l = p.l[t];
    l.H = (int)d[0]; l.Fin = (int)d[1]; l.Co = (int)d[2]; l.Ns = (int)d[3];
    l.w = reinterpret_cast<const float*>(d[4]); l.ldw = d[5];
    l.att_r = reinterpret_cast<const float*>(d[6]); l.att_l = reinterpret_cast<const float*>(d[7]);
    l.wp = reinterpret_cast<float*>(d[8]);
    l.gw = reinterpret_cast<float*>(d[9]); l.gar = reinterpret_cast<float*>(d[10]); l.gal = reinterpret_cast<float*>(d[11]);
    if (!l.w || !l.att_r || !l.att_l || !l.wp || l.Fin <= 0 || l.ldw < l.Fin || (unpack && (!l.gw || !l.gar || !l.gal))) return TSGNN_EINVAL;
    if (!gatconv_ok(l.H, l.Co) || l.Ns < l.H * l.Co + 2 * l.H || l.Ns - l.H * l.Co - 2 * l.H > 256) return TSGNN_EUNSUPPORTED;
    l.blk0 = blk;
    blk += l.Fin + (unpack ? (l.H * l.Co + 3) / 4 : 0);
  }
  return blk;
}

}  // namespace

extern "C" {

int tsgnn_gatconv_supported(int H, int Co) { return gatconv_ok(H, Co) ? 1 : 0; }

/* words per layer in the host description of tsgnn_gatconv_pack_f32 / _unpack_f32:
 * [L <= 4, L x (H, Fin, Co, Ns, w, ldw, att_r, att_l, wp, gw, gar, gal)] */
int tsgnn_gatconv_pack_desc_words(void) { return 12; }

int tsgnn_gatconv_pack_f32(const int64_t* desc, tsgnn_stream_t stream) {
  GcPack p{};
  const int blk = fill_pack(desc, p, false);
  if (blk < 0) return blk;
  TSGNN_KNAME("gatconv_pack_kernel");
  gatconv_pack_kernel<<<(unsigned)blk, 256, 0, stream>>>(p);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gatconv_unpack_f32(const int64_t* desc, tsgnn_stream_t stream) {
  GcPack p{};
  const int blk = fill_pack(desc, p, true);
  if (blk < 0) return blk;
  TSGNN_KNAME("gatconv_unpack_kernel");
  gatconv_unpack_kernel<<<(unsigned)blk, 256, 0, stream>>>(p);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gatconv_fwd_f32(const float* hp, int64_t ldh, const int* rowptr, const int* col, int64_t rows, int H, int Co, float slope,
                          int mean_heads, int apply_elu, const float* bias, float* stat, float* y, int64_t ldy, tsgnn_stream_t stream) {
  if (!hp || !rowptr || !col || !y || !stat || (reinterpret_cast<uintptr_t>(stat) & 7) || rows < 0) return TSGNN_EINVAL;
  if (!gatconv_ok(H, Co)) return TSGNN_EUNSUPPORTED;
  const int C = H * Co;
  if (ldh < C + 2 * H || (ldh % 4) || (ldy % 4) || ldy < (mean_heads ? Co : C) ||
      ((reinterpret_cast<uintptr_t>(hp) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias)) & 15))
    return TSGNN_EUNSUPPORTED;
  if (rows == 0) return TSGNN_OK;
  gatconv_row_stats_kernel<<<(unsigned)ceil_div64(rows * H * 8, 256), 256, 0, stream>>>(hp, ldh, rowptr, col, rows, H, C, slope,
                                                                                           reinterpret_cast<float2*>(stat));
  GcFwd a{hp, ldh, rowptr, col, reinterpret_cast<const float2*>(stat), rows, H, Co, slope, mean_heads, apply_elu, bias, y, ldy};
  const unsigned grid = (unsigned)ceil_div64(rows, 4);
  TSGNN_KNAME("gatconv_fwd_kernel<%d>", Co / 4);
  switch (Co / 4) {
    case 1: gatconv_fwd_kernel<1><<<grid, 256, 0, stream>>>(a); break;
    case 2: gatconv_fwd_kernel<2><<<grid, 256, 0, stream>>>(a); break;
    case 4: gatconv_fwd_kernel<4><<<grid, 256, 0, stream>>>(a); break;
    case 8: gatconv_fwd_kernel<8><<<grid, 256, 0, stream>>>(a); break;
    default: gatconv_fwd_kernel<16><<<grid, 256, 0, stream>>>(a); break;
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_gatconv_bwd_rows_f32(const float* hp, int64_t ldh, const float* y, int64_t ldy, const float* dy, int64_t lddy, const int* rowptr,
                               const int* col, int64_t rows, int H, int Co, float slope, int mean_heads, int apply_elu, const float* stat,
                               float* dpre, int64_t lddp, float* dhp, int Ns, float* alpha, float* t1, float* t2, float* S,
                               tsgnn_stream_t stream) {
  if (!hp || !y || !dy || !rowptr || !col || !stat || !dpre || !dhp || !alpha || !t1 || !t2 || !S || rows < 0) return TSGNN_EINVAL;
  if (!gatconv_ok(H, Co)) return TSGNN_EUNSUPPORTED;
  const int C = H * Co, Cy = mean_heads ? Co : C;
  if (ldh < C + 2 * H || Ns > ldh || Ns < C + 2 * H || Ns - C - 2 * H > 64 || (ldh % 4) || (ldy % 4) || (lddy % 4) || (lddp % 4) || ldy < Cy ||
      lddy < Cy || lddp < C ||
      ((reinterpret_cast<uintptr_t>(hp) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dpre)) & 15))
    return TSGNN_EUNSUPPORTED;
  if (rows == 0) return TSGNN_OK;
  GcBwd a{hp, ldh, y, ldy, dy, lddy, rowptr, col, reinterpret_cast<const float2*>(stat), rows, H, Co, slope, mean_heads, apply_elu, dpre, lddp,
          dhp, Ns, alpha, t1, t2, S};
  const unsigned grid = (unsigned)ceil_div64(rows, 4);
  TSGNN_KNAME("gatconv_bwd_rows_kernel<%d>", Co / 4);
  switch (Co / 4) {
    case 1: gatconv_bwd_rows_kernel<1><<<grid, 256, 0, stream>>>(a); break;
    case 2: gatconv_bwd_rows_kernel<2><<<grid, 256, 0, stream>>>(a); break;
    case 4: gatconv_bwd_rows_kernel<4><<<grid, 256, 0, stream>>>(a); break;
    case 8: gatconv_bwd_rows_kernel<8><<<grid, 256, 0, stream>>>(a); break;
    default: gatconv_bwd_rows_kernel<16><<<grid, 256, 0, stream>>>(a); break;
  }
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
