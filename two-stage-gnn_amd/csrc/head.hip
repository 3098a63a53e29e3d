// Graph-level prediction head of the encoders (encoders.py:207-217 / :396-406): two chained nn.Linear on the
// concatenated readout [B, P]  ->  vec [B, E]  ->  y [B, C],  forward and backward in 1 + 1 launches instead of the
// ~8 library launches (2 addmm, 4 mm, 2 bias reductions) a B = 32 batch spends most of its time dispatching.
// Shapes are tiny (B <= a few hundred rows): one workgroup per graph row, weights streamed from L2 with 16-byte loads.
#include "common.h"
#include "../../include/tsgnn.h"
#include "readout_body.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// block b: vec[b,:] = W1 out[b,:] + b1 ; y[b,:] = W2 vec[b,:] + b2        (W1 [E,P], W2 [C,E] row-major = nn.Linear.weight)
constexpr int HW = 16;   // waves per row block (1024 threads): one batch of 8 weight rows per wave covers E = 128

// shared tail: xs[P] (the readout row of graph b) is in LDS; vec[b,:] = W1 xs + b1 ; y[b,:] = W2 vec[b,:] + b2
__device__ __forceinline__ void head2_fwd_tail(const float* xs, float* vs, int b, const float* __restrict__ w1,
                                               const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                               int P, int E, int C, float* __restrict__ vec, float* __restrict__ y) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int P4 = P >> 2;
  // each wave owns rows j = wid, wid+16, ...; eight rows are in flight per iteration (independent 16-byte loads),
  // their dot products are reduced together
  for (int j0 = wid; j0 < E; j0 += 8 * HW) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int k4 = lane; k4 < P4; k4 += 64) {
      const float4 x = *reinterpret_cast<const float4*>(xs + 4 * k4);
      float4 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + HW * u;
        w[u] = ld4(w1 + (int64_t)(j < E ? j : 0) * P + 4 * k4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += (w[u].x * x.x + w[u].y * x.y) + (w[u].z * x.z + w[u].w * x.w);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + HW * u;
      const float r = wave_sum(acc[u]);
      if (lane == 0 && j < E) { const float v = r + (b1 ? b1[j] : 0.f); vs[j] = v; vec[(int64_t)b * E + j] = v; }
    }
  }
  __syncthreads();
  for (int c = wid; c < C; c += HW) {
    float acc = 0.f;
    for (int j = lane; j < E; j += 64) acc = fmaf(w2[(int64_t)c * E + j], vs[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) y[(int64_t)b * C + c] = acc + (b2 ? b2[c] : 0.f);
  }
}

__global__ __launch_bounds__(64 * HW) void head2_fwd_kernel(const float* __restrict__ out, int64_t ldo, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, int P, int E, int C,
                                                        float* __restrict__ vec, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                 // [P]
  float* vs = smem + ((P + 3) & ~3);  // [E]
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < P; k += 64 * HW) xs[k] = out[(int64_t)b * ldo + k];
  __syncthreads();
  head2_fwd_tail(xs, vs, b, w1, b1, w2, b2, P, E, C, vec, y);
}

// The GraphSage stack's tail in ONE launch: block b finishes graph b's concatenated max readout and runs the head on it.
//   layers 0 .. L-2 : decode the packed (value, ~row) maxima the per-layer partial kernels accumulated;
//   last layer      : scan graph b's rows of v_last directly (32 row lanes x 32 float4 lanes; first ghost row = all ghost
//                     rows of the un-normalised last layer) - no partial launch, no atomics for that layer;
//   then vec = W1 out + b1, y = W2 vec + b2.   out[b, :] and arg (winning rows, packed-layout order) are kept for backward.
struct ReadoutHeadArgs {
  const unsigned long long* packed; int B, L, Fh, Fl;
  const float* v_last; int64_t ldv;
  const int* graph_ptr; int64_t n_real; int nslots; int n_ghost;
  float* out; int64_t ldo; int* arg;
};
__global__ __launch_bounds__(64 * HW) void readout_head_fwd_kernel(ReadoutHeadArgs a, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, int P, int E, int C,
                                                               float* __restrict__ vec, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                                   // [P]
  float* vs = smem + ((P + 3) & ~3);                                  // [E]
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem + ((P + 3) & ~3) + ((E + 3) & ~3));   // [32][Fl]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int PH = (a.L - 1) * a.Fh;
  for (int k = tid; k < PH; k += 64 * HW) {
    const int l = k / a.Fh, f = k % a.Fh;
    const int64_t i = (int64_t)l * a.B * a.Fh + (int64_t)b * a.Fh + f;
    const unsigned long long p = a.packed[i];
    const float val = p ? ordered_f32((unsigned)(p >> 32)) : 0.f;
    xs[k] = val;
    a.out[(int64_t)b * a.ldo + k] = val;
    a.arg[i] = p ? (int)(0xFFFFFFFFu - (unsigned)(p & 0xFFFFFFFFull)) : -1;
  }
  {
    const int c4 = tid & 31, rl = tid >> 5, F4 = a.Fl >> 2;
    const int g0 = a.graph_ptr[b], sz = a.graph_ptr[b + 1] - g0;
    unsigned long long m0 = 0ull, m1 = 0ull, m2 = 0ull, m3 = 0ull;
    if (c4 < F4) {
      const int nrows = sz + ((a.n_ghost && sz < a.nslots) ? 1 : 0);   // + the first ghost row (the others equal it)
      // four rows in flight per lane (independent 16-byte loads), maxima folded afterwards
      for (int n0 = rl; n0 < nrows; n0 += 8 * HW) {
        float4 t[4];
        int64_t r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = n0 + 2 * HW * u;
          r[u] = n < sz ? (int64_t)g0 + n : a.n_real + n;
          t[u] = n < nrows ? ld4(a.v_last + r[u] * a.ldv + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (n0 + 2 * HW * u < nrows) {
            const unsigned long long p0 = pack_max(t[u].x, (unsigned)r[u]), p1 = pack_max(t[u].y, (unsigned)r[u]),
                                     p2 = pack_max(t[u].z, (unsigned)r[u]), p3 = pack_max(t[u].w, (unsigned)r[u]);
            m0 = p0 > m0 ? p0 : m0; m1 = p1 > m1 ? p1 : m1; m2 = p2 > m2 ? p2 : m2; m3 = p3 > m3 ? p3 : m3;
          }
        }
      }
      best[rl * a.Fl + 4 * c4 + 0] = m0; best[rl * a.Fl + 4 * c4 + 1] = m1; best[rl * a.Fl + 4 * c4 + 2] = m2; best[rl * a.Fl + 4 * c4 + 3] = m3;
    }
    __syncthreads();
    for (int f = tid; f < a.Fl; f += 64 * HW) {
      unsigned long long m = best[f];
      for (int w = 1; w < 2 * HW; ++w) { const unsigned long long o = best[w * a.Fl + f]; m = o > m ? o : m; }
      const float val = m ? ordered_f32((unsigned)(m >> 32)) : 0.f;
      xs[PH + f] = val;
      a.out[(int64_t)b * a.ldo + PH + f] = val;
      a.arg[(int64_t)(a.L - 1) * a.B * a.Fh + (int64_t)b * a.Fl + f] = m ? (int)(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull)) : -1;
    }
  }
  __syncthreads();
  head2_fwd_tail(xs, vs, b, w1, b1, w2, b2, P, E, C, vec, y);
}

// Backward in ONE launch.  Blocks [0, B): block b computes dvt[b,:] = dvec[b,:] + W2^T dy[b,:] and dout[b,:] = W1^T dvt[b,:].
// Blocks [B, B + ceil(E/4)]: the weight gradients; they rebuild the four dvt columns they need from dy, dvec and W2
// (C fused multiply-adds per value) instead of waiting for the row blocks, so nothing orders the two groups.
__device__ __forceinline__ void head2_bwd_rows(float* smem, int b, const float* __restrict__ dy, const float* __restrict__ dvec,
                                               const float* __restrict__ w1, const float* __restrict__ w2, int P, int E, int C,
                                               float* __restrict__ dout, int64_t ldo) {
  float* ds = smem;                          // [E] dvt row
  float* part = smem + ((E + 3) & ~3);       // [HW][P] partial dout
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  for (int j = tid; j < E; j += 64 * HW) {
    float acc = dvec ? dvec[(int64_t)b * E + j] : 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(dy[(int64_t)b * C + c], w2[(int64_t)c * E + j], acc);
    ds[j] = acc;
  }
  __syncthreads();
  // each wave takes every 4th row j of W1 and accumulates its contribution to all P columns
  const int P4 = P >> 2;
  for (int k4 = lane; k4 < P4 + ((P & 3) ? 1 : 0); k4 += 64) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k4 < P4) {
      for (int j = wid; j < E; j += HW) {
        const float4 w = ld4(w1 + (int64_t)j * P + 4 * k4);
        const float d = ds[j];
        acc.x = fmaf(d, w.x, acc.x); acc.y = fmaf(d, w.y, acc.y); acc.z = fmaf(d, w.z, acc.z); acc.w = fmaf(d, w.w, acc.w);
      }
      *reinterpret_cast<float4*>(part + wid * ((P + 3) & ~3) + 4 * k4) = acc;
    } else {
      for (int k = 4 * P4; k < P; ++k) {
        float a = 0.f;
        for (int j = wid; j < E; j += HW) a = fmaf(ds[j], w1[(int64_t)j * P + k], a);
        part[wid * ((P + 3) & ~3) + k] = a;
      }
    }
  }
  __syncthreads();
  const int PP = (P + 3) & ~3;
  for (int k = tid; k < P; k += 64 * HW) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < HW; ++w) a += part[w * PP + k];
    dout[(int64_t)b * ldo + k] = a;
  }
}

// weights: block jb owns rows j = 4*jb .. 4*jb+3 of dW1 (+ db1); the last block produces dW2, db2
__device__ __forceinline__ void head2_bwd_weights(float* smem /* [B][4] dvt slice */, int jb, const float* __restrict__ out, int64_t ldo,
                                                  const float* __restrict__ vec, const float* __restrict__ dy,
                                                  const float* __restrict__ dvec, const float* __restrict__ w2, int B, int P, int E,
                                                  int C, float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                  float* __restrict__ db2, float* __restrict__ normparts) {
  const int tid = threadIdx.x, NTH = 64 * HW;
  const int nj = (E + 3) / 4;
  float sq = 0.f;                                        // |gradient written by this thread|^2
  if (jb < nj) {
    const int j0 = 4 * jb;
    for (int i = tid; i < B * 4; i += NTH) {
      const int bb = i >> 2, jj = j0 + (i & 3);
      float a = 0.f;
      if (jj < E) {
        a = dvec ? dvec[(int64_t)bb * E + jj] : 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(dy[(int64_t)bb * C + c], w2[(int64_t)c * E + jj], a);   // same order as the row blocks
      }
      smem[i] = a;
    }
    __syncthreads();
    for (int k = tid; k < P; k += NTH) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int bb = 0;
      for (; bb + 8 <= B; bb += 8) {                     // eight rows in flight (independent loads), same summation order
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = out[(int64_t)(bb + u) * ldo + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 d = *reinterpret_cast<const float4*>(smem + 4 * (bb + u));
          a0 = fmaf(d.x, x[u], a0); a1 = fmaf(d.y, x[u], a1); a2 = fmaf(d.z, x[u], a2); a3 = fmaf(d.w, x[u], a3);
        }
      }
      for (; bb < B; ++bb) {
        const float x = out[(int64_t)bb * ldo + k];
        a0 = fmaf(smem[4 * bb + 0], x, a0); a1 = fmaf(smem[4 * bb + 1], x, a1);
        a2 = fmaf(smem[4 * bb + 2], x, a2); a3 = fmaf(smem[4 * bb + 3], x, a3);
      }
      if (j0 + 0 < E) { dw1[(int64_t)(j0 + 0) * P + k] = a0; sq = fmaf(a0, a0, sq); }
      if (j0 + 1 < E) { dw1[(int64_t)(j0 + 1) * P + k] = a1; sq = fmaf(a1, a1, sq); }
      if (j0 + 2 < E) { dw1[(int64_t)(j0 + 2) * P + k] = a2; sq = fmaf(a2, a2, sq); }
      if (j0 + 3 < E) { dw1[(int64_t)(j0 + 3) * P + k] = a3; sq = fmaf(a3, a3, sq); }
    }
    if (db1 && tid < 4 && j0 + tid < E) {
      float a = 0.f;
      for (int bb = 0; bb < B; ++bb) a += smem[4 * bb + tid];
      db1[j0 + tid] = a;
      sq = fmaf(a, a, sq);
    }
  } else {
    for (int i = tid; i < C * E; i += NTH) {
      const int c = i / E, j = i % E;
      float a = 0.f;
      for (int bb = 0; bb < B; ++bb) a = fmaf(dy[(int64_t)bb * C + c], vec[(int64_t)bb * E + j], a);
      dw2[i] = a;
      sq = fmaf(a, a, sq);
    }
    if (db2) for (int c = tid; c < C; c += NTH) {
      float a = 0.f;
      for (int bb = 0; bb < B; ++bb) a += dy[(int64_t)bb * C + c];
      db2[c] = a;
      sq = fmaf(a, a, sq);
    }
  }
  if (normparts) {                                       // block total in a fixed order: waves through LDS
    sq = wave_sum(sq);
    __syncthreads();
    if ((tid & 63) == 0) smem[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < HW; ++w) t += smem[w];
      normparts[jb] = t;
    }
  }
}

// ce_label != NULL: dy is not an input — it is the gradient of mean softmax cross-entropy of the logits ce_y[B, C] w.r.t. them
// (F.cross_entropy, encoders.py:221-224), rebuilt by every block in LDS (B * C values) instead of being produced by a launch
// of its own between the forward and this kernel; block 0 also writes the loss value.
__global__ __launch_bounds__(64 * HW) void head2_bwd_kernel(const float* __restrict__ out, int64_t ldo, const float* __restrict__ vec,
                                                        const float* __restrict__ dy, const float* __restrict__ dvec,
                                                        const float* __restrict__ w1, const float* __restrict__ w2, int B, int P, int E,
                                                        int C, float* __restrict__ dout, int64_t lddo, float* __restrict__ dw1,
                                                        float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                        float* __restrict__ normparts, const float* __restrict__ ce_y,
                                                        const int64_t* __restrict__ ce_label, float* __restrict__ ce_loss) {
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  float* smem = smem_all;
  if (ce_label != nullptr) {
    float* dyl = smem_all;                               // [B * C]
    float* lb = smem_all + ((B * C + 3) & ~3);           // [B] per-graph loss terms
    smem = lb + ((B + 3) & ~3);
    const float invB = 1.f / (float)B;
    // weight blocks (and block 0, which also writes the loss) need every row's gradient, a row block only its own
    const bool all_rows = (int)blockIdx.x >= B || blockIdx.x == 0;
    for (int b = all_rows ? (int)threadIdx.x : (int)blockIdx.x + (int)threadIdx.x * B; b < B; b += 64 * HW) {
      const float* row = ce_y + (int64_t)b * C;
      float m = -INFINITY;
      for (int c = 0; c < C; ++c) m = fmaxf(m, row[c]);
      float d = 0.f;
      for (int c = 0; c < C; ++c) d += expf(row[c] - m);
      const int yb = (int)ce_label[b];
      const float logz = m + logf(d);
      lb[b] = logz - row[yb];
      for (int c = 0; c < C; ++c) dyl[b * C + c] = (expf(row[c] - logz) - (c == yb ? 1.f : 0.f)) * invB;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      float t = 0.f;
      for (int b = 0; b < B; ++b) t += lb[b];            // graph order: reproducible
      ce_loss[0] = t * invB;
    }
    dy = dyl;
  }
  if ((int)blockIdx.x < B) head2_bwd_rows(smem, blockIdx.x, dy, dvec, w1, w2, P, E, C, dout, lddo);
  else head2_bwd_weights(smem, (int)blockIdx.x - B, out, ldo, vec, dy, dvec, w2, B, P, E, C, dw1, db1, dw2, db2, normparts);
}

}  // namespace

extern "C" {

int tsgnn_head2_fwd_f32(const float* out, int64_t ldo, const float* w1, const float* b1, const float* w2, const float* b2, int B, int P,
                        int E, int C, float* vec, float* y, tsgnn_stream_t stream) {
  if (!out || !w1 || !w2 || !vec || !y || B <= 0 || P <= 0 || E <= 0 || C <= 0 || ldo < P) return TSGNN_EINVAL;
  if ((P % 4) || P > 4096 || E > 4096 || (reinterpret_cast<uintptr_t>(w1) & 15)) return TSGNN_EUNSUPPORTED;
  const size_t lds = sizeof(float) * (size_t)(((P + 3) & ~3) + E);
  TSGNN_KNAME("head2_fwd_kernel");
  head2_fwd_kernel<<<B, 64 * HW, lds, stream>>>(out, ldo, w1, b1, w2, b2, P, E, C, vec, y);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_readout_head_fwd_f32(const unsigned long long* packed, int B, int L, int Fh, int Fl, const float* v_last, int64_t ldv,
                               const int* graph_ptr, int64_t n_real, int nslots, int n_ghost, float* out, int64_t ldo, int* arg,
                               const float* w1, const float* b1, const float* w2, const float* b2, int E, int C, float* vec, float* y,
                               tsgnn_stream_t stream) {
  if (!packed || !v_last || !graph_ptr || !out || !arg || !w1 || !w2 || !vec || !y || B <= 0 || L <= 0 || Fh <= 0 || Fl <= 0 ||
      E <= 0 || C <= 0 || nslots <= 0 || (n_ghost != 0 && n_ghost != nslots))
    return TSGNN_EINVAL;
  const int P = (L - 1) * Fh + Fl;
  if ((P % 4) || P > 4096 || E > 4096 || (Fl % 4) || Fl > 128 || (ldv % 4) || ldo < P || (reinterpret_cast<uintptr_t>(w1) & 15) ||
      (reinterpret_cast<uintptr_t>(v_last) & 15))
    return TSGNN_EUNSUPPORTED;
  ReadoutHeadArgs a{packed, B, L, Fh, Fl, v_last, ldv, graph_ptr, n_real, nslots, n_ghost, out, ldo, arg};
  const size_t lds = sizeof(float) * (size_t)(((P + 3) & ~3) + ((E + 3) & ~3)) + sizeof(unsigned long long) * (size_t)(2 * HW) * Fl;
  TSGNN_KNAME("readout_head_fwd_kernel");
  readout_head_fwd_kernel<<<B, 64 * HW, lds, stream>>>(a, w1, b1, w2, b2, P, E, C, vec, y);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

static int head2_bwd_launch(const float* out, int64_t ldo, const float* vec, const float* dy, const float* dvec, const float* w1,
                            const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo, float* dw1, float* db1,
                            float* dw2, float* db2, float* normparts, const float* ce_y, const int64_t* ce_label, float* ce_loss,
                            tsgnn_stream_t stream) {
  if (!out || !vec || (!dy && !ce_label) || !w1 || !w2 || !dout || !dw1 || !dw2 || B <= 0 || P <= 0 || E <= 0 || C <= 0) return TSGNN_EINVAL;
  if (ce_label && (!ce_y || !ce_loss)) return TSGNN_EINVAL;
  if ((P % 4) || P > 2048 || E > 4096 || B > 1024 || (reinterpret_cast<uintptr_t>(w1) & 15)) return TSGNN_EUNSUPPORTED;
  size_t lds = sizeof(float) * (size_t)(((E + 3) & ~3) + HW * ((P + 3) & ~3));
  if (lds < sizeof(float) * (4 * (size_t)B + HW)) lds = sizeof(float) * (4 * (size_t)B + HW);
  if (ce_label) lds += sizeof(float) * (size_t)(((B * C + 3) & ~3) + ((B + 3) & ~3));
  TSGNN_KNAME("head2_bwd_kernel");
  head2_bwd_kernel<<<B + (E + 3) / 4 + 1, 64 * HW, lds, stream>>>(out, ldo, vec, dy, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2,
                                                                db2, normparts, ce_y, ce_label, ce_loss);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_head2_bwd_f32(const float* out, int64_t ldo, const float* vec, const float* dy, const float* dvec, const float* w1,
                        const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo, float* dw1, float* db1,
                        float* dw2, float* db2, float* normparts, tsgnn_stream_t stream) {
  return head2_bwd_launch(out, ldo, vec, dy, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2, db2, normparts, nullptr, nullptr,
                          nullptr, stream);
}

/* the same with the loss folded in: dy = d mean-softmax-cross-entropy(y, label) / dy is rebuilt inside the kernel and the loss
 * value written to loss[0] (F.cross_entropy of encoders.py:221-224 without a launch of its own) */
int tsgnn_head2_bwd_ce_f32(const float* out, int64_t ldo, const float* vec, const float* y, const int64_t* label, float* loss,
                           const float* dvec, const float* w1, const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo,
                           float* dw1, float* db1, float* dw2, float* db2, float* normparts, tsgnn_stream_t stream) {
  if (!y || !label || !loss) return TSGNN_EINVAL;
  return head2_bwd_launch(out, ldo, vec, nullptr, dvec, w1, w2, B, P, E, C, dout, lddo, dw1, db1, dw2, db2, normparts, y, label, loss,
                          stream);
}

}  // extern "C"
