// DiffPool link-prediction side loss (encoders.py:416-440), value AND gradient in one pass, without the
// [B, N, N] tensors the reference builds (pred_adj = S S^T, the element-wise BCE, the mask product):
//
//   L = 1/num_entries * sum_b sum_{i,j < n_b} [ -a_ij log(p_ij + eps) - (1 - a_ij) log(1 - p_ij + eps) ],  p = min(S S^T, clamp)
//     = 1/num_entries * [ sum_pairs f0(p_ij)  +  sum_{(i,j) in E} a_ij (f1(p_ij) - f0(p_ij)) ]
//   f0(p) = -log(1 - p + eps),  f1(p) = -log(p + eps)                    (the BCE is linear in a_ij)
//
// The first sum runs over all row pairs of each graph and needs no adjacency: row tiles of S are staged in LDS, the
// P tile and dL/dP tile live in registers / LDS only, and dS_i = 2 sum_j g0(p_ij) s_j is accumulated tile by tile
// (P is symmetric).  The second sum is an SDDMM over the CSR entries.  Rows of S are the packed rows of a GraphBatch
// (ghost rows excluded = the reference's adj_mask, encoders.py:433-436) or all padded rows (batch_num_nodes=None, :430).
// The reference clamps with an UNINITIALISED tensor (`torch.min(pred_adj, torch.Tensor(1).cuda())`, :424); `clamp` is a
// parameter here, 1.0 (the value the DiffPool authors' later code uses) by default — see DESIGN.md.
// FMA work (K <= 128 assignment columns, n_b^2 K flops per graph): an auxiliary loss, not a GEMM worth MFMA tiles.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

constexpr int LP_TILE = 32;
constexpr int LP_KMAX = 128;
constexpr float LP_EPS = 1e-7f;          // encoders.py:413

__device__ __forceinline__ float lp_clamp(float p, float clamp, float& gate) {
  // torch.min(p, c): gradient 1 where p < c, 1/2 on ties, 0 above
  gate = p < clamp ? 1.f : (p == clamp ? 0.5f : 0.f);
  return fminf(p, clamp);
}

// one block per 32-row slab of one graph; loops over the graph's row tiles.  LDS rows are 132 floats apart (16-byte
// aligned, 4 banks per row step): every fragment read is a conflict-free ds_read_b128 — with one block per CU the kernel
// is bound by LDS round trips, so reads are wide and batched (the first version, one b32 read per FMA, ran 5x slower).
constexpr int LP_LD = LP_KMAX + 4;
constexpr int LP_NY = 4;                 // column-tile chunks per row slab (grid.y): 4 blocks per CU hide each other's LDS / L2 round trips
// Y (nullable = S): the operand of the COLUMN index j, p_ij = <S_i, Y_j> — adj_hop > 1 (encoders.py:419-423) is
// sum_p (S S^T)^p = (S M) S^T with a small per-graph M, i.e. two different row operands; grad_scale: 2 for the symmetric
// one-operand form (row i receives entry (i, j) and its mirror), 1 when the caller runs one pass per operand.
__global__ __launch_bounds__(256) void linkpred_pairs_kernel(const float* __restrict__ S, int64_t lds_, const float* __restrict__ Y,
                                                             int64_t ldy, int K,
                                                             const int* __restrict__ slab_row_ptr, const int* __restrict__ slab_graph,
                                                             const int* __restrict__ graph_ptr, float clamp, float inv_entries,
                                                             float grad_scale, int count_loss,
                                                             float* __restrict__ dSp, int64_t ldd, int64_t rows,
                                                             float* __restrict__ part) {
  if (!Y) { Y = S; ldy = lds_; }
  __shared__ __attribute__((aligned(16))) float s_i[LP_TILE][LP_LD];
  __shared__ __attribute__((aligned(16))) float s_j[LP_TILE][LP_LD];
  __shared__ __attribute__((aligned(16))) float s_g[LP_TILE][LP_TILE + 4];
  __shared__ float s_red[4];
  const int slab = blockIdx.x;
  const int i0 = slab_row_ptr[slab], i1 = slab_row_ptr[slab + 1];
  const int b = slab_graph[slab];
  const int g0 = graph_ptr[b], g1 = graph_ptr[b + 1];
  const int tid = threadIdx.x;
  const int K4 = (K + 3) & ~3;                    // columns [K, K4) are zero in LDS
  const int nk4 = K4 >> 2;
  for (int t = tid; t < LP_TILE * K4; t += 256) {
    const int r = t / K4, k = t - r * K4;
    s_i[r][k] = (i0 + r < i1 && k < K) ? S[(int64_t)(i0 + r) * lds_ + k] : 0.f;
  }
  const int ti = tid >> 3, tj = tid & 7;          // P tile: thread (row ti, columns tj, tj + 8, tj + 16, tj + 24)
  const int kq = tid & 7;                         // dS accumulation: thread (row ti, k = 4 kq .. 4 kq + 3 (+ 32 q))
  float4 acc[LP_KMAX / 32];
#pragma unroll
  for (int q = 0; q < LP_KMAX / 32; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  float loss = 0.f;
  for (int j0 = g0 + (int)blockIdx.y * LP_TILE; j0 < g1; j0 += LP_TILE * LP_NY) {      // this block's share of the column tiles
    __syncthreads();
    for (int t = tid; t < LP_TILE * K4; t += 256) {
      const int r = t / K4, k = t - r * K4;
      s_j[r][k] = (j0 + r < g1 && k < K) ? Y[(int64_t)(j0 + r) * ldy + k] : 0.f;
    }
    __syncthreads();
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k4 = 0; k4 < nk4; ++k4) {
      const float4 a = *reinterpret_cast<const float4*>(&s_i[ti][4 * k4]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 c = *reinterpret_cast<const float4*>(&s_j[tj + 8 * u][4 * k4]);
        p[u] = fmaf(a.x, c.x, fmaf(a.y, c.y, fmaf(a.z, c.z, fmaf(a.w, c.w, p[u]))));
      }
    }
    const bool vi = i0 + ti < i1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool v = vi && (j0 + tj + 8 * u < g1);
      float gate;
      const float pc = lp_clamp(p[u], clamp, gate);
      const float om = 1.f - pc + LP_EPS;
      if (v) loss -= logf(om);
      s_g[ti][tj + 8 * u] = v ? gate / om : 0.f;                       // d f0 / d p
    }
    __syncthreads();
#pragma unroll 2
    for (int j4 = 0; j4 < LP_TILE / 4; ++j4) {
      const float4 g4 = *reinterpret_cast<const float4*>(&s_g[ti][4 * j4]);
      const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int q = 0; q < LP_KMAX / 32; ++q) {
          if (4 * kq + 32 * q < K4) {
            const float4 c = *reinterpret_cast<const float4*>(&s_j[4 * j4 + jj][4 * kq + 32 * q]);
            acc[q].x = fmaf(gv[jj], c.x, acc[q].x); acc[q].y = fmaf(gv[jj], c.y, acc[q].y);
            acc[q].z = fmaf(gv[jj], c.z, acc[q].z); acc[q].w = fmaf(gv[jj], c.w, acc[q].w);
          }
        }
      }
    }
  }
  if (i0 + ti < i1) {
    const float sc = grad_scale * inv_entries;
#pragma unroll
    for (int q = 0; q < LP_KMAX / 32; ++q) {
      const int k = 4 * kq + 32 * q;
      float* d = dSp + ((int64_t)blockIdx.y * rows + (i0 + ti)) * ldd + k;
      if (k < K) d[0] = sc * acc[q].x;
      if (k + 1 < K) d[1] = sc * acc[q].y;
      if (k + 2 < K) d[2] = sc * acc[q].z;
      if (k + 3 < K) d[3] = sc * acc[q].w;
    }
  }
  loss = wave_sum(loss);
  if ((tid & 63) == 0) s_red[tid >> 6] = loss;
  __syncthreads();
  if (tid == 0) part[slab * LP_NY + blockIdx.y] = count_loss ? ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * inv_entries : 0.f;
}

// edge corrections, one wave per row i: entries (i, j) of the CSR (and of its transpose when A is not symmetric):
//   loss += a_ij (f1 - f0)(p_ij);  dS_i += sym_factor * a_ij (f1 - f0)'(p_ij) s_j
__global__ __launch_bounds__(256) void linkpred_edges_kernel(const float* __restrict__ S, int64_t lds_, const float* __restrict__ Y,
                                                             int64_t ldy, int K,
                                                             const int* __restrict__ rowptr, const int* __restrict__ col,
                                                             const float* __restrict__ val, int64_t rows, float clamp,
                                                             float inv_entries, float grad_factor, int count_loss,
                                                             const float* __restrict__ dSp /*nullable: LP_NY partials to start from*/,
                                                             float* __restrict__ dS, int64_t ldd, float* __restrict__ part) {
  if (!Y) { Y = S; ldy = lds_; }
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  __shared__ float s_red[4];
  float loss = 0.f;
  if (i < rows) {
    const float a0 = lane < K ? S[i * lds_ + lane] : 0.f;
    const float a1 = lane + 64 < K ? S[i * lds_ + lane + 64] : 0.f;
    float d0 = 0.f, d1 = 0.f;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int eb = e0; eb < e1; eb += 64) {
      // the row's entries (and weights) with one coalesced load, then the neighbours' rows four at a time: entry -> row was a
      // dependent pair of round trips per neighbour
      const int me = eb + lane;
      const int cj = me < e1 ? col[me] : 0;
      const float av = me < e1 ? (val ? val[me] : 1.f) : 0.f;
      const int cnt = min(64, e1 - eb);
      for (int q0 = 0; q0 < cnt; q0 += 4) {
        float b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t j = __shfl(cj, min(q0 + u, cnt - 1), 64);
          b0[u] = (lane < K && q0 + u < cnt) ? Y[j * ldy + lane] : 0.f;
          b1[u] = (lane + 64 < K && q0 + u < cnt) ? Y[j * ldy + lane + 64] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (q0 + u < cnt) {                                            // wave-uniform
            const float p = wave_sum(fmaf(a0, b0[u], a1 * b1[u]));
            float gate;
            const float pc = lp_clamp(p, clamp, gate);
            const float a = __shfl(av, q0 + u, 64);
            const float om = 1.f - pc + LP_EPS, pp = pc + LP_EPS;
            loss += a * (logf(om) - logf(pp));                           // f1 - f0
            const float gc = a * gate * (-1.f / pp - 1.f / om) * grad_factor * inv_entries;
            d0 = fmaf(gc, b0[u], d0);
            d1 = fmaf(gc, b1[u], d1);
          }
        }
      }
    }
    float o0 = 0.f, o1 = 0.f;
    if (dSp != nullptr) {
#pragma unroll
      for (int y = 0; y < LP_NY; ++y) {
        if (lane < K) o0 += dSp[((int64_t)y * rows + i) * ldd + lane];
        if (lane + 64 < K) o1 += dSp[((int64_t)y * rows + i) * ldd + lane + 64];
      }
    } else {
      if (lane < K) o0 = dS[i * ldd + lane];
      if (lane + 64 < K) o1 = dS[i * ldd + lane + 64];
    }
    if (lane < K) dS[i * ldd + lane] = o0 + d0;
    if (lane + 64 < K) dS[i * ldd + lane + 64] = o1 + d1;
  }
  if (lane == 0) s_red[threadIdx.x >> 6] = (i < rows && count_loss) ? loss * inv_entries : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// loss[0] = sum of the partials, fixed order
__global__ __launch_bounds__(256) void linkpred_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ loss) {
  __shared__ float s[256];
  float a = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) a += part[k];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = s[0];
}

}  // namespace

extern "C" {

int tsgnn_linkpred_tile_rows(void) { return LP_TILE; }
int tsgnn_linkpred_chunks(void) { return LP_NY; }

int tsgnn_linkpred_loss_f32(const float* S, int64_t lds, int K, int64_t rows, const int* slab_row_ptr, const int* slab_graph,
                            int nslab, const int* graph_ptr, const int* rowptr, const int* col, const float* val,
                            const int* rowptr_t, const int* col_t, const float* val_t, float clamp, float inv_entries, float* dS,
                            int64_t ldd, float* ws, float* part, float* loss, tsgnn_stream_t stream) {
  if (!S || !slab_row_ptr || !slab_graph || !graph_ptr || !rowptr || !dS || !ws || !part || !loss || nslab <= 0 || rows <= 0 || K <= 0 ||
      lds < K || ldd < K)
    return TSGNN_EINVAL;
  if (K > LP_KMAX) return TSGNN_EUNSUPPORTED;
  if ((rowptr_t == nullptr) != (col_t == nullptr)) return TSGNN_EINVAL;
  const unsigned eblk = (unsigned)ceil_div64(rows, 4);
  linkpred_pairs_kernel<<<dim3((unsigned)nslab, LP_NY), 256, 0, stream>>>(S, lds, nullptr, 0, K, slab_row_ptr, slab_graph, graph_ptr, clamp,
                                                                         inv_entries, 2.f, 1, ws, ldd, rows, part);
  // symmetric adjacency: entry (i,j) and its mirror give row i the same term twice
  linkpred_edges_kernel<<<eblk, 256, 0, stream>>>(S, lds, nullptr, 0, K, rowptr, col, val, rows, clamp, inv_entries, rowptr_t ? 1.f : 2.f, 1, ws,
                                                  dS, ldd, part + nslab * LP_NY);
  int nparts = nslab * LP_NY + (int)eblk;
  if (rowptr_t) {
    linkpred_edges_kernel<<<eblk, 256, 0, stream>>>(S, lds, nullptr, 0, K, rowptr_t, col_t, val_t, rows, clamp, inv_entries, 1.f, 0, nullptr,
                                                    dS, ldd, part + nparts);
    nparts += (int)eblk;
  }
  linkpred_sum_kernel<<<1, 256, 0, stream>>>(part, nparts, loss);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* two-operand form for adj_hop > 1 (encoders.py:419-423): p_ij = <X_i, Y_j> over the row pairs of every graph and the CSR
 * entries (i, j).  Writes dX[i] = dL/dX_i = sum_j (dL/dp_ij) Y_j (the caller runs a second pass with the operands swapped and
 * the transposed CSR for dL/dY) and, with count_loss, the loss value.  part: nslab * chunks + ceil(rows / 4) floats. */
int tsgnn_linkpred_loss_xy_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int K, int64_t rows, const int* slab_row_ptr,
                               const int* slab_graph, int nslab, const int* graph_ptr, const int* rowptr, const int* col,
                               const float* val, float clamp, float inv_entries, int count_loss, float* dX, int64_t ldd, float* ws,
                               float* part, float* loss, tsgnn_stream_t stream) {
  if (!X || !Y || !slab_row_ptr || !slab_graph || !graph_ptr || !rowptr || !dX || !ws || !part || (count_loss && !loss) || nslab <= 0 ||
      rows <= 0 || K <= 0 || ldx < K || ldy < K || ldd < K)
    return TSGNN_EINVAL;
  if (K > LP_KMAX) return TSGNN_EUNSUPPORTED;
  const unsigned eblk = (unsigned)ceil_div64(rows, 4);
  linkpred_pairs_kernel<<<dim3((unsigned)nslab, LP_NY), 256, 0, stream>>>(X, ldx, Y, ldy, K, slab_row_ptr, slab_graph, graph_ptr, clamp,
                                                                         inv_entries, 1.f, count_loss, ws, ldd, rows, part);
  linkpred_edges_kernel<<<eblk, 256, 0, stream>>>(X, ldx, Y, ldy, K, rowptr, col, val, rows, clamp, inv_entries, 1.f, count_loss, ws, dX, ldd,
                                                  part + nslab * LP_NY);
  if (count_loss) linkpred_sum_kernel<<<1, 256, 0, stream>>>(part, nslab * LP_NY + (int)eblk, loss);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
