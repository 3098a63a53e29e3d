// Optimiser step of the reference's training loop on ONE flat fp32 buffer (train.py:128-129:
// nn.utils.clip_grad_norm(model.parameters(), clip) then Adam.step()), so that the data-parallel
// all-reduce (one RCCL call on the same flat gradient buffer) and the update are 1 + 1 launches per
// step (1 + 3 for models over 131,072 parameters) instead of ~10 launches per parameter tensor.  Graph-replay safe: the step counter lives in
// device memory.
#include "common.h"
#include "../../include/tsgnn.h"

namespace {

constexpr int NPART = 256;

// `poison` (nullable): the device's error word (message_passing.device_error_word).  A kernel with a bounded device-wide barrier
// that could not complete it (dense_stack.hip) stores a non-zero value there: the gradients of that step are invalid.  Every
// optimiser kernel reads the word first and, if it is set, leaves parameters, moments and the step counter untouched and marks
// state[3] = 1 — the host raises at its next synchronisation (check_device_errors), and nothing wrong was applied meanwhile.
__device__ __forceinline__ bool poisoned(const float* __restrict__ poison) {
  return poison != nullptr && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.f;
}

__global__ __launch_bounds__(256) void sqnorm_partial(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float lds[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s = fmaf(g[i], g[i], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
// state[0] = step count (float), state[1] = total grad norm of the last step, state[2] = clip coefficient
__global__ __launch_bounds__(256) void sqnorm_final(const float* __restrict__ part, int nparts, float grad_scale, float max_norm,
                                                    float* __restrict__ state, const float* __restrict__ poison) {
  __shared__ float lds[4];
  if (poisoned(poison)) { if (threadIdx.x == 0) state[3] = 1.f; return; }
  float s = threadIdx.x < nparts ? part[threadIdx.x] : 0.f;
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf(lds[0] + lds[1] + lds[2] + lds[3]) * grad_scale;
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    state[0] += 1.f;
    state[1] = norm;
    state[2] = coef * grad_scale;
  }
}
__global__ void adam_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float wd, const float* __restrict__ state,
                            const float* __restrict__ poison) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || poisoned(poison)) return;
  const float step = state[0], scale = state[2];
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  float gi = g[i] * scale;
  if (wd != 0.f) gi = fmaf(wd, p[i], gi);
  const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
  const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}


// One-launch variant for small models, WITHOUT a device-wide barrier: every block computes the norm of the WHOLE gradient
// itself (same fixed order in every block -> the same bits everywhere; the buffer is <= 512 KB and sits in L2 after the
// all-reduce wrote it), then clips and runs Adam on its own 256*CV elements.  Nothing waits for another block, so there is
// no co-residency assumption, no spin, no timeout and therefore no way to apply a partial update (round 1's variant met at
// a bounded spin barrier and could give up half-way).  Redundant reads: nblocks x n x 4 B of L2 traffic (61 k parameters:
// 60 blocks x 244 KB = 15 MB), a few microseconds of latency instead of a second launch.
constexpr int CV = 4;               // elements per thread
#ifndef TSGNN_SELFNORM_MAX_N
#define TSGNN_SELFNORM_MAX_N 131072
#endif
constexpr int64_t SELFNORM_MAX_N = TSGNN_SELFNORM_MAX_N;
__global__ __launch_bounds__(256) void clip_adam_selfnorm(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                          float wd, float max_norm, float grad_scale, float* __restrict__ state,
                                                          unsigned* __restrict__ done, const float* __restrict__ poison) {
  __shared__ float lds[4];
  const int tid = threadIdx.x;
  if (poisoned(poison)) {                               // sign off all the same: `done` must be re-armed for the next launch
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == gridDim.x - 1) { __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); state[3] = 1.f; }
    }
    return;
  }
  const int64_t base = ((int64_t)blockIdx.x * 256 + tid) * CV;
  float gv[CV], pv[CV], mv[CV], vv[CV];                // the update's operands travel while the norm is summed
#pragma unroll
  for (int c = 0; c < CV; ++c) {
    const bool ok = (base + c) < n;
    gv[c] = ok ? g[base + c] : 0.f; pv[c] = ok ? p[base + c] : 0.f; mv[c] = ok ? m[base + c] : 0.f; vv[c] = ok ? v[base + c] : 0.f;
  }
  const float step = state[0] + 1.f;                   // every block reads the OLD counter; whichever block retires last stores
                                                       // the new one (see the end of the kernel): nobody waits for anybody
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  const int64_t n4 = n >> 2;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t i = tid;
  // SIXTEEN independent 16-byte loads in flight per thread, then four: every block reads the whole gradient out of the L2s, and with
  // four requests in flight a thread's 58 (61 k parameters) to 128 requests were 15 to 32 dependent round trips — 13.4 us for the SAGPool
  // network's 59 k parameters.  The partial sums are added in the order of the four-at-a-time loop (s0 .. s3 take every fourth request):
  // the same bits.
  for (; i + 15 * 256 < n4; i += 16 * 256) {
    float4 q[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) q[u] = g4[i + u * 256];
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      s0 = fmaf(q[u].x, q[u].x, s0); s0 = fmaf(q[u].y, q[u].y, s0); s0 = fmaf(q[u].z, q[u].z, s0); s0 = fmaf(q[u].w, q[u].w, s0);
      s1 = fmaf(q[u + 1].x, q[u + 1].x, s1); s1 = fmaf(q[u + 1].y, q[u + 1].y, s1); s1 = fmaf(q[u + 1].z, q[u + 1].z, s1); s1 = fmaf(q[u + 1].w, q[u + 1].w, s1);
      s2 = fmaf(q[u + 2].x, q[u + 2].x, s2); s2 = fmaf(q[u + 2].y, q[u + 2].y, s2); s2 = fmaf(q[u + 2].z, q[u + 2].z, s2); s2 = fmaf(q[u + 2].w, q[u + 2].w, s2);
      s3 = fmaf(q[u + 3].x, q[u + 3].x, s3); s3 = fmaf(q[u + 3].y, q[u + 3].y, s3); s3 = fmaf(q[u + 3].z, q[u + 3].z, s3); s3 = fmaf(q[u + 3].w, q[u + 3].w, s3);
    }
  }
  for (; i + 3 * 256 < n4; i += 4 * 256) {
    const float4 a = g4[i], b = g4[i + 256], c = g4[i + 512], d = g4[i + 768];
    s0 = fmaf(a.x, a.x, s0); s0 = fmaf(a.y, a.y, s0); s0 = fmaf(a.z, a.z, s0); s0 = fmaf(a.w, a.w, s0);
    s1 = fmaf(b.x, b.x, s1); s1 = fmaf(b.y, b.y, s1); s1 = fmaf(b.z, b.z, s1); s1 = fmaf(b.w, b.w, s1);
    s2 = fmaf(c.x, c.x, s2); s2 = fmaf(c.y, c.y, s2); s2 = fmaf(c.z, c.z, s2); s2 = fmaf(c.w, c.w, s2);
    s3 = fmaf(d.x, d.x, s3); s3 = fmaf(d.y, d.y, s3); s3 = fmaf(d.z, d.z, s3); s3 = fmaf(d.w, d.w, s3);
  }
  for (; i < n4; i += 256) {
    const float4 a = g4[i];
    s0 = fmaf(a.x, a.x, s0); s0 = fmaf(a.y, a.y, s0); s0 = fmaf(a.z, a.z, s0); s0 = fmaf(a.w, a.w, s0);
  }
  for (int64_t k = (n4 << 2) + tid; k < n; k += 256) s1 = fmaf(g[k], g[k], s1);
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((tid & 63) == 0) lds[tid >> 6] = s;
  __syncthreads();
  const float norm = sqrtf((lds[0] + lds[1]) + (lds[2] + lds[3])) * grad_scale;
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
  const float scale = coef * grad_scale;
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
#pragma unroll
  for (int c = 0; c < CV; ++c) {
    const int64_t k = base + c;
    if (k < n) {
      float gi = gv[c] * scale;
      if (wd != 0.f) gi = fmaf(wd, pv[c], gi);
      const float mi = fmaf(b1, mv[c], (1.f - b1) * gi);
      const float vi = fmaf(b2, vv[c], (1.f - b2) * gi * gi);
      m[k] = mi; v[k] = vi;
      const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
      p[k] = pv[c] - (lr / bc1) * (mi / denom);
    }
  }
  // Step counter: each block has read state[0] by now; it signs off on `done` (release) and the block that finds itself last
  // (acquire: every other block's read of state[0] happened before) stores the new counter and re-arms `done` for the next
  // launch / graph replay.  A sign-off, not a barrier: no block ever waits.
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      state[0] = step; state[1] = norm; state[2] = scale;
    }
  }
}

// Barrier-free variant for a single GPU: the kernels that PRODUCE the gradients (tn_rows_reduce_multi, head2_bwd) already
// left their share of |grad|^2 in parts[0..nparts) and bumped the step counter, so every block can finish the norm itself
// (same fixed-order sum everywhere) and go straight to the update.
__global__ __launch_bounds__(256) void adam_from_partials(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                          float wd, float max_norm, float* __restrict__ state,
                                                          const float* __restrict__ parts, int nparts, const float* __restrict__ poison) {
  __shared__ float lds[4];
  const int tid = threadIdx.x;
  if (poisoned(poison)) {                               // (the producers advanced the counter for this step: take that back)
    if (blockIdx.x == 0 && tid == 0) { state[0] -= 1.f; state[3] = 1.f; }
    return;
  }
  const int64_t base = ((int64_t)blockIdx.x * 256 + tid) * CV;
  float gv[CV], pv[CV], mv[CV], vv[CV];
#pragma unroll
  for (int c = 0; c < CV; ++c) {
    const bool ok = (base + c) < n;
    gv[c] = ok ? g[base + c] : 0.f; pv[c] = ok ? p[base + c] : 0.f; mv[c] = ok ? m[base + c] : 0.f; vv[c] = ok ? v[base + c] : 0.f;
  }
  // (the shares, ~3 per thread: requested together — a `t += parts[k]` loop was one dependent L2 round trip per iteration —
  // and added in the loop's order: the same bits)
  float t = 0.f;
  for (int k0 = tid; k0 < nparts; k0 += 4 * 256) {
    float q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = parts[min(k0 + 256 * u, nparts - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (k0 + 256 * u < nparts) t += q[u];
  }
  t = wave_sum(t);
  if ((tid & 63) == 0) lds[tid >> 6] = t;
  __syncthreads();
  const float norm = sqrtf((lds[0] + lds[1]) + (lds[2] + lds[3]));
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
  const float step = state[0];                           // already counts this step (the gradient reduction bumped it)
  if (blockIdx.x == 0 && tid == 0) { state[1] = norm; state[2] = coef; }
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
#pragma unroll
  for (int c = 0; c < CV; ++c) {
    const int64_t i = base + c;
    if (i < n) {
      float gi = gv[c] * coef;
      if (wd != 0.f) gi = fmaf(wd, pv[c], gi);
      const float mi = fmaf(b1, mv[c], (1.f - b1) * gi);
      const float vi = fmaf(b2, vv[c], (1.f - b2) * gi * gi);
      m[i] = mi; v[i] = vi;
      const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
      p[i] = pv[c] - (lr / bc1) * (mi / denom);
    }
  }
}


// softmax cross-entropy (mean over rows) with its gradient in the same pass: model.loss() of the reference
// (F.cross_entropy, encoders.py:221-224).  One thread per row (few classes), block reduction of the loss.
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ label,
                                                         int B, int C, float* __restrict__ loss, float* __restrict__ dlogits) {
  __shared__ float lds[4];
  float part = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* row = logits + (int64_t)b * ld;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, row[c]);
    float d = 0.f;
    for (int c = 0; c < C; ++c) d += expf(row[c] - m);
    const int y = (int)label[b];
    const float logz = m + logf(d);
    part += logz - row[y];
    const float invB = 1.f / (float)B;
    for (int c = 0; c < C; ++c) dlogits[(int64_t)b * C + c] = (expf(row[c] - logz) - (c == y ? 1.f : 0.f)) * invB;
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) *loss = ((lds[0] + lds[1]) + (lds[2] + lds[3])) / (float)B;
}

}  // namespace

extern "C" {

int tsgnn_softmax_ce_f32(const float* logits, int64_t ld, const int64_t* label, int B, int C, float* loss, float* dlogits,
                         tsgnn_stream_t stream) {
  if (!logits || !label || !loss || !dlogits || B <= 0 || C <= 0 || ld < C) return TSGNN_EINVAL;
  TSGNN_KNAME("softmax_ce_kernel");
  softmax_ce_kernel<<<1, 256, 0, stream>>>(logits, ld, label, B, C, loss, dlogits);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* Single-GPU optimiser step whose gradient norm was prepared by the gradient producers: parts[0..nparts) hold shares of
 * |grad|^2 (tsgnn_wgrad_reduce_multi_f32 / tsgnn_head2_bwd_f32 normparts) and state[0] was already advanced by
 * tsgnn_wgrad_reduce_multi_f32(step_state).  One launch, no device-wide barrier. */
int tsgnn_adam_from_partials_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, float max_norm, float* state, const float* parts, int nparts,
                                 const float* poison, tsgnn_stream_t stream) {
  if (!param || !grad || !m || !v || !state || !parts || n <= 0 || nparts <= 0) return TSGNN_EINVAL;
  TSGNN_KNAME("adam_from_partials");
  adam_from_partials<<<(unsigned)ceil_div64(n, 256 * CV), 256, 0, stream>>>(param, grad, m, v, n, lr, beta1, beta2, eps, weight_decay,
                                                                           max_norm, state, parts, nparts, poison);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* One optimiser step on flat buffers: g *= grad_scale (1/world after the all-reduce), clip to max_norm
 * (<=0: off), Adam.  ws >= 258 floats, 8-byte aligned, zeroed once before the first step (word 256 is the sign-off counter
 * of the one-launch variant, which re-arms it itself); state = 4 floats {step, grad_norm, applied scale, skipped (1 after a
 * launch that found *poison != 0 and left everything untouched)}, zero before step 1.  n <= 131,072: ONE launch, no device-wide barrier (every block sums the whole norm); larger: three. */
int tsgnn_clip_adam_step_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float max_norm, float grad_scale, float* state,
                             float* ws, const float* poison, tsgnn_stream_t stream) {
  if (!param || !grad || !m || !v || !state || !ws || n <= 0) return TSGNN_EINVAL;
  if (reinterpret_cast<uintptr_t>(ws) & 7) return TSGNN_EINVAL;
  if (n <= SELFNORM_MAX_N) {
    const int nbc = (int)ceil_div64(n, 256 * CV);
    TSGNN_KNAME("clip_adam_selfnorm");
    clip_adam_selfnorm<<<nbc, 256, 0, stream>>>(param, grad, m, v, n, lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, state,
                                                reinterpret_cast<unsigned*>(ws + 256), poison);
    TSGNN_CHECK_LAUNCH();
    return TSGNN_OK;
  }
  int nb = (int)ceil_div64(n, 256 * 8);
  if (nb > NPART) nb = NPART;
  TSGNN_KNAME("sqnorm_partial + sqnorm_final + adam_update");
  sqnorm_partial<<<nb, 256, 0, stream>>>(grad, n, ws);
  sqnorm_final<<<1, 256, 0, stream>>>(ws, nb, grad_scale, max_norm, state, poison);
  adam_update<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(param, grad, m, v, n, lr, beta1, beta2, eps, weight_decay, state, poison);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

}  // extern "C"
