// Shared device/host helpers for libtsgnn_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TSGNN_OK 0
#define TSGNN_EINVAL (-1)
#define TSGNN_ELAUNCH (-2)
#define TSGNN_EUNSUPPORTED (-3)

#define TSGNN_WAVE 64

// every entry point: validate -> launch -> report launch errors (never sync, never allocate)
#define TSGNN_CHECK_LAUNCH()                              \
  do {                                                    \
    hipError_t e__ = hipGetLastError();                   \
    if (e__ != hipSuccess) return TSGNN_ELAUNCH;          \
  } while (0)

// developer instrumentation (scripts/trace_*.hip build a kernel file with -DTSGNN_TRACE): per-wave s_memtime stamps,
// 16 slots per wave, first 1024 blocks x 4 waves.  Compiled out of the library.
#ifdef TSGNN_TRACE
#ifndef TSGNN_TRACE_WPB
#define TSGNN_TRACE_WPB 4                 /* waves per block of the traced kernel (8 for the 512-thread split-K row panels) */
#endif
__device__ long long g_trace[4096 * 16];
#define TR_SLOT_(slot) g_trace[((blockIdx.x + gridDim.x * blockIdx.y) * TSGNN_TRACE_WPB + (threadIdx.x >> 6)) * 16 + (slot)]
#define TR_ON_ ((threadIdx.x & 63) == 0 && (blockIdx.x + gridDim.x * blockIdx.y) < 4096 / TSGNN_TRACE_WPB)
#define TR(slot) do { if (TR_ON_) { TR_SLOT_(slot) = __builtin_readcyclecounter(); if ((slot) == 0) TR_SLOT_(14) = wall_clock64(); } } while (0)
#define TR_END() do { if (TR_ON_) TR_SLOT_(15) = wall_clock64(); } while (0)   /* 100 MHz, common to the whole device */
/* stamp once the 32-bit value v has ARRIVED (the read makes the wave wait for the load that produces it) */
#define TR_AFTER(v, slot) do { if (__builtin_amdgcn_readfirstlane(v) != 0x7fffffff) TR(slot); } while (0)
#else
#define TR_AFTER(v, slot) do { } while (0)
#define TR(slot) do { } while (0)
#define TR_END() do { } while (0)
#endif

// diagnostic note: the device kernel the last entry point of this thread dispatched (tsgnn_last_kernel(); bench.py reports
// kernel names from the dispatch instead of hard-coding them).  Annotated at the dispatch sites of the training-step and
// aggregation entry points; a few dozen nanoseconds of host time per launch.
#include <cstdio>
extern thread_local char tsgnn_kname_[160];
#define TSGNN_KNAME(...) (void)snprintf(tsgnn_kname_, sizeof(tsgnn_kname_), __VA_ARGS__)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// store of a value that THIS kernel will not read again (results handed to a later launch): -DTSGNN_NT_STORES=1 marks them
// non-temporal.  Measured (round 3, A/B of two builds through TSGNN_LIB_PATH): the producers' own burst times drop (row panels
// 10.0 -> 9.4 / 14.5 -> 13.8 / 13.1 -> 12.1 us: less to write back at the kernel boundary) and the replayed STEP does not move at all
// (0.1295 ms both ways): what a producer saves its consumer pays on the read.  Off.
#ifndef TSGNN_NT_STORES
#define TSGNN_NT_STORES 0
#endif
typedef float tsgnn_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_out(float* p, float v) {
#if TSGNN_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st_out(float4* p, float4 v) {
#if TSGNN_NT_STORES
  tsgnn_f32x4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<tsgnn_f32x4*>(p));
#else
  *p = v;
#endif
}

// ---- cross-lane reductions on DPP (no LDS-crossbar round trips): quad_perm xor1 / xor2, row_half_mirror,
// row_mirror give every lane its 16-lane row total; rows are combined through v_readlane (SGPR broadcast).
// All 64 lanes of the wave must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float lane_f32(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);   // row_half_mirror
  v += dpp_f32<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}
__device__ __forceinline__ int wave_sum_i(int v) {
  v += dpp_i32<0xB1>(v);
  v += dpp_i32<0x4E>(v);
  v += dpp_i32<0x141>(v);
  v += dpp_i32<0x140>(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}
// Transposing reduction: every lane holds s[0..15]; lane l returns the sum of s[l & 15] over the 16 lanes of its DPP
// row.  Each step halves the values a lane carries (keep one half, hand the other half to the partner), so the whole
// thing is 15 DPP adds + 30 selects instead of 64 DPP adds.  Partner order row_mirror (l^15), row_half_mirror (l^7),
// quad xor 2, quad xor 1: each partner agrees on the bits already used and differs in the bit selected on.
__device__ __forceinline__ float row16_sum_transpose(const float (&s)[16]) {
  const unsigned l = threadIdx.x;
  const bool b3 = l & 8u, b2 = l & 4u, b1 = l & 2u, b0 = l & 1u;
  float a8[8], a4[4], a2[2];
#pragma unroll
  for (int p = 0; p < 8; ++p) a8[p] = (b3 ? s[p + 8] : s[p]) + dpp_f32<0x140>(b3 ? s[p] : s[p + 8]);
#pragma unroll
  for (int p = 0; p < 4; ++p) a4[p] = (b2 ? a8[p + 4] : a8[p]) + dpp_f32<0x141>(b2 ? a8[p] : a8[p + 4]);
#pragma unroll
  for (int p = 0; p < 2; ++p) a2[p] = (b1 ? a4[p + 2] : a4[p]) + dpp_f32<0x4E>(b1 ? a4[p] : a4[p + 2]);
  return (b0 ? a2[1] : a2[0]) + dpp_f32<0xB1>(b0 ? a2[0] : a2[1]);
}

// reduction inside an aligned power-of-two lane group of width G (<= 64); every lane of the group gets the result
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if (G >= 2) v += dpp_f32<0xB1>(v);
  if (G >= 4) v += dpp_f32<0x4E>(v);
  if (G >= 8) v += dpp_f32<0x141>(v);
  if (G >= 16) v += dpp_f32<0x140>(v);
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
  if (G >= 2) v = fmaxf(v, dpp_f32<0xB1>(v));
  if (G >= 4) v = fmaxf(v, dpp_f32<0x4E>(v));
  if (G >= 8) v = fmaxf(v, dpp_f32<0x141>(v));
  if (G >= 16) v = fmaxf(v, dpp_f32<0x140>(v));
  if (G >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  if (G >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// ---------------------------------------------------------------- Philox4x32-10 (Salmon et al., SC'11), counter-based
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}

// XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch, speed only).
// Gives every XCD one contiguous range of logical blocks so that neighbouring rows (which share
// gathered feature rows of the same graph) meet in one L2.  Bijective for any nblk.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// order-preserving float <-> uint map (for packed atomicMax of (value, index))
__device__ __forceinline__ unsigned f32_ordered(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_f32(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// (mean, rstd) of a batch-norm slot from the 64-bit fixed-point sums (2^-40 units) of its real rows + the ghost row's two numbers
// times its multiplicity (rowgemm_body.h STATS / BNIN, readout_body.h).  The combination runs in double (the sums are exact, so
// E[x^2] - mean^2 loses nothing); no double division or square root: one reciprocal of the count per call site, a float rsqrt at
// the end (four slots per thread with double div + sqrt were 2 us of a row panel's prologue).
__device__ __forceinline__ float2 bn_stats_from_sums(unsigned long long s1, unsigned long long s2, int have, float g1, float g2, int B,
                                                     double inv_cnt, float eps) {
  const double mult = (double)(B - have), fix = 1.0 / 1099511627776.0;
  const double m = fma((double)(long long)s1, fix, mult * (double)g1) * inv_cnt;
  const double e2 = fma((double)(long long)s2, fix, mult * (double)g2) * inv_cnt;
  const float var = fmaxf((float)(e2 - m * m), 0.f);
  return make_float2((float)m, 1.0f / sqrtf(var + eps));
}
