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

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// reduction inside an aligned power-of-two lane group of width G (<= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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
