// micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 for ONE dependent accumulator chain per wave against two / four
// interleaved independent chains, at 1 and 2 waves per SIMD:   hipcc --offload-arch=gfx950 -O3 scripts/mfma_chain_bench.hip -o scripts/_build/mfma_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void chain(float* out, int n, long long* cyc) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-3f;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
void run(int blocks_per_cu, int threads) {
  float* out; long long* cyc; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&cyc, 8);
  const int n = 4096 / CH;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  chain<CH><<<256 * blocks_per_cu, threads>>>(out, n, cyc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  chain<CH><<<256 * blocks_per_cu, threads>>>(out, n, cyc);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = 4096.0;
  const double waves = 256.0 * blocks_per_cu * threads / 64;
  printf("chains/wave %d, waves/SIMD %d: %.1f cycles per MFMA per wave; whole chip %.1f TF (fp32 MFMA peak 157.3)\n", CH, blocks_per_cu * threads / 256,
         (double)c / mf, waves * mf * 4096.0 / (ms * 1e-3) / 1e12);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<1>(1, 256); run<2>(1, 256); run<4>(1, 256);
  run<1>(2, 256); run<2>(2, 256); run<1>(4, 256); run<1>(1, 512);
  return 0;
}
