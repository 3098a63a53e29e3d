// Developer probe: what does a kernel boundary cost inside a replayed hipGraph chain?
//   hipcc --offload-arch=gfx950 -O3 scripts/graph_gap_bench.hip -o scripts/_build/graph_gap_bench
// Chains of N dependent launches: (a) the same empty kernel, (b) N different empty kernels, (c) the same kernel writing W MB that the next
// launch reads (dirty lines written back at the boundary, refetched behind it), each timed per launch from hipGraph replays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ID>
__global__ void k_empty(float* p) { if (p && threadIdx.x == 1023 + ID) p[ID] = (float)ID; }
__global__ void k_stream(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = in[i]; v.x += 1.f; out[i] = v;
  }
}
static float replay_us(hipGraphExec_t ex, hipStream_t s, int reps, int per) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) (void)hipGraphLaunch(ex, s);
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ex, s);
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps / per;
}
template <typename F>
static hipGraphExec_t capture(hipStream_t s, F&& f) {
  hipGraph_t g; hipGraphExec_t ex;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  f();
  (void)hipStreamEndCapture(s, &g);
  (void)hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  return ex;
}
int main() {
  hipStream_t s; (void)hipStreamCreate(&s);
  float* d; (void)hipMalloc(&d, 1 << 20);
  const int N = 12;
  auto same = capture(s, [&] { for (int i = 0; i < N; ++i) k_empty<0><<<256, 256, 0, s>>>(d); });
  auto diff = capture(s, [&] {
    k_empty<0><<<256, 256, 0, s>>>(d); k_empty<1><<<256, 256, 0, s>>>(d); k_empty<2><<<256, 256, 0, s>>>(d); k_empty<3><<<256, 256, 0, s>>>(d);
    k_empty<4><<<256, 256, 0, s>>>(d); k_empty<5><<<256, 256, 0, s>>>(d); k_empty<6><<<256, 256, 0, s>>>(d); k_empty<7><<<256, 256, 0, s>>>(d);
    k_empty<8><<<256, 256, 0, s>>>(d); k_empty<9><<<256, 256, 0, s>>>(d); k_empty<10><<<256, 256, 0, s>>>(d); k_empty<11><<<256, 256, 0, s>>>(d); });
  printf("chain of %d empty launches, same kernel: %.2f us per launch; %d different kernels: %.2f us per launch\n", N, replay_us(same, s, 200, N), N,
         replay_us(diff, s, 200, N));
  for (size_t mb : {1, 4, 8, 16, 32}) {
    const size_t n4 = mb * (1 << 20) / 16;
    float4 *a, *b; (void)hipMalloc(&a, n4 * 16); (void)hipMalloc(&b, n4 * 16);
    (void)hipMemset(a, 0, n4 * 16); (void)hipMemset(b, 0, n4 * 16);
    auto st = capture(s, [&] { for (int i = 0; i < N; i += 2) { k_stream<<<1024, 256, 0, s>>>(a, b, n4); k_stream<<<1024, 256, 0, s>>>(b, a, n4); } });
    const float us = replay_us(st, s, 100, N);
    printf("ping-pong stream kernel, %zu MB read + %zu MB written per launch, consumer = next launch: %.2f us per launch (%.0f GB/s)\n", mb, mb, us,
           2.0 * mb * 1048576 / us / 1e3);
    (void)hipFree(a); (void)hipFree(b);
  }
  return 0;
}
