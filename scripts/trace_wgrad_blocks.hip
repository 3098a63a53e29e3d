// Developer harness: per-wave phase timeline of wgrad_blocks_kernel (blocked weight gradient of the GAT projections), built as
//   hipcc --offload-arch=gfx950 -O3 -DTSGNN_TRACE scripts/trace_wgrad_blocks.hip -o scripts/_build/trace_wgrad_blocks
#include "../two-stage-gnn_amd/csrc/gemm.hip"
#include "trace_util.h"
thread_local char tsgnn_kname_[160];
#include <cstdio>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
  const int64_t R = argc > 1 ? atoll(argv[1]) : 8518;
  const int K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 264;
  int nslab; int64_t rps, need;
  tsgnn_wgrad_blocks_plan(R, K, N, K, N, &nslab, &rps, &need);
  float *z, *du, *ws, *dw;
  (void)hipMalloc(&z, R * K * 4); (void)hipMalloc(&du, R * N * 4); (void)hipMalloc(&ws, need * 4); (void)hipMalloc(&dw, (size_t)K * N * 4);
  std::vector<float> h(R * std::max(K, N));
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  (void)hipMemcpy(z, h.data(), R * K * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(du, h.data(), R * N * 4, hipMemcpyHostToDevice);
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 20; ++it) tsgnn_wgrad_blocks_f32(z, K, du, N, R, K, N, nslab, rps, ws, dw, N, s);
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 200; ++it) tsgnn_wgrad_blocks_f32(z, K, du, N, R, K, N, nslab, rps, ws, dw, N, s);
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  const int nsets = ((K + 127) / 128) * ((N + 127) / 128);
  printf("R=%lld K=%d N=%d nslab=%d rows_per_slab=%lld sets=%d: %.2f us per product (2 launches, back-to-back, incl. trace stores)\n",
         (long long)R, K, N, nslab, (long long)rps, nsets, ms * 1000 / 200);
  const int last = 12;
  trace_report(t, nslab * 2 * nsets, last);
  for (int w : {0, 1, 2, 3, 400, 401})
    { printf("  wave %d:", w); for (int k = 0; k <= last; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  return 0;
}
