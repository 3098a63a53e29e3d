// Developer harness: per-wave phase timeline of gemm_tn_rows_kernel (weight-gradient slabs), built as
//   hipcc --offload-arch=gfx950 -O3 -DTSGNN_TRACE scripts/trace_tn_rows.hip -o scripts/_build/trace_tn_rows
#include "../two-stage-gnn_amd/csrc/gemm.hip"
#include "trace_util.h"
#include <cstdio>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
  const int64_t R = argc > 1 ? atoll(argv[1]) : 9151;
  const int K = 128, N = 128;
  int nslab; int64_t rps, need;
  tsgnn_linear_wgrad_plan(R, K, N, K, N, &nslab, &rps, &need);
  float *z, *du, *ws, *dw, *db;
  (void)hipMalloc(&z, R * K * 4); (void)hipMalloc(&du, R * N * 4); (void)hipMalloc(&ws, need * 4); (void)hipMalloc(&dw, K * N * 4); (void)hipMalloc(&db, N * 4);
  std::vector<float> h(R * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  (void)hipMemcpy(z, h.data(), R * K * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(du, h.data(), R * N * 4, hipMemcpyHostToDevice);
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 20; ++it) tsgnn_linear_wgrad_f32(z, K, du, N, R, K, N, nslab, rps, 0, ws, nullptr, nullptr, s);
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 200; ++it) tsgnn_linear_wgrad_f32(z, K, du, N, R, K, N, nslab, rps, 0, ws, nullptr, nullptr, s);
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  printf("nslab=%d rows_per_slab=%lld: %.2f us per launch (back-to-back, incl. trace stores)\n", nslab, (long long)rps, ms * 1000 / 200);
  const int nw = std::min(nslab * 2, 1024) * 4;
  const int last = 12;
  trace_report(t, nslab * 2, last);
  for (int w : {0, 1, 2, 3, 400, 401})
    if (w < nw) { printf("  wave %d:", w); for (int k = 0; k <= last; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  return 0;
}
