// Developer harness: per-wave phase timeline of rowgemm_kernel (s_memtime stamps), built as
//   hipcc --offload-arch=gfx950 -O3 -DTSGNN_TRACE scripts/trace_rowgemm.hip -o gpurun_out/trace_rowgemm
// (add -DTSGNN_TRACE_WPB=8 for the gather variant: its row panels are 512-thread blocks, waves 4-7 = the second K group)
#include "../two-stage-gnn_amd/csrc/rowgemm.hip"
#include "trace_util.h"
thread_local char tsgnn_kname_[160] = "";
#include <cstdio>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
  const int64_t R = argc > 1 ? atoll(argv[1]) : 9151;
  const int K = 128, N = 128;
  float *a, *b, *bias, *c, *rinv;
  hipMalloc(&a, R * K * 4); hipMalloc(&b, K * N * 4); hipMalloc(&bias, N * 4); hipMalloc(&c, R * N * 4); hipMalloc(&rinv, R * 4);
  std::vector<float> h(R * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(a, h.data(), R * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), K * N * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), N * 4, hipMemcpyHostToDevice);
  // synthetic neighbour table: degree 2..8 (avg 5) inside a +-300 row window (DD-like locality), width 16
  std::vector<int> hell(R * 16, -1);
  for (int64_t r = 0; r < R; ++r) {
    const int deg = 2 + (int)((r * 2654435761u >> 7) % 7);
    for (int k = 0; k < deg; ++k) {
      long long j = r + (long long)(((r * 40503u + k * 9973u) >> 3) % 600) - 300;
      hell[r * 16 + k] = (int)std::min<long long>(std::max<long long>(j, 0), R - 1);
    }
  }
  int* ell; hipMalloc(&ell, R * 16 * 4); hipMemcpy(ell, hell.data(), R * 16 * 4, hipMemcpyHostToDevice);
  float* zout; hipMalloc(&zout, R * K * 4);
  const bool gather = argc > 2 && atoi(argv[2]) != 0;
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int trans = 0; trans < 2; ++trans) {
    auto run = [&]() {
      if (gather) tsgnn_gather_rowgemm_f32(ell, 16, nullptr, nullptr, a, K, b, N, trans, trans ? nullptr : bias, c, N, trans ? nullptr : rinv,
                                           trans ? nullptr : zout, K, R, K, N, trans ? 0 : 1, 0, s);
      else tsgnn_rowgemm_f32(a, K, b, N, trans, trans ? nullptr : bias, c, N, trans ? nullptr : rinv, R, K, N, trans ? 0 : 1, 0, s);
    };
    for (int it = 0; it < 20; ++it) run();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int it = 0; it < 200; ++it) run();
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t(4096 * 16);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
    const int nb = (int)std::min<int64_t>((R + 31) / 32, 1024);
    printf("trans_b=%d  %.2f us per launch (back-to-back, incl. trace stores)\n", trans, ms * 1000 / 200);
    const int last = trans ? 10 : 13;
    trace_report(t, nb, last);
    for (int w : {0, 1, 4, 5, 400, 404})
      if (w < nb * TSGNN_TRACE_WPB) { printf("  wave %d:", w); for (int k = 0; k <= last; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  }
  return 0;
}
