// Developer harness: per-wave phase timeline of sage_layer_bwd_kernel (row panels dX = (A dU) W^T  ||  weight-gradient slabs), built as
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTSGNN_TRACE scripts/trace_layer_bwd.hip -o scripts/_build/trace_layer_bwd
#include "../two-stage-gnn_amd/csrc/layer_bwd.hip"
#include "trace_util.h"
thread_local char tsgnn_kname_[160] = "";
extern "C" int tsgnn_linear_wgrad_plan(int64_t, int, int, int64_t, int64_t, int*, int64_t*, int64_t*) { return 0; }
#include <cstdio>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
  const int64_t R = argc > 1 ? atoll(argv[1]) : 8151;
  const int K = 128, N = 128;
  const int nslab = argc > 2 ? atoi(argv[2]) : 128;
  const int64_t rps = ((R + nslab - 1) / nslab + 7) / 8 * 8;
  float *du, *w, *dxs, *z, *ws;
  hipMalloc(&du, (R + 1024) * N * 4); hipMalloc(&w, K * N * 4); hipMalloc(&dxs, R * K * 4); hipMalloc(&z, R * K * 4);
  hipMalloc(&ws, (size_t)nslab * (K + 1) * N * 4);
  std::vector<float> h((R + 1024) * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(du, h.data(), (R + 1024) * N * 4, hipMemcpyHostToDevice);
  hipMemcpy(z, h.data(), R * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), K * N * 4, hipMemcpyHostToDevice);
  std::vector<int> hell(R * 16, -1);
  for (int64_t r = 0; r < R; ++r) {
    const int deg = 2 + (int)((r * 2654435761u >> 7) % 7);
    for (int k = 0; k < deg; ++k) {
      long long j = r + (long long)(((r * 40503u + k * 9973u) >> 3) % 600) - 300;
      hell[r * 16 + k] = (int)std::min<long long>(std::max<long long>(j, 0), R - 1);
    }
  }
  int* ell; hipMalloc(&ell, R * 16 * 4); hipMemcpy(ell, hell.data(), R * 16 * 4, hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&]() { tsgnn_sage_layer_bwd_f32(ell, 16, nullptr, nullptr, du, N, w, N, dxs, K, z, K, R, nslab, rps, 500, ws, s); };
  for (int it = 0; it < 20; ++it) run();
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int it = 0; it < 200; ++it) run();
  hipEventRecord(e1, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  const int npan = (int)((R + 31) / 32), nblk = npan + 2 * nslab;
  printf("rows %lld: %d panels + %d slab blocks (rows per slab %lld): %.2f us per launch (back-to-back, incl. trace stores)\n", (long long)R, npan,
         2 * nslab, (long long)rps, ms * 1000 / 200);
  // wall clock per role (slots 14 = wave start, 15 = wave end; 10 ns units)
  long long s0 = 1LL << 62;
  const int nw = std::min(nblk, 1024) * 4;
  for (int wv = 0; wv < nw; ++wv) s0 = std::min(s0, t[wv * 16 + 14]);
  auto role = [&](const char* name, int b0, int b1, int last) {
    std::vector<double> st, en; double avg[16] = {0}; int n = 0;
    for (int b = b0; b < std::min(b1, 1024); ++b)
      for (int wv = 0; wv < 4; ++wv) {
        const long long* q = &t[(b * 4 + wv) * 16];
        st.push_back((q[14] - s0) * 0.01); en.push_back((q[15] - s0) * 0.01);
        for (int k = 0; k <= last; ++k) avg[k] += (double)(q[k] - q[0]);
        ++n;
      }
    if (!n) return;
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
    printf("  %s: starts p50 %.2f max %.2f us; ends p10 %.2f p50 %.2f p90 %.2f max %.2f us\n    mean ticks since wave start:", name, st[n / 2], st.back(),
           en[n / 10], en[n / 2], en[n * 9 / 10], en.back());
    for (int k = 0; k <= last; ++k) printf(" [%d]%.0f", k, avg[k] / n);
    printf("\n");
  };
  role("row panels", 0, npan, 13);
  role("slab blocks", npan, nblk, 12);
  return 0;
}
