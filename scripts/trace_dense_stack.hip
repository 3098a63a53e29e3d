// Developer harness: per-wave phase timeline of dense_stack_fwd_kernel (level-1 shapes of BASELINE config 5), built as
//   hipcc --offload-arch=gfx950 -O3 -DTSGNN_TRACE scripts/trace_dense_stack.hip -o scripts/_build/trace_dense_stack
#include "../two-stage-gnn_amd/csrc/dense_stack.hip"
#include "trace_util.h"
thread_local char tsgnn_kname_[160] = "";
#include <cstdio>
#include <vector>

int main() {
  const int B = 16, K = 64, fin0 = 192, H = 64, L = 3, nstack = 2;
  const int R = B * K;
  auto dmalloc = [](size_t n) { float* p; (void)hipMalloc(&p, n * 4); (void)hipMemset(p, 0, n * 4); return p; };
  std::vector<float> h((size_t)R * fin0);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  float* x = dmalloc(R * fin0); (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  float* adj = dmalloc(B * K * K); (void)hipMemcpy(adj, h.data(), B * K * K * 4, hipMemcpyHostToDevice);
  std::vector<int64_t> d(19 + 2 * (5 + 4 * 14), 0);
  auto P = [](void* p) { return (int64_t)(uintptr_t)p; };
  d[0] = P(x); d[1] = fin0; d[2] = fin0; d[3] = P(adj); d[4] = B; d[5] = K; d[6] = nstack;
  d[7] = P(dmalloc(4 * nstack * R * 2)); d[8] = P(dmalloc(64)); d[9] = P(dmalloc(4));
  int o = 19;
  for (int s = 0; s < 2; ++s) {
    const int widths[3] = {H, H, s == 0 ? 64 : 8};
    const int total = widths[0] + widths[1] + widths[2];
    d[o] = P(dmalloc((size_t)R * total)); d[o + 1] = total; d[o + 4] = L;
    o += 5;
    int fin = fin0, off = 0;
    for (int l = 0; l < 4; ++l) {
      if (l < L) {
        const int n = widths[l];
        float* w = dmalloc((size_t)fin * n); (void)hipMemcpy(w, h.data(), (size_t)fin * n * 4, hipMemcpyHostToDevice);
        d[o] = P(w); d[o + 1] = n; d[o + 2] = P(dmalloc(n)); d[o + 3] = fin; d[o + 4] = n; d[o + 5] = off;
        d[o + 6] = P(dmalloc((size_t)R * fin)); d[o + 7] = P(dmalloc((size_t)R * n)); d[o + 8] = P(dmalloc(R));
        d[o + 9] = P(dmalloc(K)); d[o + 10] = P(dmalloc(K));
        off += n; fin = n;
      }
      o += 14;
    }
  }
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) { int rc = tsgnn_dense_stack_fwd_f32(d.data(), s); if (rc) { printf("rc %d\n", rc); return 1; } }
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 50; ++it) tsgnn_dense_stack_fwd_f32(d.data(), s);
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  printf("dense_stack_fwd 2 stacks B=16 K=64: %.2f us per launch (back-to-back, incl. trace stores)\n", ms * 1000 / 50);
  trace_report(t, 128, 9);
  for (int w : {0, 1, 64, 300, 511}) { printf("  wave %d:", w); for (int k = 0; k <= 9; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  return 0;
}
