// Developer harness: per-wave phase timeline of dense_stack_fwd_kernel (level-1 shapes of BASELINE config 5), built as
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTSGNN_TRACE -DTSGNN_TRACE_WPB=8 [-DTSGNN_TRACE_BWD] scripts/trace_dense_stack.hip -o scripts/_build/trace_dense_stack
// (-DTSGNN_TRACE_BWD: the backward kernel's timeline instead; 8 waves per workgroup since the helper waves)
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
  std::vector<int64_t> d(19 + 2 * (5 + 4 * 14) + 1, 0);
  auto P = [](void* p) { return (int64_t)(uintptr_t)p; };
  d[0] = P(x); d[1] = fin0; d[2] = fin0; d[3] = P(adj); d[4] = B; d[5] = K; d[6] = nstack;
  d[7] = P(dmalloc(4 * nstack * R * 2)); d[8] = P(dmalloc(32 + 256)); d[9] = P(dmalloc(4));
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
#ifdef TSGNN_TRACE_BWD
  {   // backward operands: dout per stack, dagg / dxn / slabs, gradients
    const int finmax = fin0;
    int64_t slab = 0;
    int o2 = 19;
    for (int st = 0; st < 2; ++st) {
      const int total = (int)d[o2 + 1];
      d[o2 + 2] = P(dmalloc((size_t)R * total)); d[o2 + 3] = total;
      (void)hipMemcpy((void*)d[o2 + 2], h.data(), (size_t)R * total * 4 < h.size() * 4 ? (size_t)R * total * 4 : h.size() * 4, hipMemcpyHostToDevice);
      o2 += 5;
      int64_t off = 0;
      for (int l = 0; l < 4; ++l) {
        if (l < L) {
          const int fin = (int)d[o2 + 3], n = (int)d[o2 + 4];
          d[o2 + 11] = P(dmalloc((size_t)fin * n)); d[o2 + 12] = P(dmalloc(n)); d[o2 + 13] = off;
          off += (int64_t)(fin + 1) * n;
        }
        o2 += 14;
      }
      if (off > slab) slab = off;
    }
    const int tiles = (K + 15) / 16;
    d[10] = P(dmalloc((size_t)nstack * R * finmax)); d[11] = P(dmalloc((size_t)nstack * R * finmax));
    d[12] = P(dmalloc((size_t)tiles * B * nstack * slab)); d[13] = slab; d[14] = finmax;
    d[15] = P(dmalloc((size_t)R * fin0)); d[16] = fin0; d[17] = P(dmalloc((size_t)B * K * K)); d[18] = P(dmalloc((size_t)2 * R * K));
  }
#endif
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#ifdef TSGNN_TRACE_BWD
  { int rc = tsgnn_dense_stack_fwd_f32(d.data(), s); if (rc) { printf("fwd rc %d\n", rc); return 1; } }
#define RUN_ tsgnn_dense_stack_bwd_f32
#else
#define RUN_ tsgnn_dense_stack_fwd_f32
#endif
  for (int it = 0; it < 5; ++it) { int rc = RUN_(d.data(), s); if (rc) { printf("rc %d\n", rc); return 1; } }
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 50; ++it) RUN_(d.data(), s);
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
#ifdef TSGNN_TRACE_BWD
  const int last = 13;
  printf("dense_stack_bwd 2 stacks B=16 K=64: %.2f us per launch (back-to-back, incl. trace stores)\n", ms * 1000 / 50);
#else
  const int last = 9;
  printf("dense_stack_fwd 2 stacks B=16 K=64: %.2f us per launch (back-to-back, incl. trace stores)\n", ms * 1000 / 50);
#endif
  trace_report(t, 128, last);
  for (int w : {0, 1, 64, 300, 511}) { printf("  wave %d:", w); for (int k = 0; k <= last; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  return 0;
}
