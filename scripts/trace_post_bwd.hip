// Developer harness: per-wave phase timeline of slot_post_bwd on a DD-like batch (32 graphs, sizes 269 +- 94), built as
//   hipcc --offload-arch=gfx950 -O3 -DTSGNN_TRACE scripts/trace_post_bwd.hip -o scripts/_build/trace_post_bwd
#include "../two-stage-gnn_amd/csrc/sage_fused.hip"
#include "trace_util.h"
thread_local char tsgnn_kname_[160];
#include <cstdio>
#include <vector>
#include <algorithm>

int main() {
  const int B = 32, F = 128;
  std::vector<int> gp(B + 1, 0);
  int maxsz = 0;
  for (int b = 0; b < B; ++b) { const int sz = 120 + (int)((b * 2654435761u >> 8) % 330); gp[b + 1] = gp[b] + sz; maxsz = std::max(maxsz, sz); }
  const int n_real = gp[B], nslots = maxsz + 1, R = n_real + nslots;
  std::vector<int> sc(nslots, 0);
  for (int n = 0; n < nslots; ++n) for (int b = 0; b < B; ++b) if (gp[b + 1] - gp[b] > n) ++sc[n];
  std::vector<float> h((size_t)R * F);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.4f;
  std::vector<int> arg(B * F);
  for (int b = 0; b < B; ++b) for (int f = 0; f < F; ++f) arg[b * F + f] = gp[b] + (int)(((b * 131 + f * 7919u) >> 2) % (gp[b + 1] - gp[b]));
  int *d_gp, *d_sc, *d_arg; float *v, *dxs, *dout, *mean, *rstd, *rinv, *du;
  (void)hipMalloc(&d_gp, (B + 1) * 4); (void)hipMalloc(&d_sc, nslots * 4); (void)hipMalloc(&d_arg, B * F * 4);
  (void)hipMalloc(&v, (size_t)R * F * 4); (void)hipMalloc(&dxs, (size_t)R * F * 4); (void)hipMalloc(&du, (size_t)R * F * 4);
  (void)hipMalloc(&dout, B * F * 4); (void)hipMalloc(&mean, nslots * 4); (void)hipMalloc(&rstd, nslots * 4); (void)hipMalloc(&rinv, R * 4);
  (void)hipMemcpy(d_gp, gp.data(), (B + 1) * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_sc, sc.data(), nslots * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_arg, arg.data(), B * F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(v, h.data(), (size_t)R * F * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dxs, h.data(), (size_t)R * F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dout, h.data(), B * F * 4, hipMemcpyHostToDevice); (void)hipMemcpy(mean, h.data(), nslots * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(rstd, h.data() + 5000, nslots * 4, hipMemcpyHostToDevice); (void)hipMemcpy(rinv, h.data() + 9000, R * 4, hipMemcpyHostToDevice);
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&]() { tsgnn_slot_post_bwd_f32(d_gp, d_sc, B, nslots, n_real, nslots, v, F, dxs, F, nullptr, 0, dout, F, d_arg, F, 1, 1, mean, rstd, rinv, du, F, s); };
  for (int it = 0; it < 20; ++it) run();
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 200; ++it) run();
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  printf("rows %d (+%d ghost slots): %.2f us per launch (back-to-back, incl. trace stores)\n", n_real, nslots, ms * 1000 / 200);
  trace_report(t, nslots, 6);
  for (int w : {0, 400, 1200, 1700})
    if (w < nslots * 4) { printf("  wave %d:", w); for (int k = 0; k <= 6; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  return 0;
}
