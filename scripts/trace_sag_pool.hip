// Developer harness: per-wave phase timeline of sag_pool_graph_kernel (SAGPool level tail, one workgroup per graph), built as
//   hipcc --offload-arch=gfx950 -O3 -DTSGNN_TRACE scripts/trace_sag_pool.hip -o scripts/_build/trace_sag_pool
// Synthetic IMDB-B-like batch: B graphs of 12..40 nodes, ~5 neighbours per node, symmetric, F = 128.
#include "../two-stage-gnn_amd/csrc/sagpool.hip"
#include "trace_util.h"
#include <cstdio>
#include <vector>
#include <set>
#include <cmath>
#include <algorithm>

template <class T> static T* dev(const std::vector<T>& h) {
  T* p; (void)hipMalloc(&p, std::max<size_t>(h.size(), 1) * sizeof(T));
  (void)hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, F = 128;
  unsigned rng = 12345u;
  auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  std::vector<int> gp(B + 1, 0), gpn(B + 1, 0);
  int max_seg = 0;
  for (int b = 0; b < B; ++b) {
    const int n = 12 + (int)(rnd() % 29);
    gp[b + 1] = gp[b] + n; gpn[b + 1] = gpn[b] + (n + 1) / 2; max_seg = std::max(max_seg, n);
  }
  const int N = gp[B], K = gpn[B];
  std::vector<std::set<int>> adj(N);
  for (int b = 0; b < B; ++b) {
    const int n = gp[b + 1] - gp[b];
    for (int e = 0; e < n * 5 / 2; ++e) {
      const int u = gp[b] + (int)(rnd() % n), v = gp[b] + (int)(rnd() % n);
      if (u != v) { adj[u].insert(v); adj[v].insert(u); }
    }
  }
  std::vector<int> rowptr(N + 1, 0), col;
  for (int r = 0; r < N; ++r) { for (int c : adj[r]) col.push_back(c); rowptr[r + 1] = (int)col.size(); }
  std::vector<float> dinv(N), selfw(N), y((size_t)N * F), ws(F), bs(1, 0.1f);
  for (int r = 0; r < N; ++r) { const float d = (float)adj[r].size() + 1.f; dinv[r] = 1.f / std::sqrt(d); selfw[r] = 1.f / d; }
  for (auto& v : y) v = (float)(rnd() % 2000) / 1000.f - 1.f;
  for (auto& v : ws) v = (float)(rnd() % 2000) / 1000.f - 1.f;
  int *d_rowptr = dev(rowptr), *d_col = dev(col), *d_gp = dev(gp), *d_gpn = dev(gpn);
  float *d_dinv = dev(dinv), *d_selfw = dev(selfw), *d_y = dev(y), *d_ws = dev(ws), *d_bs = dev(bs);
  float *score, *xp, *out, *dinv_n, *selfw_n, *aggn; int *perm, *new_id, *cnt, *arg, *rp_n, *re_n, *col_n;
  (void)hipMalloc(&score, N * 4); (void)hipMalloc(&xp, (size_t)K * F * 4); (void)hipMalloc(&out, (size_t)B * 2 * F * 4);
  (void)hipMalloc(&dinv_n, K * 4); (void)hipMalloc(&selfw_n, K * 4); (void)hipMalloc(&aggn, (size_t)K * F * 4);
  (void)hipMalloc(&perm, K * 4); (void)hipMalloc(&new_id, N * 4); (void)hipMalloc(&cnt, K * 4); (void)hipMalloc(&arg, (size_t)B * F * 4);
  (void)hipMalloc(&rp_n, K * 4); (void)hipMalloc(&re_n, K * 4); (void)hipMalloc(&col_n, std::max<size_t>(col.size(), 1) * 4);
  hipStream_t s; (void)hipStreamCreate(&s);
  const bool next = argc > 2 ? atoi(argv[2]) != 0 : true;
  auto run = [&]() {
    return tsgnn_sag_pool_graph_f32(d_y, F, d_rowptr, nullptr, d_col, d_dinv, d_selfw, d_ws, d_bs, d_gp, d_gpn, B, max_seg, F, score, perm,
                                    new_id, xp, F, cnt, out, 2 * F, arg, 0, rp_n, re_n, col_n, dinv_n, selfw_n, next ? aggn : nullptr,
                                    next ? F : 0, s);
  };
  for (int it = 0; it < 20; ++it) if (run() != 0) { printf("launch failed\n"); return 1; }
  (void)hipStreamSynchronize(s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 200; ++it) run();
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  printf("B=%d rows=%d kept=%d entries=%zu max_seg=%d next-level aggregation %s: %.2f us per launch (back-to-back, incl. trace stores)\n", B, N, K,
         col.size(), max_seg, next ? "on" : "off", ms * 1000 / 200);
  printf("  marks: 0 start, 1 t=relu(y).w, 2 scores+keys, 3 sort, 4 perm/new_id, 5 gather+readout partials, 6 counts, 7 scan+filter fill, 8 next agg, 9 combine\n");
  trace_report(t, B, 9);
  for (int w : {0, 1, 2, 3, 40, 41})
    if (w < B * 4) { printf("  wave %d:", w); for (int k = 0; k <= 9; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  // ---- backward of the same level: pooled-row gradients -> score layer backward -> du
  std::vector<float> hd((size_t)K * F), hr((size_t)B * 2 * F);
  for (auto& v : hd) v = (float)(rnd() % 2000) / 1000.f - 1.f;
  for (auto& v : hr) v = (float)(rnd() % 2000) / 1000.f - 1.f;
  float *d_dagg = dev(hd), *d_dread = dev(hr), *du, *part, *dws, *dbs;
  (void)hipMalloc(&du, (size_t)N * F * 4); (void)hipMalloc(&part, (size_t)B * (F + 4) * 4); (void)hipMalloc(&dws, F * 4); (void)hipMalloc(&dbs, 4);
  auto runb = [&]() {
    // next = 1: the gradient arrives as the next level's dagg (A^' applied in the kernel); 0: as dxp
    return tsgnn_sag_pool_graph_bwd_f32(d_y, F, score, new_id, d_gp, d_gpn, arg, next ? nullptr : d_dagg, next ? 0 : F, d_dread, 2 * F, d_rowptr,
                                        nullptr, d_col, d_dinv, d_selfw, d_ws, B, max_seg, F, du, F, part, dws, dbs, next ? d_dagg : nullptr,
                                        next ? F : 0, next ? rp_n : nullptr, next ? re_n : nullptr, next ? col_n : nullptr,
                                        next ? dinv_n : nullptr, next ? selfw_n : nullptr, s);
  };
  std::vector<long long> zero(4096 * 16, 0);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), zero.data(), zero.size() * 8);
  for (int it = 0; it < 20; ++it) if (runb() != 0) { printf("bwd launch failed\n"); return 1; }
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 200; ++it) runb();
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  printf("backward (+ partial-sum reduction launch): %.2f us per call\n", ms * 1000 / 200);
  printf("  marks: 0 start, 1 (A) gather/readout gradient + dscore, 2 (B) dt = A^ dscore, 3 (C) du + dw_s partials, 4 (D) graph partials\n");
  trace_report(t, B, 4);
  for (int w : {0, 1, 2, 3})
    if (w < B * 4) { printf("  wave %d:", w); for (int k = 0; k <= 4; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  return 0;
}
