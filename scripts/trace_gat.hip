// Developer harness: per-wave phase timeline of the fused GAT attention kernels on a DD-like synthetic batch, built as
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTSGNN_TRACE scripts/trace_gat.hip -o scripts/_build/trace_gat
// usage: trace_gat [fwd|bwd] [mean_heads 0|1]
#include "../two-stage-gnn_amd/csrc/gat_fused.hip"
#include "trace_util.h"
thread_local char tsgnn_kname_[160];
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>

int main(int argc, char** argv) {
  const bool bwd = argc > 1 && !strcmp(argv[1], "bwd");
  const int mean = argc > 2 ? atoi(argv[2]) : 0;
  const int B = 32, n = 266, H = 4, Fh = 64, C = H * Fh, Ns = C + 2 * H, nmax = 1000;
  const int64_t R = (int64_t)B * n;
  // symmetric graph: ring +-1, +-2 inside every graph plus one random chord per node; the last row of a graph is edge-less (ghost)
  std::mt19937 rng(1);
  std::vector<std::vector<int>> adj(R);
  for (int b = 0; b < B; ++b) {
    const int m = n - 1;
    for (int i = 0; i < m; ++i) {
      for (int d : {1, 2}) { const int j = (i + d) % m; adj[b * n + i].push_back(b * n + j); adj[b * n + j].push_back(b * n + i); }
      const int j = (int)(rng() % m);
      if (j != i) { adj[b * n + i].push_back(b * n + j); adj[b * n + j].push_back(b * n + i); }
    }
  }
  std::vector<int> rp(R + 1, 0), col, gp(B + 1), rg(R), iso_idx(B), iso_ptr(B + 1);
  for (int64_t r = 0; r < R; ++r) {
    auto& a = adj[r]; std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end());
    rp[r + 1] = rp[r] + (int)a.size(); col.insert(col.end(), a.begin(), a.end());
  }
  const int nnz = rp[R];
  std::vector<int> eperm(nnz);
  for (int e = 0; e < nnz; ++e) eperm[e] = e;      // (symmetric with sorted rows: entry (i,j) of A^T row j sits where (j,i) sits in A; the harness only times)
  std::vector<float> iso_w(B, (float)(nmax - n + 1)), iso_row(R * H, 0.f);
  for (int b = 0; b <= B; ++b) { gp[b] = b * n; iso_ptr[b] = b; }
  for (int b = 0; b < B; ++b) { iso_idx[b] = b * n + n - 1; for (int h = 0; h < H; ++h) iso_row[(int64_t)(b * n + n - 1) * H + h] = iso_w[b]; }
  for (int64_t r = 0; r < R; ++r) rg[r] = (int)(r / n);
  std::vector<float> hp(R * Ns), dyh(R * C);
  for (size_t i = 0; i < hp.size(); ++i) hp[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  for (size_t i = 0; i < dyh.size(); ++i) dyh[i] = (float)((i * 40503u) % 1000) / 1000.f - 0.5f;
  auto dev = [](const void* h, size_t bytes) { void* d; (void)hipMalloc(&d, bytes); (void)hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); return d; };
  int *d_rp = (int*)dev(rp.data(), rp.size() * 4), *d_col = (int*)dev(col.data(), col.size() * 4), *d_gp = (int*)dev(gp.data(), gp.size() * 4);
  int *d_rg = (int*)dev(rg.data(), rg.size() * 4), *d_ii = (int*)dev(iso_idx.data(), B * 4), *d_ip = (int*)dev(iso_ptr.data(), (B + 1) * 4);
  int* d_ep = (int*)dev(eperm.data(), nnz * 4);
  float *d_iw = (float*)dev(iso_w.data(), B * 4), *d_ir = (float*)dev(iso_row.data(), iso_row.size() * 4);
  float *d_hp = (float*)dev(hp.data(), hp.size() * 4), *d_dy = (float*)dev(dyh.data(), dyh.size() * 4);
  float *d_y, *d_stat, *d_dhp, *d_t1, *d_t2, *d_S, *d_du;
  (void)hipMalloc(&d_y, R * C * 4); (void)hipMalloc(&d_stat, R * H * 8); (void)hipMalloc(&d_dhp, R * Ns * 4);
  (void)hipMalloc(&d_t1, (size_t)nnz * H * 4); (void)hipMalloc(&d_t2, (size_t)nnz * H * 4); (void)hipMalloc(&d_S, R * H * 4);
  (void)hipMalloc(&d_du, (size_t)B * 8 * C * 4);
  hipStream_t s; (void)hipStreamCreate(&s);
  const int Co = mean ? Fh : C;
  auto fwd = [&] { return tsgnn_gat_attn_fwd_f32(d_hp, Ns, d_rp, d_col, d_rp, d_col, R, H, Fh, 0.2f, d_rg, nmax, d_ii, d_iw, d_ip, 1.f / nmax, mean, 1, 0.f, 0, nullptr, d_stat, d_y, Co, s); };
  auto bw = [&] {
    int rc = tsgnn_gat_attn_bwd_f32(d_hp, Ns, d_y, Co, d_dy, Co, d_rp, d_col, R, H, Fh, 0.2f, mean, 1, d_gp, B, d_ii, d_iw, d_ip, d_ir, H, 1.f / nmax, 0.f, 0, nullptr,
                                    d_stat, d_dhp, Ns, d_t1, d_t2, d_S, d_du, s);
    return rc;
  };
  int rc = fwd(); if (rc) { printf("fwd rc %d\n", rc); return 1; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 10; ++it) rc = bwd ? bw() : fwd();
  if (rc) { printf("rc %d\n", rc); return 1; }
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 100; ++it) bwd ? bw() : fwd();
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  printf("%s mean_heads=%d rows=%lld nnz=%d: %.2f us per call (back-to-back, incl. trace stores%s)\n", bwd ? "bwd" : "fwd", mean, (long long)R, nnz,
         ms * 1000 / 100, bwd ? "" : "; two launches: column statistics + attention");
  trace_report(t, 1024, 5);
  for (int w : {300 * 4, 300 * 4 + 1, 700 * 4, 1000 * 4 + 2})
    { printf("  wave %d:", w); for (int k = 0; k <= 5; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  return 0;
}
