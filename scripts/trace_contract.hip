// Developer harness: per-wave phase timeline of contract_rows_bwd_kernel (DiffPool level-1 contraction backward), built as
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTSGNN_TRACE scripts/trace_contract.hip -o scripts/_build/trace_contract
#include "../two-stage-gnn_amd/csrc/contract.hip"
#include "trace_util.h"
thread_local char tsgnn_kname_[160];
#include <cstdio>
#include <vector>

int main() {
  const int B = 16, n = 277, K = 64, F = 192, nghost = 512;
  const int64_t R = (int64_t)B * n, RT = R + nghost;
  std::vector<int> srp, sg;
  for (int b = 0; b < B; ++b) for (int r = b * n; r < (b + 1) * n; r += 32) { srp.push_back(r); sg.push_back(b); }
  const int nslab = (int)srp.size();
  srp.push_back((int)R);
  auto dmal = [](size_t bytes) { void* d; (void)hipMalloc(&d, bytes); (void)hipMemset(d, 0, bytes); return d; };
  float *S = (float*)dmal(RT * K * 4), *Z = (float*)dmal(RT * F * 4), *AS = (float*)dmal(RT * K * 4);
  float *dxo = (float*)dmal((size_t)B * K * F * 4), *dao = (float*)dmal((size_t)B * K * K * 4);
  float *dZ = (float*)dmal(RT * F * 4), *dS = (float*)dmal(RT * K * 4), *dAS = (float*)dmal(RT * K * 4);
  int *d_srp = (int*)dmal(srp.size() * 4), *d_sg = (int*)dmal(sg.size() * 4);
  (void)hipMemcpy(d_srp, srp.data(), srp.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_sg, sg.data(), sg.size() * 4, hipMemcpyHostToDevice);
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&] { return tsgnn_contract_rows_bwd_f32(S, K, Z, F, AS, K, dxo, dao, d_srp, d_sg, nslab, K, F, dZ, F, dS, K, dAS, K, R, RT, s); };
  int rc = 0;
  for (int it = 0; it < 10; ++it) rc = run();
  if (rc) { printf("rc %d\n", rc); return 1; }
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < 100; ++it) run();
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(4096 * 16);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
  printf("nslab=%d: %.2f us per launch (back-to-back, incl. trace stores)\n", nslab, ms * 1000 / 100);
  trace_report(t, nslab, 11);
  for (int w : {0, 1, 2, 3, 200, 201, 202, 203})
    { printf("  wave %d:", w); for (int k = 0; k <= 11; ++k) printf(" %lld", t[w * 16 + k] - t[w * 16]); printf("\n"); }
  return 0;
}
