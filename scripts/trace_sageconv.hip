// Developer harness: per-wave phase timeline of sage_conv_kernel (s_memtime stamps), built as
//   hipcc --offload-arch=gfx950 -O3 -DTSGNN_TRACE -DTSGNN_TRACE_WPB=8 scripts/trace_sageconv.hip -o gpurun_out/trace_sageconv
// stamps: 0 start, 6 ids arrived, 7 self rows arrived, 1 barrier #1, 8 W registers arrived, 2 group 1's chain done, 9 neighbour rows
// arrived, 3 barrier #2, 4 group 0's chain done, 5 end
#include "../two-stage-gnn_amd/csrc/sageconv.hip"
thread_local char tsgnn_kname_[160] = "";
#include <cstdio>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
  const int64_t R = argc > 1 ? atoll(argv[1]) : 8151;
  const int K = argc > 2 ? atoi(argv[2]) : 128, N = 128;
  const int ldw = argc > 3 ? atoi(argv[3]) : 128;
  float *a, *b, *b2, *bias, *c, *z, *invd;
  hipMalloc(&a, R * 128 * 4); hipMalloc(&b, 128 * 128 * 4); hipMalloc(&b2, 128 * 128 * 4); hipMalloc(&bias, N * 4); hipMalloc(&c, R * N * 4);
  hipMalloc(&z, R * 128 * 4); hipMalloc(&invd, R * 4);
  std::vector<float> h(R * 128);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(a, h.data(), R * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), 128 * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(b2, h.data() + 999, 128 * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), N * 4, hipMemcpyHostToDevice);
  hipMemcpy(invd, h.data(), R * 4, hipMemcpyHostToDevice);
  std::vector<int> hell(R * 16, -1);
  for (int64_t r = 0; r < R; ++r) {
    const int deg = 2 + (int)((r * 2654435761u >> 7) % 7);
    for (int k = 0; k < deg; ++k) {
      long long j = r + (long long)(((r * 40503u + k * 9973u) >> 3) % 600) - 300;
      hell[r * 16 + k] = (int)std::min<long long>(std::max<long long>(j, 0), R - 1);
    }
  }
  int* ell; hipMalloc(&ell, R * 16 * 4); hipMemcpy(ell, hell.data(), R * 16 * 4, hipMemcpyHostToDevice);
  float* pk; hipMalloc(&pk, 4 * 16384 * 4);
  hipStream_t s; hipStreamCreate(&s);
  {
    int64_t desc[1 + 4 * 6] = {4, (int64_t)b, ldw, K, N, 0, (int64_t)pk, (int64_t)b2, ldw, K, N, 0, (int64_t)(pk + 16384),
                               (int64_t)b, 128, N, K, 1, (int64_t)(pk + 2 * 16384), (int64_t)b2, 128, N, K, 1, (int64_t)(pk + 3 * 16384)};
    tsgnn_sage_conv_pack_f32(desc, s);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kn = 0; kn < 2; ++kn) {
    auto run = [&]() {
      if (!kn) tsgnn_sage_conv_f32(ell, 16, nullptr, nullptr, a, 128, a, 128, invd, pk, pk + 16384, bias, c, N, z, 128, nullptr, R, K, N, 1, 0,
                                   nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, s);
      else tsgnn_sage_conv_f32(ell, 16, nullptr, nullptr, a, 128, a, 128, nullptr, pk + 2 * 16384, pk + 3 * 16384, nullptr, c, N, nullptr, 0, nullptr, R, N, K, 0, 0,
                               nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, s);
    };
    for (int it = 0; it < 20; ++it) run();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int it = 0; it < 200; ++it) run();
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t(4096 * 16);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8);
    const int nb = (int)std::min<int64_t>((R + 31) / 32, 4096 / 8);
    printf("w_kn=%d K=%d ldw=%d: %.2f us per launch (back-to-back, incl. trace stores)\n", kn, K, ldw, ms * 1000 / 200);
    for (int grp = 0; grp < 2; ++grp) {
      double avg[16] = {0}; int n = 0;
      for (int bidx = 0; bidx < nb; ++bidx)
        for (int w = 4 * grp; w < 4 * grp + 4; ++w, ++n)
          for (int k = 0; k < 10; ++k) avg[k] += (double)(t[(bidx * 8 + w) * 16 + k] - t[(bidx * 8 + w) * 16]);
      printf("  group %d mean ticks: ids[6] %.0f self[7] %.0f bar1[1] %.0f W[8] %.0f chain1[2] %.0f rows[9] %.0f bar2[3] %.0f chain0[4] %.0f end[5] %.0f\n", grp,
             avg[6] / n, avg[7] / n, avg[1] / n, avg[8] / n, avg[2] / n, avg[9] / n, avg[3] / n, avg[4] / n, avg[5] / n);
    }
    long long s0 = 1ll << 62, emax = 0;
    for (int w = 0; w < nb * 8; ++w) { s0 = std::min(s0, t[w * 16 + 14]); emax = std::max(emax, t[w * 16 + 15]); }
    printf("  kernel span by the wall clock: %.2f us\n", (emax - s0) * 0.01);
  }
  return 0;
}
