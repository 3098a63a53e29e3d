// shared reporting for the scripts/trace_*.hip harnesses: per-wave phase means and per-XCD dispatch skew
#pragma once
#include <cstdio>
#include <vector>
#include <algorithm>

// t: [nblocks*4][16] stamps; blocks are dealt round-robin to the 8 XCDs, each XCD has its own counter
static void trace_report(const std::vector<long long>& t, int nblocks, int last) {
#ifndef TSGNN_TRACE_WPB
#define TSGNN_TRACE_WPB 4
#endif
  const int nw = std::min(nblocks, 4096 / TSGNN_TRACE_WPB) * TSGNN_TRACE_WPB;
  double avg[16] = {0};
  for (int w = 0; w < nw; ++w)
    for (int k = 0; k <= last; ++k) avg[k] += (double)(t[w * 16 + k] - t[w * 16]);
  printf("  mean ticks since wave start: ");
  for (int k = 0; k <= last; ++k) printf("[%d]%.0f ", k, avg[k] / nw);
  printf("\n");
  // device-wide wall clock (slots 14 = wave start, 15 = wave end, 10 ns units): dispatch skew and kernel span
  std::vector<long long> st, en;
  for (int w = 0; w < nw; ++w) { st.push_back(t[w * 16 + 14]); en.push_back(t[w * 16 + 15]); }
  const long long s0 = *std::min_element(st.begin(), st.end());
  std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
  printf("  wall clock (us after the first wave start): starts p50=%.2f p90=%.2f max=%.2f; ends p10=%.2f p50=%.2f max=%.2f\n",
         (st[nw / 2] - s0) * 0.01, (st[nw * 9 / 10] - s0) * 0.01, (st.back() - s0) * 0.01, (en[nw / 10] - s0) * 0.01,
         (en[nw / 2] - s0) * 0.01, (en.back() - s0) * 0.01);
}
