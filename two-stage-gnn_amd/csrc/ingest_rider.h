// The pull of the NEXT mini-batch (flat copy of its pinned host staging buffer into its device mirror, csrc/ingest.hip) as
// extra workgroups of a kernel of the CURRENT step: a PCIe round trip costs ~8 us on this stack, and as a launch of its own it
// is 10 us of every step; as passengers of a 12 us row-panel launch (which leaves a second workgroup slot free on every CU) the
// copy is over before the launch is.  tsgnn_ingest_arm_pull_rider arms it (thread-local), the next tsgnn_sage_layer_fwd*_f32 call of
// the thread takes it along; tsgnn_ingest_flush_pull_rider launches it alone if nothing did.
// (The expansion of the pulled batch was tried as passengers of the following layer launch as well: the step took as long as
// with the expansion as its first launch — 0.1825 vs 0.1819 ms — so it stayed a launch.)
#pragma once
#include "common.h"

struct PullRider { const int4* host; int4* mirror; long long n4; unsigned blocks; };
extern thread_local PullRider tsgnn_pull_rider_;          // armed while blocks > 0

static inline PullRider take_pull_rider() {
  PullRider r = tsgnn_pull_rider_;
  tsgnn_pull_rider_.blocks = 0;
  return r;
}

// workgroup b of p.blocks: no dependence on the batch's header, so every thread's 16-byte loads are in flight at once
__device__ __forceinline__ void pull_rider_body(const PullRider& p, unsigned b) {
  const long long gtid = (long long)b * 256 + threadIdx.x, gsize = (long long)p.blocks * 256;
  int4 v[4];
  long long i = gtid;
  for (; i + 3 * gsize < p.n4; i += 4 * gsize) {
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p.host[i + u * gsize];
#pragma unroll
    for (int u = 0; u < 4; ++u) p.mirror[i + u * gsize] = v[u];
  }
  for (; i < p.n4; i += gsize) p.mirror[i] = p.host[i];
}
