// The pull of the NEXT mini-batch (flat copy of its pinned host staging buffer into its device mirror, csrc/ingest.hip) as
// extra workgroups of a kernel of the CURRENT step: a PCIe round trip costs ~8 us on this stack, and as a launch of its own it
// is 10 us of every step; as passengers of a 12 us row-panel launch (which leaves a second workgroup slot free on every CU) the
// copy is over before the launch is.  tsgnn_ingest_arm_pull_rider arms it (thread-local), the next tsgnn_sage_layer_fwd*_f32 call of
// the thread takes it along; tsgnn_ingest_flush_pull_rider launches it alone if nothing did.
// The EXPANSION of the pulled batch (mirror -> row maps, neighbour table, tail pointers, one-hot feature rows, + the echo of its
// sequence word) rides later in the same step, in tsgnn_packed_head_fwd_f32's launch: a few workgroups of latency-bound work that
// leave most of the chip idle.  (As passengers of the next layer-product launch it lengthened that launch by what a launch of its
// own cost: 0.1825 vs 0.1819 ms.)  tsgnn_ingest_arm_expand_rider arms it.
#pragma once
#include "common.h"

// lo / n4: the range of 16-byte words this launch copies; parts_left > 1: the armed copy is dealt over several carrier launches (the
// staging buffer of a DD batch is ~340 KB = ~15 us of PCIe, longer than any one launch of the step: carried whole by a 14.5 us launch it
// ended 4.4 us after it; half in the first layer's product and half in the second ends inside both)
struct PullRider { const int4* host; int4* mirror; long long n4; unsigned blocks; long long lo; int parts_left; int skip; };   // skip: carrier launches to let pass first
extern thread_local PullRider tsgnn_pull_rider_;          // armed while blocks > 0

// the next part of the armed copy, sized for carrier workgroups of `threads` threads; the rider stays armed until its last part is taken.
static inline PullRider take_pull_rider(int threads = 256) {
  PullRider& a = tsgnn_pull_rider_;
  PullRider r = a;
  if (a.blocks == 0) return r;
  if (a.skip > 0) { --a.skip; r.blocks = 0; return r; }   // this carrier goes without passengers
  const int parts = a.parts_left > 1 ? a.parts_left : 1;
  const long long left = a.n4 - a.lo;
  const long long take = parts > 1 ? ((left / parts + 3) & ~3ll) : left;
  r.n4 = a.lo + (take < left ? take : left);
  long long blocks = (r.n4 - r.lo + 2ll * threads - 1) / (2ll * threads);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  r.blocks = (unsigned)blocks;
  a.lo = r.n4;
  a.parts_left = parts - 1;
  if (a.parts_left <= 0 || a.lo >= a.n4) a.blocks = 0;
  return r;
}
static inline void disarm_pull_rider() { tsgnn_pull_rider_.blocks = 0; }

// workgroup b of p.blocks: no dependence on the batch's header, so every thread's 16-byte loads are in flight at once
__device__ __forceinline__ void pull_rider_body(const PullRider& p, unsigned b) {
  const long long gtid = (long long)b * blockDim.x + threadIdx.x, gsize = (long long)p.blocks * blockDim.x;
  int4 v[4];
  long long i = p.lo + gtid;
  for (; i + 3 * gsize < p.n4; i += 4 * gsize) {
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p.host[i + u * gsize];
#pragma unroll
    for (int u = 0; u < 4; ++u) p.mirror[i + u * gsize] = v[u];
  }
  for (; i < p.n4; i += gsize) p.mirror[i] = p.host[i];
}

// ---- the expansion
struct CLayout {
  int64_t header, graph_ptr, slot_count, label, rowptr, node_label, tail_ptr, col, tail_col, total;
};

struct ExpandArgs {
  const int32_t* mirror; CLayout L;
  int B, nmax, ell_w, F, ld4; int64_t row_cap;
  int32_t* row_graph; int32_t* row_slot; int32_t* ell; int32_t* tail_ptr; float* x; int64_t ldx;
  int64_t* host_ack;                     // nullable (pinned host memory): receives the batch's sequence word once it is pulled
  // nullable: the neighbour table / the CSR tail once more with the neighbour's SLOT beside its row (entry = slot << 20 | row), the
  // operand of the layers that form their input's slot batch-norm on the fly (tsgnn_sage_layer_fwd_bn_f32).  A neighbour lives in
  // the row's own graph, so its slot is its row minus the graph's first row.
  int32_t* ell_slots; int32_t* tail_slots;
};
// gp: the batch's graph pointers — the mirror's copy, or the workgroup's copy of it in LDS (expand_rider_body): the search is five
// DEPENDENT reads for 32 graphs, and out of global memory they were the longest chain of a row's expansion
__device__ __forceinline__ int expand_graph_of(const ExpandArgs& a, int64_t r, const int32_t* gp) {      // the graph whose row range holds r (r < n)
  int lo = 0, hi = a.B;                                  // gp[lo] <= r < gp[hi]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (gp[mid] <= r) lo = mid; else hi = mid; }
  return lo;
}

// 32 lanes per row, 8 rows per block.  Lane q of a row: q < ell_w/4 writes four entries of the row's neighbour table, q == ell_w/4
// the row maps and the tail pointer, the lanes after that (looping when a row has more than 32 - ell_w/4 - 1 float4) the one-hot
// feature row.
__device__ __forceinline__ void expand_row_lane(const ExpandArgs& a, int64_t r, int q, const int32_t* gp) {
  const int64_t total_rows = a.row_cap + a.nmax;
  if (a.host_ack && r == 0 && q == 0) {
    // the pull launch ahead of this one has finished reading the staging buffer: echo the batch's sequence word to the host,
    // which may refill the buffer once it sees it (the collate workers wait on this word — no event between the step's launches)
    __hip_atomic_store(a.host_ack, (int64_t)a.mirror[a.L.header + 4], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (r >= total_rows) return;
  const int64_t n = a.mirror[a.L.header];
  const int32_t ntail = a.mirror[a.L.header + 2];
  const int EQ = a.ell_w / 4;
  const int l = (r < n) ? a.mirror[a.L.node_label + r] : -1;
  if (q < EQ) {
    int4 v = make_int4(-1, -1, -1, -1);
    if (r < n) {
      const int32_t* rowptr = a.mirror + a.L.rowptr;
      const int e0 = rowptr[r], d = rowptr[r + 1] - e0;
      const int32_t* col = a.mirror + a.L.col + e0;
      const int k = 4 * q;
      if (k < d) v.x = col[k];
      if (k + 1 < d) v.y = col[k + 1];
      if (k + 2 < d) v.z = col[k + 2];
      if (k + 3 < d) v.w = col[k + 3];
    }
    *reinterpret_cast<int4*>(a.ell + r * a.ell_w + 4 * q) = v;
    if (a.ell_slots) {
      int4 w = make_int4(-1, -1, -1, -1);
      if (r < n) {
        const int g0 = gp[expand_graph_of(a, r, gp)];
        if (v.x >= 0) w.x = ((v.x - g0) << 20) | v.x;
        if (v.y >= 0) w.y = ((v.y - g0) << 20) | v.y;
        if (v.z >= 0) w.z = ((v.z - g0) << 20) | v.z;
        if (v.w >= 0) w.w = ((v.w - g0) << 20) | v.w;
      }
      *reinterpret_cast<int4*>(a.ell_slots + r * a.ell_w + 4 * q) = w;
    }
  } else if (q == EQ) {
    int g0 = 0;
    if (r < a.row_cap) {
      int g = a.B, slot = -1;                               // (padding rows of the capacity: no graph, no slot)
      if (r < n) {
        g = expand_graph_of(a, r, gp);
        g0 = gp[g];
        slot = (int)(r - g0);
      }
      a.row_graph[r] = g;
      a.row_slot[r] = slot;
    }
    a.tail_ptr[r] = r < n ? a.mirror[a.L.tail_ptr + r] : ntail;
    if (r == total_rows - 1) a.tail_ptr[total_rows] = ntail;
    if (a.tail_slots && r < n) {                            // this row's share of the CSR tail (rows with more than ell_w neighbours: few)
      const int e1 = (r + 1 < n) ? a.mirror[a.L.tail_ptr + r + 1] : ntail;
      for (int e = a.mirror[a.L.tail_ptr + r]; e < e1; ++e) {
        const int j = a.mirror[a.L.tail_col + e];
        a.tail_slots[e] = ((j - g0) << 20) | j;
      }
    }
  } else {
    const bool ok = l >= 0 && l < a.F;                      // (no dynamic register indexing: that would go through scratch)
    for (int c4 = q - EQ - 1; c4 < a.ld4; c4 += 32 - EQ - 1) {
      const int c = 4 * c4;
      const float4 v = make_float4((ok && l == c) ? 1.f : 0.f, (ok && l == c + 1) ? 1.f : 0.f, (ok && l == c + 2) ? 1.f : 0.f,
                                   (ok && l == c + 3) ? 1.f : 0.f);
      *reinterpret_cast<float4*>(a.x + r * a.ldx + c) = v;
    }
  }
}

struct ExpandRider { ExpandArgs ex; long long rows; unsigned blocks; };   // armed while blocks > 0
extern thread_local ExpandRider tsgnn_expand_rider_;
static inline ExpandRider take_expand_rider() {
  ExpandRider r = tsgnn_expand_rider_;
  // the expansion reads what the pull wrote: while a share of the pull still waits for a carrier (a placement with more shares than the
  // model has carriers in front of the head) the expansion does not ride either — tsgnn_ingest_flush_pull_rider launches both, in order
  if (tsgnn_pull_rider_.blocks != 0) { r.blocks = 0; return r; }
  tsgnn_expand_rider_.blocks = 0;
  return r;
}
static inline ExpandRider take_expand_rider_for_flush() {
  ExpandRider r = tsgnn_expand_rider_;
  tsgnn_expand_rider_.blocks = 0;
  return r;
}
// workgroup b of e.blocks, `nthreads` threads (a multiple of 32): 32 lanes a row, looping over the rows
constexpr int EXPAND_GP_LDS = 256;          // graph pointers a workgroup keeps in LDS (larger batches search the mirror)
__device__ __forceinline__ void expand_rider_body(const ExpandRider& e, unsigned b, int nthreads) {
  __shared__ int32_t gp_s[EXPAND_GP_LDS + 1];
  const int32_t* gp = e.ex.mirror + e.ex.L.graph_ptr;
  if (e.ex.B <= EXPAND_GP_LDS) {             // (uniform; every thread of a rider workgroup is here)
    for (int i = threadIdx.x; i <= e.ex.B; i += nthreads) gp_s[i] = gp[i];
    __syncthreads();
    gp = gp_s;
  }
  const int rpb = nthreads >> 5;
  for (long long r0 = (long long)b * rpb; r0 < e.rows; r0 += (long long)e.blocks * rpb)
    expand_row_lane(e.ex, r0 + (threadIdx.x >> 5), threadIdx.x & 31, gp);
}
