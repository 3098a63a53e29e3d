// Mini-batch ingest for the training loop (graph_sampler.py:102-114 + default collate + the per-step .cuda() of
// train.py:114-119, which ship B x Nmax^2 floats per step).  Here a batch is a CAPACITY-PADDED packed CSR batch:
//
//   rows [0, n)            the real nodes of the B graphs, graph after graph        (n = graph_ptr[B] <= row_cap)
//   rows [n, row_cap)      padding: no edges, no graph (row_graph = B, a dummy graph whose results nobody reads)
//   rows [row_cap, +nmax)  the ghost-slot representatives
//
// so that every launch of the training step has the SAME shape for every batch (one hipGraph serves all steps) while the data
// that does vary (graph_ptr, slot_count, row maps, neighbour table, labels) lives in one device buffer that is refreshed with ONE
// host->device copy per batch.  tsgnn_host_collate_tu is plain host code (the reference does this in python/networkx/numpy per
// item, graph_sampler.py:24-46,102-114): it writes the batch straight in device layout into a (pinned) staging buffer.
// The node features of the TU "node-label" mode (train.py:227-231: one-hot of the node label) are expanded on the device from the
// uploaded labels (tsgnn_onehot_rows_f32): 4 bytes per node cross PCIe instead of 4 * F.
#include "common.h"
#include "../../include/tsgnn.h"
#include "ingest_rider.h"

#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <chrono>
#include <vector>

thread_local PullRider tsgnn_pull_rider_ = {nullptr, nullptr, 0, 0, 0, 0, 0};
thread_local ExpandRider tsgnn_expand_rider_ = {};

namespace {

inline int64_t align4(int64_t words) { return (words + 3) & ~(int64_t)3; }      // 16-byte segments

struct Layout {
  int64_t graph_ptr, slot_count, row_graph, row_slot, ell, tail_ptr, tail_col, node_label, label, total;
};
inline Layout make_layout(int B, int nmax, int64_t row_cap, int ell_w, int64_t tail_cap) {
  Layout L;
  int64_t o = 0;
  L.graph_ptr = o; o += align4(B + 2);
  L.slot_count = o; o += align4(nmax);
  L.row_graph = o; o += align4(row_cap);
  L.row_slot = o; o += align4(row_cap);
  L.ell = o; o += align4((row_cap + nmax) * ell_w);
  L.tail_ptr = o; o += align4(row_cap + nmax + 1);
  L.tail_col = o; o += align4(tail_cap > 0 ? tail_cap : 1);
  L.node_label = o; o += align4(row_cap);
  L.label = o; o += align4(2 * (int64_t)B);                                    // int64[B]
  L.total = o;
  return L;
}

__global__ __launch_bounds__(256) void onehot_rows_kernel(const int* __restrict__ label, int64_t n_rows, int64_t total_rows, int F, int ld4,
                                                          float* __restrict__ x, int64_t ldx) {
  // one float4 per thread; rows >= n_rows (padding, ghost representatives) are zero
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total_rows * ld4) return;
  const int64_t r = i / ld4;
  const int c = 4 * (int)(i - r * ld4);
  const int l = (r < n_rows) ? label[r] : -1;
  const bool ok = l >= 0 && l < F;
  const float4 v = make_float4((ok && l == c) ? 1.f : 0.f, (ok && l == c + 1) ? 1.f : 0.f, (ok && l == c + 2) ? 1.f : 0.f,
                               (ok && l == c + 3) ? 1.f : 0.f);
  *reinterpret_cast<float4*>(x + r * ldx + c) = v;
}


// ---- second generation (compact staging, pulled by the step's own hipGraph) -----------------------------------------------
// The host writes a COMPACT batch (CSR, not the expanded table): header {n, nnz, ntail, largest}, graph_ptr[B+2], slot_count[nmax],
// label int64[B], rowptr[row_cap+1], node_label[row_cap], tail_ptr[row_cap+1], col[edge_cap], tail_col[tail_cap] — about a third
// of the expanded layout.  Two kernels at the head of the step's hipGraph bring it in: `ingest_pull` copies the pinned host
// buffer over PCIe (coalesced 16-byte loads) into its device mirror — graph_ptr, slot_count, labels, node labels and the tail
// columns are used straight out of the mirror —; `ingest_expand` builds row maps, the fixed-width neighbour table and the one-hot
// feature rows from it.  No copy engine,
// no second stream, no cross-stream events: the copy-engine -> shader hand-over alone cost more than the whole transfer.
inline CLayout make_clayout(int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap) {
  CLayout L;
  int64_t o = 0;
  L.header = o; o += 8;                                 // {n, nnz, ntail, largest, sequence word (written by the caller), 3 spare}
  L.graph_ptr = o; o += align4(B + 2);
  L.slot_count = o; o += align4(nmax);
  L.label = o; o += align4(2 * (int64_t)B);
  L.rowptr = o; o += align4(row_cap + 1);
  L.node_label = o; o += align4(row_cap);
  L.tail_ptr = o; o += align4(row_cap + 1);
  L.col = o; o += align4(edge_cap);
  L.tail_col = o; o += align4(tail_cap > 0 ? tail_cap : 1);
  L.total = o;
  return L;
}

// flat copy of the whole staging buffer (capacity sized: ~0.34 MB for 32 DD graphs): no dependence on the batch's header, so
// every thread's 16-byte loads are in flight at once — one PCIe round trip, then bandwidth
__global__ __launch_bounds__(256) void ingest_pull_kernel(const int4* __restrict__ host, int4* __restrict__ mirror, int64_t n4) {
  const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsize = (int64_t)gridDim.x * 256;
  int4 v[4];
  int64_t i = gtid;
  for (; i + 3 * gsize < n4; i += 4 * gsize) {
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = host[i + u * gsize];
#pragma unroll
    for (int u = 0; u < 4; ++u) mirror[i + u * gsize] = v[u];
  }
  for (; i < n4; i += gsize) mirror[i] = host[i];
}

__global__ __launch_bounds__(256) void ingest_pull_rider_kernel(PullRider p) { pull_rider_body(p, blockIdx.x); }

__global__ __launch_bounds__(256) void ingest_expand_kernel(ExpandArgs a) {
  expand_row_lane(a, (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5), threadIdx.x & 31, a.mirror + a.L.graph_ptr);
}
__global__ __launch_bounds__(256) void ingest_expand_rider_kernel(ExpandRider e) { expand_rider_body(e, blockIdx.x, 256); }

}  // namespace

extern "C" {

/* word offsets of the COMPACT staging layout: off[0..8] = header{n, nnz, ntail, largest, sequence word, 3 spare}, graph_ptr[B+2], slot_count[nmax],
 * label (int64[B]), rowptr[row_cap+1], node_label[row_cap], tail_ptr[row_cap+1], col[edge_cap], tail_col[tail_cap]; off[9] = total */
int tsgnn_ingest_compact_layout(int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap, int64_t* off) {
  if (!off || B <= 0 || nmax <= 0 || row_cap <= 0 || edge_cap <= 0 || tail_cap < 0) return TSGNN_EINVAL;
  const CLayout L = make_clayout(B, nmax, row_cap, edge_cap, tail_cap);
  off[0] = L.header; off[1] = L.graph_ptr; off[2] = L.slot_count; off[3] = L.label; off[4] = L.rowptr; off[5] = L.node_label;
  off[6] = L.tail_ptr; off[7] = L.col; off[8] = L.tail_col; off[9] = L.total;
  return TSGNN_OK;
}

/* HOST function: the mini-batch ids[0..B) of a CSR-resident dataset in the compact layout (columns shifted to batch row ids;
 * tail_ptr / tail_col = the entries beyond the first ell_w of each row).  out[0..3] = rows, directed edges, tail entries,
 * largest graph (also the header).  TSGNN_EUNSUPPORTED: a graph over nmax nodes, rows over row_cap, edges over edge_cap, tail
 * over tail_cap. */
int tsgnn_host_collate_compact(const int64_t* ds_graph_ptr, const int64_t* ds_rowptr, const int64_t* ds_col, const int64_t* ds_node_label,
                               const int64_t* ds_graph_label, const int64_t* ids, int B, int nmax, int64_t row_cap, int64_t edge_cap,
                               int ell_w, int64_t tail_cap, int32_t* staging, int64_t* out) {
  if (!ds_graph_ptr || !ds_rowptr || !ds_col || !ds_graph_label || !ids || !staging || !out || B <= 0 || nmax <= 0 || row_cap <= 0 ||
      edge_cap <= 0 || (ell_w != 4 && ell_w != 8 && ell_w != 16) || tail_cap < 0)
    return TSGNN_EINVAL;
  const CLayout L = make_clayout(B, nmax, row_cap, edge_cap, tail_cap);
  int32_t* hd = staging + L.header;
  int32_t* gp = staging + L.graph_ptr;
  int32_t* sc = staging + L.slot_count;
  int64_t* lab = reinterpret_cast<int64_t*>(staging + L.label);
  int32_t* rp = staging + L.rowptr;
  int32_t* nl = staging + L.node_label;
  int32_t* tp = staging + L.tail_ptr;
  int32_t* col = staging + L.col;
  int32_t* tc = staging + L.tail_col;
  int64_t n = 0, nnz = 0, ntail = 0, largest = 0;
  std::memset(sc, 0, sizeof(int32_t) * (size_t)nmax);
  for (int b = 0; b < B; ++b) {
    const int64_t a = ds_graph_ptr[ids[b]], e = ds_graph_ptr[ids[b] + 1], sz = e - a;
    if (sz > nmax || sz < 0 || n + sz > row_cap) return TSGNN_EUNSUPPORTED;
    const int64_t e_lo = ds_rowptr[a], e_hi = ds_rowptr[e];
    if (nnz + (e_hi - e_lo) > edge_cap) return TSGNN_EUNSUPPORTED;
    gp[b] = (int32_t)n;
    if (sz > 0) sc[sz - 1] += 1;
    if (sz > largest) largest = sz;
    const int32_t shift = (int32_t)(n - a);
    const int64_t eshift = nnz - e_lo;
    for (int64_t r = 0; r < sz; ++r) {
      const int64_t e0 = ds_rowptr[a + r], d = ds_rowptr[a + r + 1] - e0;
      rp[n + r] = (int32_t)(e0 + eshift);
      nl[n + r] = ds_node_label ? (int32_t)ds_node_label[a + r] : 0;
      tp[n + r] = (int32_t)ntail;
      if (d > ell_w) {
        if (ntail + (d - ell_w) > tail_cap) return TSGNN_EUNSUPPORTED;
        for (int64_t k = ell_w; k < d; ++k) tc[ntail++] = (int32_t)ds_col[e0 + k] + shift;
      }
    }
    const int64_t* cp = ds_col + e_lo;
    int32_t* dst = col + nnz;
    for (int64_t k = 0; k < e_hi - e_lo; ++k) dst[k] = (int32_t)cp[k] + shift;     // one contiguous run per graph: vectorises
    nnz += e_hi - e_lo;
    lab[b] = ds_graph_label[ids[b]];
    n += sz;
  }
  rp[n] = (int32_t)nnz;
  tp[n] = (int32_t)ntail;
  gp[B] = (int32_t)n;
  gp[B + 1] = (int32_t)row_cap;
  {
    int32_t run = 0;
    for (int s = nmax - 1; s >= 0; --s) { run += sc[s]; sc[s] = run; }
  }
  hd[0] = (int32_t)n; hd[1] = (int32_t)nnz; hd[2] = (int32_t)ntail; hd[3] = (int32_t)largest;
  out[0] = n; out[1] = nnz; out[2] = ntail; out[3] = largest;
  return TSGNN_OK;
}

/* The two launches that bring a compact batch in (graph-capturable, `host` pinned and device-readable): pull host -> mirror
 * (+ graph_ptr, slot_count, label, node_label, tail_col: plain copies), then expand mirror -> row_graph, row_slot, ell
 * [(row_cap+nmax) x ell_w], tail_ptr[row_cap+nmax+1], one-hot x [(row_cap+nmax) x ldx] (F classes).  Sizes come from the batch's
 * own header, so one captured launch pair serves every batch of the slot. */
static int pull_expand_launch(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w,
                              int64_t tail_cap, int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F, float* x,
                              int64_t ldx, int64_t* host_ack, tsgnn_stream_t stream) {
  if (!host || !mirror || !row_graph || !row_slot || !ell || !tail_ptr || !x || B <= 0 || nmax <= 0 || row_cap <= 0 || edge_cap <= 0 ||
      tail_cap < 0 || F <= 0)
    return TSGNN_EINVAL;
  const int ld4 = (F + 3) / 4;
  if ((ell_w != 4 && ell_w != 8 && ell_w != 16) || ldx < 4 * ld4 || (ldx % 4)) return TSGNN_EUNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(host) | reinterpret_cast<uintptr_t>(mirror) | reinterpret_cast<uintptr_t>(ell) |
                       reinterpret_cast<uintptr_t>(x);
  if (al & 15) return TSGNN_EUNSUPPORTED;
  const CLayout L = make_clayout(B, nmax, row_cap, edge_cap, tail_cap);
  const int64_t n4 = L.total / 4;                          // every segment is a whole number of 16-byte units
  TSGNN_KNAME("ingest_pull_kernel + ingest_expand_kernel");
  unsigned pgrid = (unsigned)ceil_div64(n4, 256 * 2);
  if (pgrid > 512) pgrid = 512;
  ingest_pull_kernel<<<pgrid, 256, 0, stream>>>(reinterpret_cast<const int4*>(host), reinterpret_cast<int4*>(mirror), n4);
  ExpandArgs ea{mirror, L, B, nmax, ell_w, F, ld4, row_cap, row_graph, row_slot, ell, tail_ptr, x, ldx, host_ack, ell_slots, tail_slots};
  ingest_expand_kernel<<<(unsigned)ceil_div64(row_cap + nmax, 8), 256, 0, stream>>>(ea);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_ingest_pull_expand_f32(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w,
                                 int64_t tail_cap, int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F, float* x,
                                 int64_t ldx, tsgnn_stream_t stream) {
  return pull_expand_launch(host, mirror, B, nmax, row_cap, edge_cap, ell_w, tail_cap, row_graph, row_slot, ell, tail_ptr, ell_slots, tail_slots, F, x, ldx,
                            nullptr, stream);
}

/* the same pair; the expand launch also echoes the batch's sequence word (staging header word 4, written by whoever collated the
 * batch) to host_ack[0] (pinned host memory, system-scope store): once host_ack[0] == s, batch s has been pulled out of `host`,
 * which may then be refilled — the hand-shake with the collate workers (tsgnn_collate_pool_submit_ack) without a HIP event per
 * step. */
int tsgnn_ingest_pull_expand_ack_f32(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w,
                                     int64_t tail_cap, int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F,
                                     float* x, int64_t ldx, int64_t* host_ack, tsgnn_stream_t stream) {
  if (!host_ack) return TSGNN_EINVAL;
  return pull_expand_launch(host, mirror, B, nmax, row_cap, edge_cap, ell_w, tail_cap, row_graph, row_slot, ell, tail_ptr, ell_slots, tail_slots, F, x, ldx,
                            host_ack, stream);
}

/* the expand launch alone (mirror -> row maps, neighbour table, tail pointers, one-hot rows; echoes the sequence word to host_ack,
 * nullable): for a batch whose pull already happened — as passengers of the previous step (tsgnn_ingest_arm_pull_rider) or by
 * tsgnn_ingest_pull_f32. */
int tsgnn_ingest_expand_ack_f32(int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w, int64_t tail_cap,
                                int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F, float* x, int64_t ldx,
                                int64_t* host_ack, tsgnn_stream_t stream) {
  if (!mirror || !row_graph || !row_slot || !ell || !tail_ptr || !x || B <= 0 || nmax <= 0 || row_cap <= 0 || edge_cap <= 0 ||
      tail_cap < 0 || F <= 0)
    return TSGNN_EINVAL;
  const int ld4 = (F + 3) / 4;
  if ((ell_w != 4 && ell_w != 8 && ell_w != 16) || ldx < 4 * ld4 || (ldx % 4)) return TSGNN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(mirror) | reinterpret_cast<uintptr_t>(ell) | reinterpret_cast<uintptr_t>(x)) & 15) return TSGNN_EUNSUPPORTED;
  const CLayout L = make_clayout(B, nmax, row_cap, edge_cap, tail_cap);
  ExpandArgs ea{mirror, L, B, nmax, ell_w, F, ld4, row_cap, row_graph, row_slot, ell, tail_ptr, x, ldx, host_ack, ell_slots, tail_slots};
  TSGNN_KNAME("ingest_expand_kernel");
  ingest_expand_kernel<<<(unsigned)ceil_div64(row_cap + nmax, 8), 256, 0, stream>>>(ea);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

static int make_rider(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap,
                      PullRider* out) {
  if (!host || !mirror || B <= 0 || nmax <= 0 || row_cap <= 0 || edge_cap <= 0 || tail_cap < 0) return TSGNN_EINVAL;
  if ((reinterpret_cast<uintptr_t>(host) | reinterpret_cast<uintptr_t>(mirror)) & 15) return TSGNN_EUNSUPPORTED;
  const CLayout L = make_clayout(B, nmax, row_cap, edge_cap, tail_cap);
  const int64_t n4 = L.total / 4;
  unsigned blocks = (unsigned)ceil_div64(n4, 256 * 2);
  if (blocks > 512) blocks = 512;
  *out = PullRider{reinterpret_cast<const int4*>(host), reinterpret_cast<int4*>(mirror), (long long)n4, blocks, 0, 1, 0};
  return TSGNN_OK;
}

/* the pull launch alone (flat copy of the compact staging buffer `host`, pinned and device-readable, into `mirror`) */
int tsgnn_ingest_pull_f32(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap,
                          tsgnn_stream_t stream) {
  PullRider p;
  const int rc = make_rider(host, mirror, B, nmax, row_cap, edge_cap, tail_cap, &p);
  if (rc != TSGNN_OK) return rc;
  TSGNN_KNAME("ingest_pull_rider_kernel");
  ingest_pull_rider_kernel<<<p.blocks, 256, 0, stream>>>(p);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* Arm the same copy as PASSENGERS of this thread's next tsgnn_sage_layer_fwd_f32 / _ro_f32 launch (extra workgroups of that
 * launch: csrc/ingest_rider.h) — the pull of the NEXT mini-batch inside the CURRENT step.  One rider at a time (arming again
 * replaces it).  tsgnn_ingest_flush_pull_rider: launches an armed rider that no launch took, alone; no-op otherwise. */
int tsgnn_ingest_arm_pull_rider(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap) {
  return tsgnn_ingest_arm_pull_rider_parts(host, mirror, B, nmax, row_cap, edge_cap, tail_cap, 1, 0);
}
/* the same, dealt over `parts` carrier launches of the thread (tsgnn_gather_rowgemm_st_f32 — the first layer's product — and
 * tsgnn_sage_layer_fwd[_bn][_ro]_f32) in equal shares, after letting `skip` carriers pass without passengers */
int tsgnn_ingest_arm_pull_rider_parts(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap,
                                      int64_t tail_cap, int parts, int skip) {
  if (parts < 1 || parts > 4 || skip < 0 || skip > 8) return TSGNN_EINVAL;
  PullRider p;
  const int rc = make_rider(host, mirror, B, nmax, row_cap, edge_cap, tail_cap, &p);
  if (rc != TSGNN_OK) return rc;
  p.parts_left = parts;
  p.skip = skip;
  tsgnn_pull_rider_ = p;
  return TSGNN_OK;
}
/* ... and the EXPANSION of that batch (arguments of tsgnn_ingest_expand_ack_f32) as passengers of this thread's next
 * tsgnn_packed_head_fwd_f32 launch — a launch of a few workgroups that leaves most of the chip idle, later in the same step than the
 * launch that carries the pull. */
int tsgnn_ingest_arm_expand_rider(int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w, int64_t tail_cap,
                                  int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F, float* x, int64_t ldx,
                                  int64_t* host_ack) {
  if (!mirror || !row_graph || !row_slot || !ell || !tail_ptr || !x || B <= 0 || nmax <= 0 || row_cap <= 0 || edge_cap <= 0 ||
      tail_cap < 0 || F <= 0)
    return TSGNN_EINVAL;
  const int ld4 = (F + 3) / 4;
  if ((ell_w != 4 && ell_w != 8 && ell_w != 16) || ldx < 4 * ld4 || (ldx % 4)) return TSGNN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(mirror) | reinterpret_cast<uintptr_t>(ell) | reinterpret_cast<uintptr_t>(x)) & 15) return TSGNN_EUNSUPPORTED;
  const CLayout L = make_clayout(B, nmax, row_cap, edge_cap, tail_cap);
  ExpandRider e;
  e.ex = ExpandArgs{mirror, L, B, nmax, ell_w, F, ld4, row_cap, row_graph, row_slot, ell, tail_ptr, x, ldx, host_ack, ell_slots, tail_slots};
  e.rows = row_cap + nmax;
  const long long want = (e.rows + 15) / 16;                 // one 512-thread workgroup per 16 rows, at most 448 of them (looping)
  e.blocks = (unsigned)(want < 448 ? want : 448);
  tsgnn_expand_rider_ = e;
  return TSGNN_OK;
}
/* launches the armed riders that no launch took, each as a launch of its own (pull, then expansion); no-op otherwise */
int tsgnn_ingest_flush_pull_rider(tsgnn_stream_t stream) {
  PullRider p = tsgnn_pull_rider_;                          // whatever is left of the armed copy, in one launch
  disarm_pull_rider();
  if (p.blocks) {
    p.parts_left = 1;
    unsigned blocks = (unsigned)ceil_div64(p.n4 - p.lo, 256 * 2);
    p.blocks = blocks > 512 ? 512u : (blocks < 1 ? 1u : blocks);
    TSGNN_KNAME("ingest_pull_rider_kernel");
    ingest_pull_rider_kernel<<<p.blocks, 256, 0, stream>>>(p);
  }
  const ExpandRider e = take_expand_rider_for_flush();
  if (e.blocks) {
    TSGNN_KNAME("ingest_expand_rider_kernel");
    ingest_expand_rider_kernel<<<e.blocks, 256, 0, stream>>>(e);
  }
  if (p.blocks || e.blocks) TSGNN_CHECK_LAUNCH();          // (nothing armed: no runtime call at all)
  return TSGNN_OK;
}

/* drops the riders this thread armed without launching them (a forward that raised between arming and its carrier launches must not
 * leave passengers behind for an unrelated later launch) */
int tsgnn_ingest_disarm_riders(void) {
  disarm_pull_rider();
  (void)take_expand_rider_for_flush();
  return TSGNN_OK;
}

/* word offsets (4-byte words) of the segments of an ingest buffer: off[0..8] = graph_ptr[B+2], slot_count[nmax],
 * row_graph[row_cap], row_slot[row_cap], ell[(row_cap+nmax)*ell_w], tail_ptr[row_cap+nmax+1], tail_col[tail_cap],
 * node_label[row_cap], label (int64[B]); off[9] = total words.  Every segment starts on a 16-byte boundary. */
int tsgnn_ingest_layout(int B, int nmax, int64_t row_cap, int ell_w, int64_t tail_cap, int64_t* off) {
  if (!off || B <= 0 || nmax <= 0 || row_cap <= 0 || (ell_w != 4 && ell_w != 8 && ell_w != 16) || tail_cap < 0) return TSGNN_EINVAL;
  const Layout L = make_layout(B, nmax, row_cap, ell_w, tail_cap);
  off[0] = L.graph_ptr; off[1] = L.slot_count; off[2] = L.row_graph; off[3] = L.row_slot; off[4] = L.ell; off[5] = L.tail_ptr;
  off[6] = L.tail_col; off[7] = L.node_label; off[8] = L.label; off[9] = L.total;
  return TSGNN_OK;
}

/* HOST function (no GPU work): collate the graphs ids[0..B) of a TU-style dataset held as one CSR over re-labelled nodes
 * (tu_data.TUDataset: ds_graph_ptr[G+1], ds_rowptr[N+1], ds_col[nnz] in dataset node ids, ds_node_label[N] nullable,
 * ds_graph_label[G]) into `staging` in the device layout of tsgnn_ingest_layout.  Columns are shifted to batch row ids;
 * a row's first ell_w neighbours go to the table (filled from the left, -1 beyond), the rest to the CSR tail.
 * out[0] = real rows n, out[1] = directed edges, out[2] = tail entries, out[3] = largest graph.
 * Errors: a graph with more than nmax nodes, n > row_cap or the tail over tail_cap -> TSGNN_EUNSUPPORTED (nothing usable
 * is left in staging). */
int tsgnn_host_collate_tu(const int64_t* ds_graph_ptr, const int64_t* ds_rowptr, const int64_t* ds_col, const int64_t* ds_node_label,
                          const int64_t* ds_graph_label, const int64_t* ids, int B, int nmax, int64_t row_cap, int ell_w,
                          int64_t tail_cap, int32_t* staging, int64_t* out) {
  if (!ds_graph_ptr || !ds_rowptr || !ds_col || !ds_graph_label || !ids || !staging || !out || B <= 0 || nmax <= 0 || row_cap <= 0 ||
      (ell_w != 4 && ell_w != 8 && ell_w != 16) || tail_cap < 0)
    return TSGNN_EINVAL;
  const Layout L = make_layout(B, nmax, row_cap, ell_w, tail_cap);
  int32_t* gp = staging + L.graph_ptr;
  int32_t* sc = staging + L.slot_count;
  int32_t* rg = staging + L.row_graph;
  int32_t* rs = staging + L.row_slot;
  int32_t* ell = staging + L.ell;
  int32_t* tp = staging + L.tail_ptr;
  int32_t* tc = staging + L.tail_col;
  int32_t* nl = staging + L.node_label;
  int64_t* lab = reinterpret_cast<int64_t*>(staging + L.label);
  int64_t n = 0, nnz = 0, ntail = 0, largest = 0;
  // the staging buffer remembers how many rows its previous batch filled (word gp[B+1] holds row_cap only for a buffer this
  // function has written before): rows beyond the larger of the two counts still hold their padding values
  const bool seen = gp[B + 1] == (int32_t)row_cap && gp[B] >= 0 && gp[B] <= row_cap;
  const int64_t prev_n = seen ? gp[B] : -1;
  std::memset(sc, 0, sizeof(int32_t) * (size_t)nmax);
  for (int b = 0; b < B; ++b) {
    const int64_t a = ds_graph_ptr[ids[b]], e = ds_graph_ptr[ids[b] + 1], sz = e - a;
    if (sz > nmax || sz < 0 || n + sz > row_cap) { gp[B + 1] = -1; return TSGNN_EUNSUPPORTED; }
    gp[b] = (int32_t)n;
    if (sz > 0) sc[sz - 1] += 1;                        // histogram of sizes; turned into "graphs with slot s" below
    if (sz > largest) largest = sz;
    const int32_t shift = (int32_t)(n - a);
    int32_t* erow = ell + n * ell_w;
    int32_t* rgp = rg + n; int32_t* rsp = rs + n; int32_t* nlp = nl + n; int32_t* tpp = tp + n;
    const int64_t* rp = ds_rowptr + a;
    int64_t e0 = rp[0];
    for (int64_t r = 0; r < sz; ++r, erow += ell_w) {
      rgp[r] = b;
      rsp[r] = (int32_t)r;
      nlp[r] = ds_node_label ? (int32_t)ds_node_label[a + r] : 0;
      const int64_t e1 = rp[r + 1], d = e1 - e0;
      const int dt = (int)(d < ell_w ? d : ell_w);
      const int64_t* cp = ds_col + e0;
      for (int k = 0; k < ell_w; ++k) erow[k] = -1;     // (fixed trip count: vector stores)
      for (int k = 0; k < dt; ++k) erow[k] = (int32_t)cp[k] + shift;
      tpp[r] = (int32_t)ntail;
      if (d > ell_w) {
        if (ntail + (d - ell_w) > tail_cap) { gp[B + 1] = -1; return TSGNN_EUNSUPPORTED; }
        for (int64_t k = ell_w; k < d; ++k) tc[ntail++] = (int32_t)cp[k] + shift;
      }
      nnz += d;
      e0 = e1;
    }
    lab[b] = ds_graph_label[ids[b]];
    n += sz;
  }
  gp[B] = (int32_t)n;                                    // end of the real rows
  gp[B + 1] = (int32_t)row_cap;                          // the dummy graph B = the padding rows
  // slot_count[s] = graphs with more than s nodes: suffix sums of the size histogram
  {
    int32_t run = 0;
    for (int s = nmax - 1; s >= 0; --s) { run += sc[s]; sc[s] = run; }
  }
  // padding: rows this buffer's previous batch used as real rows (all of [n, row_cap + nmax) the first time)
  const int64_t pad_hi = prev_n < 0 ? row_cap + nmax : (prev_n > n ? prev_n : n);
  const int64_t pad_hi_rows = pad_hi < row_cap ? pad_hi : row_cap;
  for (int64_t row = n; row < pad_hi_rows; ++row) { rg[row] = B; rs[row] = 0; nl[row] = 0; }
  for (int64_t row = n; row < pad_hi; ++row) {
    int32_t* erow = ell + row * ell_w;
    for (int k = 0; k < ell_w; ++k) erow[k] = -1;
  }
  // tail pointers behind the real rows all equal the tail length (it changes from batch to batch)
  for (int64_t row = n; row <= row_cap + nmax; ++row) tp[row] = (int32_t)ntail;
  out[0] = n; out[1] = nnz; out[2] = ntail; out[3] = largest;
  return TSGNN_OK;
}

/* x[r, :] = one-hot(label[r]) for r < n_rows, 0 for n_rows <= r < total_rows (padding and ghost rows); x rows are 16-byte
 * aligned with ldx >= 4*ceil(F/4) (the pad columns are written as zeros): the "node-label" features of train.py:227-231. */
int tsgnn_onehot_rows_f32(const int* label, int64_t n_rows, int64_t total_rows, int F, float* x, int64_t ldx, tsgnn_stream_t stream) {
  if (!label || !x || n_rows < 0 || total_rows < n_rows || F <= 0) return TSGNN_EINVAL;
  const int ld4 = (F + 3) / 4;
  if (ldx < 4 * ld4 || (ldx % 4) || (reinterpret_cast<uintptr_t>(x) & 15)) return TSGNN_EUNSUPPORTED;
  if (total_rows == 0) return TSGNN_OK;
  TSGNN_KNAME("onehot_rows_kernel");
  onehot_rows_kernel<<<(unsigned)ceil_div64(total_rows * ld4, 256), 256, 0, stream>>>(label, n_rows, total_rows, F, ld4, x, ldx);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}


/* ---- collate workers: native threads that run tsgnn_host_collate_tu for the batches AHEAD of the step being enqueued, so the
 * thread that drives the GPU only uploads and replays (python threads would serialise on the interpreter lock).  A job may
 * name a HIP event to wait for first (the previous upload out of the same staging buffer must have completed). */
struct tsgnn_collate_pool {
  struct Job {
    const int64_t *gp, *rp, *col, *nl, *gl, *ids;
    int B, nmax, ell_w; int64_t row_cap, tail_cap, edge_cap; int32_t* staging; int64_t* out; hipEvent_t after;
    int64_t ticket; int rc; bool done;
    const int64_t* ack = nullptr; int64_t ack_target = 0; int32_t seq = 0;   // alternative to `after`: wait until *ack >= ack_target,
                                                                             // stamp the collated batch with `seq`
  };
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<Job*> queue;
  std::vector<Job*> all;
  std::vector<std::thread> threads;
  int64_t next_ticket = 1;
  bool stop = false;
};

static void collate_worker(tsgnn_collate_pool* p) {
  for (;;) {
    tsgnn_collate_pool::Job* j;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty(); });
      if (p->stop && p->queue.empty()) return;
      j = p->queue.front();
      p->queue.pop_front();
    }
    if (j->after) (void)hipEventSynchronize(j->after);
    bool acked = true;
    if (j->ack) {                                          // (bounded: a device that never runs the step must not hang the worker)
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (__atomic_load_n(j->ack, __ATOMIC_ACQUIRE) < j->ack_target) {
        if (++spins > 64u) {                               // not there yet: leave the core to the enqueueing thread for a while
          std::this_thread::sleep_for(std::chrono::microseconds(20));
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) { acked = false; break; }
        }
      }
    }
    const int rc = !acked ? TSGNN_ELAUNCH : j->edge_cap > 0
                       ? tsgnn_host_collate_compact(j->gp, j->rp, j->col, j->nl, j->gl, j->ids, j->B, j->nmax, j->row_cap, j->edge_cap,
                                                    j->ell_w, j->tail_cap, j->staging, j->out)
                       : tsgnn_host_collate_tu(j->gp, j->rp, j->col, j->nl, j->gl, j->ids, j->B, j->nmax, j->row_cap, j->ell_w,
                                               j->tail_cap, j->staging, j->out);
    if (j->ack && rc == TSGNN_OK && j->edge_cap > 0)
      j->staging[make_clayout(j->B, j->nmax, j->row_cap, j->edge_cap, j->tail_cap).header + 4] = j->seq;
    {
      std::lock_guard<std::mutex> lk(p->mu);
      j->rc = rc;
      j->done = true;
    }
    p->cv_done.notify_all();
  }
}

int tsgnn_collate_pool_create(int nthreads, tsgnn_collate_pool** pool) {
  if (!pool || nthreads <= 0 || nthreads > 64) return TSGNN_EINVAL;
  tsgnn_collate_pool* p = new tsgnn_collate_pool();
  for (int t = 0; t < nthreads; ++t) p->threads.emplace_back(collate_worker, p);
  *pool = p;
  return TSGNN_OK;
}

/* queue one tsgnn_host_collate_tu call (edge_cap = 0) or tsgnn_host_collate_compact call (edge_cap > 0) with these arguments
 * (`ids` and `out` must stay valid until the job was waited for);
 * after_event (nullable hipEvent_t): the worker synchronises with it before it writes `staging`.  *ticket identifies the job. */
int tsgnn_collate_pool_submit(tsgnn_collate_pool* pool, const int64_t* ds_graph_ptr, const int64_t* ds_rowptr, const int64_t* ds_col,
                              const int64_t* ds_node_label, const int64_t* ds_graph_label, const int64_t* ids, int B, int nmax,
                              int64_t row_cap, int64_t edge_cap, int ell_w, int64_t tail_cap, int32_t* staging, int64_t* out,
                              void* after_event, int64_t* ticket) {
  if (!pool || !ticket) return TSGNN_EINVAL;
  auto* j = new tsgnn_collate_pool::Job{ds_graph_ptr, ds_rowptr, ds_col, ds_node_label, ds_graph_label, ids, B, nmax, ell_w, row_cap,
                                        tail_cap, edge_cap, staging, out, reinterpret_cast<hipEvent_t>(after_event), 0, 0, false};
  {
    std::lock_guard<std::mutex> lk(pool->mu);
    j->ticket = pool->next_ticket++;
    pool->queue.push_back(j);
    pool->all.push_back(j);
    *ticket = j->ticket;
  }
  pool->cv_work.notify_one();
  return TSGNN_OK;
}

/* tsgnn_collate_pool_submit (compact layout only) whose worker waits for host_ack[0] >= ack_target (the word
 * tsgnn_ingest_pull_expand_ack_f32 stores: the sequence word of the batch this buffer held before) instead of an event, and stamps
 * the new batch with `seq` (header word 4); gives up after 20 s (the job then reports TSGNN_ELAUNCH) */
int tsgnn_collate_pool_submit_ack(tsgnn_collate_pool* pool, const int64_t* ds_graph_ptr, const int64_t* ds_rowptr, const int64_t* ds_col,
                                  const int64_t* ds_node_label, const int64_t* ds_graph_label, const int64_t* ids, int B, int nmax,
                                  int64_t row_cap, int64_t edge_cap, int ell_w, int64_t tail_cap, int32_t* staging, int64_t* out,
                                  const int64_t* host_ack, int64_t ack_target, int seq, int64_t* ticket) {
  if (!pool || !ticket || !host_ack || edge_cap <= 0) return TSGNN_EINVAL;
  auto* j = new tsgnn_collate_pool::Job{ds_graph_ptr, ds_rowptr, ds_col, ds_node_label, ds_graph_label, ids, B, nmax, ell_w, row_cap,
                                        tail_cap, edge_cap, staging, out, nullptr, 0, 0, false};
  j->ack = host_ack;
  j->ack_target = ack_target;
  j->seq = seq;
  {
    std::lock_guard<std::mutex> lk(pool->mu);
    j->ticket = pool->next_ticket++;
    pool->queue.push_back(j);
    pool->all.push_back(j);
    *ticket = j->ticket;
  }
  pool->cv_work.notify_one();
  return TSGNN_OK;
}

/* block until the job is done; returns the collate's own status */
int tsgnn_collate_pool_wait(tsgnn_collate_pool* pool, int64_t ticket) {
  if (!pool) return TSGNN_EINVAL;
  std::unique_lock<std::mutex> lk(pool->mu);
  tsgnn_collate_pool::Job* j = nullptr;
  size_t at = 0;
  for (size_t i = 0; i < pool->all.size(); ++i)
    if (pool->all[i]->ticket == ticket) { j = pool->all[i]; at = i; break; }
  if (!j) return TSGNN_EINVAL;
  pool->cv_done.wait(lk, [&] { return j->done; });
  const int rc = j->rc;
  pool->all.erase(pool->all.begin() + (long)at);
  delete j;
  return rc;
}

int tsgnn_collate_pool_destroy(tsgnn_collate_pool* pool) {
  if (!pool) return TSGNN_EINVAL;
  {
    std::lock_guard<std::mutex> lk(pool->mu);
    pool->stop = true;
  }
  pool->cv_work.notify_all();
  for (auto& t : pool->threads) t.join();
  for (auto* j : pool->all) delete j;
  delete pool;
  return TSGNN_OK;
}

/* staging -> device (ONE asynchronous copy of `words` 4-byte words; `host` should be pinned) followed by the one-hot feature
 * expansion of tsgnn_onehot_rows_f32, both on `stream`: the whole upload of a batch in one call. */
int tsgnn_ingest_upload_f32(int32_t* dev, const int32_t* host, int64_t words, const int* node_label_dev, int64_t n_rows, int64_t total_rows,
                            int F, float* x, int64_t ldx, tsgnn_stream_t stream) {
  if (!dev || !host || words <= 0) return TSGNN_EINVAL;
  if (hipMemcpyAsync(dev, host, sizeof(int32_t) * (size_t)words, hipMemcpyHostToDevice, stream) != hipSuccess) return TSGNN_ELAUNCH;
  return tsgnn_onehot_rows_f32(node_label_dev, n_rows, total_rows, F, x, ldx, stream);
}

}  // extern "C"
