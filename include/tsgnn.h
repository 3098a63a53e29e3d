/* libtsgnn_hip — C ABI of the MI355X (gfx950) message-passing / aggregation hot path.
 *
 * The reference (manhtuando97/two-stage-gnn) is pure Python and has no FFI; its boundary for this
 * path is the nn.Module surface of Code/sage+gat+diffpool/encoders.py, encoders_GAT.py and
 * Code/sag/layers.py.  This header is what a ctypes binding on the reference side binds (see
 * INTEGRATION.md); every entry point names the reference lines whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch types; caller owns every buffer;
 *   - fp32 values, int32 indices unless stated (int64 only where PyG's edge_index is consumed);
 *   - returns 0 on success, <0 on error (TSGNN_E*); never throws, never allocates, never
 *     synchronises, launches on `stream` (graph-capturable);
 *   - re-entrant, no global state: one process per GPU is safe.
 */
#ifndef TSGNN_H
#define TSGNN_H
#include <stdint.h>

/* == hipStream_t (HIP declares it as `struct ihipStream_t*`); spelled out so that plain C callers
 * (gcc, cgo, ctypes generators) need no HIP headers. */
typedef struct ihipStream_t* tsgnn_stream_t;

#ifdef __cplusplus
extern "C" {
#endif

#define TSGNN_ABI_VERSION 1
int tsgnn_abi_version(void);
const char* tsgnn_strerror(int code);
/* Diagnostic: name (template arguments included) of the device kernel that the calling thread's most recent entry-point call
 * dispatched, "" if that entry point is not annotated.  Thread-local, read-only for the caller; used by bench.py to label
 * its per-kernel roofline table from the actual dispatch. */
const char* tsgnn_last_kernel(void);

/* ---------------------------------------------------------------- graph ingest (graph_build.hip) */

/* ints of scratch needed by tsgnn_exclusive_scan_i32 / tsgnn_csr_transpose for n items */
int tsgnn_scan_workspace_ints(int64_t n, int64_t* ws_ints);
/* out[0..n] = exclusive prefix sums of in[0..n-1] (out[n] = total) */
int tsgnn_exclusive_scan_i32(const int* in, int64_t n, int* out, int* ws, tsgnn_stream_t stream);

/* row -> graph id / slot-within-graph for rows laid out graph after graph (graph_ptr[B+1]).
 * Replaces the implicit (b, n) indexing of the padded [B,Nmax,*] tensors (graph_sampler.py:102-114)
 * and GcnEncoderGraph.construct_mask's python loop (encoders.py:121-132). */
int tsgnn_row_maps(const int* graph_ptr, int B, int64_t n_rows, int* row_graph, int* row_slot,
                   tsgnn_stream_t stream);

/* dense padded adjacency adj[B,Nmax,Nmax] (train.py:114) -> CSR over rows r = graph_ptr[b]+n,
 * columns ascending, values kept (normalised adjacencies stay weighted).
 * pass 1: row_cnt[r] = nnz of adj[b,n,0:size_b];  (scan row_cnt -> rowptr);  pass 2: fill. */
int tsgnn_dense_adj_count(const float* adj, int B, int nmax, const int* graph_ptr, const int* row_graph,
                          int64_t n_rows, int* row_cnt, tsgnn_stream_t stream);
int tsgnn_dense_adj_fill(const float* adj, int B, int nmax, const int* graph_ptr, const int* row_graph,
                         int64_t n_rows, const int* rowptr, int* col, float* val, tsgnn_stream_t stream);

/* COO edge list (PyG edge_index rows, int64; Code/sag/network.py:31) -> CSR grouped by `key`
 * (key = edge_index[1] = target for message passing).  Stable: entries of a row keep edge order;
 * eid[p] = original edge id.  bad_flag != 0 afterwards if any key was out of range. */
int tsgnn_coo_count(const int64_t* key, int64_t E, int64_t n_rows, int* cnt, int* bad_flag,
                    tsgnn_stream_t stream);
int tsgnn_coo_fill(const int64_t* key, const int64_t* other, int64_t E, int64_t n_rows, const int* rowptr,
                   int* cursor, int* col, int* eid, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- mini-batch ingest (ingest.hip)
 *
 * A capacity-padded packed batch: rows [0, n) real nodes (n = graph_ptr[B] <= row_cap), rows [n, row_cap) padding (no
 * edges, row_graph = B: a dummy graph), rows [row_cap, row_cap + nmax) the ghost-slot representatives.  Every launch of the
 * training step then has the same shape for every batch (one hipGraph serves all steps); what varies lives in ONE device
 * buffer refreshed by ONE host->device copy.  Replaces GraphSampler.__getitem__ + default collate + the per-step .cuda() of
 * adj[B,Nmax,Nmax] (graph_sampler.py:102-114, train.py:114-119). */

/* word offsets (4-byte words) of the segments of an ingest buffer: off[0..8] = graph_ptr[B+2], slot_count[nmax],
 * row_graph[row_cap], row_slot[row_cap], ell[(row_cap+nmax)*ell_w], tail_ptr[row_cap+nmax+1], tail_col[tail_cap],
 * node_label[row_cap], label (int64[B]); off[9] = total words.  Every segment starts on a 16-byte boundary. */
int tsgnn_ingest_layout(int B, int nmax, int64_t row_cap, int ell_w, int64_t tail_cap, int64_t* off);
/* HOST function (no GPU work, no stream): collate graphs ids[0..B) of a dataset held as one CSR over re-labelled nodes
 * (ds_graph_ptr[G+1], ds_rowptr[N+1], ds_col[nnz] dataset node ids, ds_node_label[N] nullable, ds_graph_label[G]) into
 * `staging` (host memory, tsgnn_ingest_layout words) in device layout.  out[0..3] = real rows, directed edges, tail entries,
 * largest graph.  TSGNN_EUNSUPPORTED: a graph over nmax nodes, rows over row_cap, tail over tail_cap. */
int tsgnn_host_collate_tu(const int64_t* ds_graph_ptr, const int64_t* ds_rowptr, const int64_t* ds_col, const int64_t* ds_node_label,
                          const int64_t* ds_graph_label, const int64_t* ids, int B, int nmax, int64_t row_cap, int ell_w,
                          int64_t tail_cap, int32_t* staging, int64_t* out);
/* x[r, :] = one-hot(label[r]) (r < n_rows), 0 for n_rows <= r < total_rows: the "node-label" input features of
 * train.py:227-231 built on the device from 4 bytes per node. */
int tsgnn_onehot_rows_f32(const int* label, int64_t n_rows, int64_t total_rows, int F, float* x, int64_t ldx, tsgnn_stream_t stream);

/* staging -> device (ONE asynchronous copy of `words` 4-byte words; pinned host memory) + tsgnn_onehot_rows_f32, on `stream` */
int tsgnn_ingest_upload_f32(int32_t* dev, const int32_t* host, int64_t words, const int* node_label_dev, int64_t n_rows, int64_t total_rows,
                            int F, float* x, int64_t ldx, tsgnn_stream_t stream);
/* COMPACT staging (about a third of the expanded layout: the CSR, not the table) and the two launches that bring it in from
 * pinned host memory at the head of the step's own hipGraph — no copy engine, no second stream, no cross-stream events.
 * Layout: off[0..8] = header{n, nnz, ntail, largest, sequence word, 3 spare}, graph_ptr[B+2], slot_count[nmax], label (int64[B]), rowptr[row_cap+1],
 * node_label[row_cap], tail_ptr[row_cap+1], col[edge_cap], tail_col[tail_cap]; off[9] = total words. */
int tsgnn_ingest_compact_layout(int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap, int64_t* off);
/* HOST function: tsgnn_host_collate_tu's batch in the compact layout (TSGNN_EUNSUPPORTED also when edges exceed edge_cap) */
int tsgnn_host_collate_compact(const int64_t* ds_graph_ptr, const int64_t* ds_rowptr, const int64_t* ds_col, const int64_t* ds_node_label,
                               const int64_t* ds_graph_label, const int64_t* ids, int B, int nmax, int64_t row_cap, int64_t edge_cap,
                               int ell_w, int64_t tail_cap, int32_t* staging, int64_t* out);
/* pull (flat copy of the whole staging buffer into its device mirror: graph_ptr, slot_count, label, node_label, tail_col are used
 * straight out of the mirror at the layout's offsets) and expand (mirror -> row maps, neighbour table, tail pointers, one-hot
 * feature rows); the expansion reads the sizes from the batch's header, so the captured pair serves every batch.
 * ell_slots [(row_cap+nmax)*ell_w] / tail_slots [tail_cap] (nullable, here and in the entry points below): the neighbour table and
 * the CSR tail once more with the neighbour's SLOT beside its row (entry = slot << 20 | row; a neighbour lives in the row's own
 * graph, so its slot is its row minus the graph's first row) — the operand of tsgnn_sage_layer_fwd_bn_f32.  row_slot of a
 * padding row (>= n) is -1. */
int tsgnn_ingest_pull_expand_f32(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w,
                                 int64_t tail_cap, int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F, float* x,
                                 int64_t ldx, tsgnn_stream_t stream);
/* The same pair; the expand launch also echoes the batch's sequence word (header word 4 of the compact layout, written by whoever
 * collated the batch) to host_ack[0] (PINNED HOST memory, system-scope store): once host_ack[0] == s, batch s has been pulled out
 * of `host`, which may be refilled.  The hand-shake with the collate workers (tsgnn_collate_pool_submit_ack) without a HIP event
 * per step (graph_sampler.py:102-114 / train.py:110-119: the per-step batch hand-over). */
int tsgnn_ingest_pull_expand_ack_f32(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w,
                                     int64_t tail_cap, int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F,
                                     float* x, int64_t ldx, int64_t* host_ack, tsgnn_stream_t stream);
/* The two halves alone, and the pull as PASSENGERS of a launch of the previous step: tsgnn_ingest_arm_pull_rider arms the flat copy
 * host -> mirror on the calling thread; the thread's next tsgnn_sage_layer_fwd_f32 / _ro_f32 launch carries it as extra workgroups
 * (the pull of the NEXT mini-batch inside the CURRENT step: a launch of its own is ~10 us of PCIe latency per step);
 * tsgnn_ingest_flush_pull_rider launches an armed rider that no launch took (no-op otherwise).  tsgnn_ingest_expand_ack_f32: the
 * expansion of an already pulled batch (+ the sequence-word echo, host_ack nullable). */
int tsgnn_ingest_pull_f32(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap,
                          tsgnn_stream_t stream);
int tsgnn_ingest_expand_ack_f32(int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w, int64_t tail_cap,
                                int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F, float* x, int64_t ldx,
                                int64_t* host_ack, tsgnn_stream_t stream);
int tsgnn_ingest_arm_pull_rider(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap);
/* the same copy dealt over `parts` (1..4) carrier launches of the thread in equal shares, after `skip` carriers that go without — carriers: tsgnn_gather_rowgemm_st_f32 (its
 * kernels) and tsgnn_sage_layer_fwd[_bn]_f32; the passengers are extra workgroups at the END of the carrier's grid.
 * A DD batch's staging buffer is ~15 us of PCIe, longer than any launch of the step (graph_sampler.py:102-114 / train.py:110-119) */
int tsgnn_ingest_arm_pull_rider_parts(const int32_t* host, int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int64_t tail_cap,
                                      int parts, int skip);
/* the EXPANSION of that batch (arguments of tsgnn_ingest_expand_ack_f32) as passengers of the thread's next
 * tsgnn_packed_head_fwd_f32 launch (a few latency-bound workgroups: most of the chip is idle under it), later in the same step than
 * the launch that carries the pull.  tsgnn_ingest_flush_pull_rider launches whichever of the two riders no launch took. */
int tsgnn_ingest_arm_expand_rider(int32_t* mirror, int B, int nmax, int64_t row_cap, int64_t edge_cap, int ell_w, int64_t tail_cap,
                                  int32_t* row_graph, int32_t* row_slot, int32_t* ell, int32_t* tail_ptr, int32_t* ell_slots, int32_t* tail_slots, int F, float* x, int64_t ldx,
                                  int64_t* host_ack);
int tsgnn_ingest_flush_pull_rider(tsgnn_stream_t stream);
/* drops the riders this thread armed WITHOUT launching them: a forward that raised between arming and its carrier launches must not
 * leave passengers behind for an unrelated later launch (ingest.IngestPipeline calls it on its error path) */
int tsgnn_ingest_disarm_riders(void);
/* Collate workers: native threads that run the host collate for the batches ahead of the step being enqueued.  submit: the
 * arguments of tsgnn_host_collate_tu (edge_cap = 0) or tsgnn_host_collate_compact (edge_cap > 0) (`ids`, `out` must stay valid
 * until waited for) + after_event (nullable hipEvent_t: the worker synchronises with it before writing `staging`); wait: blocks,
 * returns the collate's status. */
typedef struct tsgnn_collate_pool tsgnn_collate_pool;
int tsgnn_collate_pool_create(int nthreads, tsgnn_collate_pool** pool);
int tsgnn_collate_pool_submit(tsgnn_collate_pool* pool, const int64_t* ds_graph_ptr, const int64_t* ds_rowptr, const int64_t* ds_col,
                              const int64_t* ds_node_label, const int64_t* ds_graph_label, const int64_t* ids, int B, int nmax,
                              int64_t row_cap, int64_t edge_cap, int ell_w, int64_t tail_cap, int32_t* staging, int64_t* out,
                              void* after_event, int64_t* ticket);
/* submit (compact layout: edge_cap > 0) whose worker waits until host_ack[0] >= ack_target before it writes `staging` (gives up
 * after 20 s: the job then reports a launch error) and stamps the collated batch with `seq` (header word 4) */
int tsgnn_collate_pool_submit_ack(tsgnn_collate_pool* pool, const int64_t* ds_graph_ptr, const int64_t* ds_rowptr, const int64_t* ds_col,
                                  const int64_t* ds_node_label, const int64_t* ds_graph_label, const int64_t* ids, int B, int nmax,
                                  int64_t row_cap, int64_t edge_cap, int ell_w, int64_t tail_cap, int32_t* staging, int64_t* out,
                                  const int64_t* host_ack, int64_t ack_target, int seq, int64_t* ticket);
int tsgnn_collate_pool_wait(tsgnn_collate_pool* pool, int64_t ticket);
int tsgnn_collate_pool_destroy(tsgnn_collate_pool* pool);

/* CSR transpose (A^T for dX = A^T dY); rows of the result sorted by column; src_e[p] = source entry */
int tsgnn_csr_transpose(const int* rowptr, const int* col, const float* val, int64_t n_rows, int64_t n_cols,
                        int64_t nnz, int* rowptr_t, int* col_t, float* val_t, int* src_e, int* cnt_ws,
                        int* scan_ws, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- aggregation (aggregate.hip) */

/* y[i,:] (+)= sum_e val[e] * act(x[col[e],:]) + (self_w[i] + self_scalar) * act(x[i,:])
 *   val == NULL: unit weights;  act = relu if relu_in else identity.
 * Replaces torch.matmul(adj, x) [+ x] of GraphConv.forward (encoders.py:33-35) and the scatter-add
 * propagate of PyG GCNConv / SAGEConv / GraphConv (Code/sag/network.py:34, layers.py:18).
 * Backward of the same op = this call on the transposed CSR. */
int tsgnn_csr_spmm_f32(const int* rowptr, const int* col, const float* val, const float* self_w,
                       const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows, int feat,
                       float self_scalar, int relu_in, int accumulate, tsgnn_stream_t stream);

/* Fixed-width (ELL) view of a CSR for low-degree graphs: ell[r][k] = k-th neighbour of row r or -1 (W in {4,8,16});
 * tail_cnt[r] = entries beyond W (kept in a CSR tail: scan tail_cnt -> tail_ptr, then tsgnn_csr_tail_fill).
 * tsgnn_ell_spmm_f32: y = A_ell.x (+ self_scalar*x) — same sums, same order as tsgnn_csr_spmm_f32 on the first W
 * entries of every row, without the rowptr->col dependent hop; finish overflow rows with
 * tsgnn_csr_spmm_f32(tail_ptr, tail_col, ..., accumulate=1).  feat % 4 == 0, feat <= 256. */
int tsgnn_csr_to_ell(const int* rowptr, const int* col, int64_t n_rows, int W, int* ell, int* tail_cnt, tsgnn_stream_t stream);
int tsgnn_csr_tail_fill(const int* rowptr, const int* col, const int* tail_ptr, int64_t n_rows, int W, int* tail_col,
                        tsgnn_stream_t stream);
int tsgnn_ell_spmm_f32(const int* ell, int W, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows, int feat,
                       float self_scalar, tsgnn_stream_t stream);

/* PyG gcn_norm (add remaining self loops of weight self_fill, symmetric D^-1/2 (A+I) D^-1/2):
 * dinv[i], val_out[e] = dinv[i]*val[e]*dinv[col[e]], self_w[i] = dinv[i]^2*self_fill (0 if the row
 * already has a self loop).  Call site: GCNConv in Code/sag/network.py:19-23, layers.py:12. */
int tsgnn_gcn_norm_f32(const int* rowptr, const int* col, const float* val, int64_t n_rows, float self_fill,
                       float* dinv, float* val_out, float* self_w, tsgnn_stream_t stream);

/* Gradient of the weighted aggregation y[i] = self_w[i] x[i] + sum_e val[e] x[col[e]] with respect to its weights (PyG
 * GCNConv(edge_weight=) with weights that require grad, torch_geometric gcn_norm; the reference passes none, Code/sag/layers.py:18):
 * dval[e] = dy[i] . x[col[e]], dself[i] = dy[i] . x[i] (dself nullable).  feat <= 1024. */
int tsgnn_sddmm_rows_f32(const int* rowptr, const int* col, const float* dy, int64_t lddy, const float* x, int64_t ldx,
                         int64_t n_rows, int feat, float* dval, float* dself, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- dense transform (gemm.hip, linear.hip) */

/* C[z] (+)= alpha * op(A[z]) . op(B[z]); element (m,k) of op(A) at A + m*sam + k*sak (transposes are
 * strides).  batch: strided (stride_*) or ragged via seg_ptr (ragged = 1: K dimension covers rows
 * [seg_ptr[z], seg_ptr[z+1]) of both operands; ragged = 2: the M dimension of A and C; max_seg =
 * longest segment).  fp32 MFMA.  Replaces torch.matmul(y, W) (encoders.py:36), h = x[0] @ w
 * (encoders_GAT.py:32) and DiffPool's S^T Z / S^T A S (encoders.py:374-375) incl. their backward. */
int tsgnn_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                   int64_t scm, int64_t scn, int M, int N, int K, int batch, int64_t stride_a, int64_t stride_b,
                   int64_t stride_c, const int* seg_ptr, int ragged, int max_seg, float alpha, int accumulate,
                   tsgnn_stream_t stream);
/* split-K plan + product for weight gradients dW = Z^T dU (K = number of graph rows); partial slabs are
 * summed in a fixed order (bitwise reproducible, no float atomics). C dense row-major [M,N]. */
int tsgnn_gemm_splitk_plan(int M, int N, int K, int* ksplit, int64_t* ws_floats);
int tsgnn_gemm_splitk_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                          float* C, int M, int N, int K, int ksplit, float* ws, int accumulate,
                          tsgnn_stream_t stream);
/* Weight + bias gradient of the GraphConv transform in one pass over the rows (backward of encoders.py:36-38):
 * dw[K_in,N] = z[:, :K_in]^T . du,  db[N] = colsum(du) (nullable).  Row slabs -> fixed-order reduce (reproducible).
 * K_in, N <= 128, 16-byte rows; tsgnn_linear_wgrad_plan returns nslab = 0 for unsupported shapes.
 * bias_only_rows: rows [rows, rows + bias_only_rows) of du add to db but not to dw (the ghost rows of a packed batch:
 * their z is identically zero, so they are left out of the product). */
int tsgnn_linear_wgrad_plan(int64_t rows, int K_in, int N, int64_t ldz, int64_t lddu, int* nslab, int64_t* rows_per_slab,
                            int64_t* ws_floats);
int tsgnn_linear_wgrad_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                           int64_t rows_per_slab, int64_t bias_only_rows, float* ws, float* dw, float* db, tsgnn_stream_t stream);
/* dw == NULL above leaves the slabs in ws; this reduces up to four such slab sets (all layers of one backward pass) in ONE
 * launch.  Unused sets: ws == NULL.  normparts (nullable): block k of the launch (sum over sets of ceil((K+1)*N/64) blocks)
 * stores the sum of squares of the gradient entries it wrote; step_state (nullable): step_state[0] += 1 (both feed
 * tsgnn_adam_from_partials_f32). */
int tsgnn_wgrad_reduce_multi_f32(const float* ws0, int nslab0, int K0, int N0, float* dw0, float* db0, const float* ws1, int nslab1,
                                 int K1, int N1, float* dw1, float* db1, const float* ws2, int nslab2, int K2, int N2, float* dw2,
                                 float* db2, const float* ws3, int nslab3, int K3, int N3, float* dw3, float* db3,
                                 float* normparts, float* step_state, tsgnn_stream_t stream);
/* dW[K_in, N] = z[:, :K_in]^T . du for K_in, N <= 512 (the GAT projections: 92 x 264, 256 x 264): the slab kernel on 128 x 128
 * output blocks, every block and slab in ONE launch, then one fixed-order reduction (no float atomics).  Plan first
 * (nslab = 0: unsupported shape); ws holds ws_floats floats. */
int tsgnn_wgrad_blocks_plan(int64_t rows, int K_in, int N, int64_t ldz, int64_t lddu, int* nslab, int64_t* rows_per_slab,
                            int64_t* ws_floats);
int tsgnn_wgrad_blocks_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                           int64_t rows_per_slab, float* ws, float* dw, int64_t lddw, tsgnn_stream_t stream);
/* The same product written in torch.nn.Linear's layout, dw_oi[N][K_in] (row stride lddw >= K_in), with db[N] = colsum(du)
 * (nullable) from the same pass: the gradients of y = x W^T + b (DiffPool's assign_pred, encoders.py:369; Code/sag/network.py:
 * 48-53) arrive as contiguous tensors of the parameters' own layout. */
int tsgnn_wgrad_blocks_oi_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                              int64_t rows_per_slab, float* ws, float* dw_oi, int64_t lddw, float* db, tsgnn_stream_t stream);
/* Ragged batched out[b][K,N] = s[rows_b,:K]^T . x[rows_b,:N] — DiffPool's S^T Z and S^T (A S) (encoders.py:374-375) over
 * the row ranges of the graphs: graph b owns slabs [seg_slab_ptr[b], seg_slab_ptr[b+1]); slab t covers rows
 * [slab_row_ptr[t], slab_row_ptr[t+1]).  ws >= nslab*(K+1)*N floats.  ceil(K/32)*ceil(N/32) <= 16. */
int tsgnn_ragged_tn_f32(const float* s_mat, int64_t lds_, const float* x, int64_t ldx, int K, int N, const int* slab_row_ptr,
                        int nslab, const int* seg_slab_ptr, int nseg, float* ws, float* out, tsgnn_stream_t stream);
/* The same products for SHORT segments and K <= 128, without slabs and for two X operands at once (DiffPool's first contraction:
 * S^T Z and S^T (A S), encoders.py:374-375): out0[b] = S[rows_b]^T X0[rows_b] ([B, K, N0]) and, when x1 is given,
 * out1[b] = S[rows_b]^T X1[rows_b] ([B, K, N1]) in ONE launch — workgroup (32 x 32 output tile, segment) walks the whole segment, its eight
 * waves split the rows and add their accumulators in wave order (csrc/ragged.hip).  rows_b = [graph_ptr[b], graph_ptr[b+1]). */
int tsgnn_ragged_tn_direct_supported(int K, int64_t max_rows);
int tsgnn_ragged_tn_direct_f32(const float* s_mat, int64_t lds_, int K, const int* graph_ptr, int B, const float* x0, int64_t ldx0,
                               int N0, float* out0, const float* x1, int64_t ldx1, int N1, float* out1, tsgnn_stream_t stream);
/* the same + the max readout of x0 over each segment's node slots (encoders.py:353; trap T5) from the values the product's
 * workgroups hold anyway: ro_out [B, N0] (leading dimension ro_ldo), ro_arg [B, N0] = winning row.  ghost_zero != 0: a segment
 * shorter than nmax also has ZERO rows behind the real ones (row n_real + slot: the masked embeddings of a packed batch); the first
 * of them takes part, exactly as in tsgnn_readout_max_fwd_f32.  ghost_zero = 0: the segment's rows are all of its slots. */
int tsgnn_ragged_tn_direct_ro_f32(const float* s_mat, int64_t lds_, int K, const int* graph_ptr, int B, const float* x0, int64_t ldx0,
                                  int N0, float* out0, const float* x1, int64_t ldx1, int N1, float* out1, float* ro_out,
                                  int64_t ro_ldo, int* ro_arg, int nmax, int64_t n_real, int ghost_zero, tsgnn_stream_t stream);
/* Weight + bias gradient of a layer with a narrow input (K_in <= 4, e.g. the one-column constant feature of IMDB-B,
 * network.py:34 with num_features = 1): dwb[(K_in + 1), N], rows 0..K_in-1 = z[:, :K_in]^T du, row K_in = column sums of du,
 * from one pass over du.  ws: *ws_floats of tsgnn_wgrad_narrow_plan(rows, K_in, N, &ws_floats). */
int tsgnn_wgrad_narrow_plan(int64_t rows, int K_in, int N, int64_t* ws_floats);
int tsgnn_wgrad_narrow_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, float* ws,
                           float* dwb, tsgnn_stream_t stream);
/* out[f] (+)= sum_r x[r,f] (bias gradients); ws >= ceil(rows/128)*F floats */
int tsgnn_colsum_f32(const float* x, int64_t ld, int64_t rows, int F, float* out, float* ws, int accumulate,
                     tsgnn_stream_t stream);

/* v = normalize(z.W + bias, p=2, dim=1, eps=1e-12) (normalize=0: plain affine); rinv[r] = 1/max(|u_r|,eps).
 * GraphConv.forward lines encoders.py:36-40 as one kernel.  N <= 256. */
int tsgnn_linear_l2norm_f32(const float* z, int64_t ldz, const float* w, int64_t ldw, const float* bias, float* v,
                            int64_t ldv, float* rinv, int64_t rows, int K, int N, int normalize,
                            tsgnn_stream_t stream);
/* Row-panel variant of the same product (16-byte global loads, register prefetch, ds_read_b128 A fragments):
 * c = a[rows,K] . b  with b = B[K,N] (trans_b = 0) or b = W[N,K] used transposed (trans_b = 1: dZ = dU . W^T),
 * optional + bias and row L2 normalise.  tsgnn_rowgemm_supported() tells whether the operands qualify
 * (16-byte aligned rows, N <= 256); callers fall back to tsgnn_linear_l2norm_f32 / tsgnn_gemm_f32 otherwise.
 * fill_rows: rows [rows, rows + fill_rows) of c (and rinv) receive the epilogue of an all-zero input row, i.e. the
 * (normalised) bias, without going through the product: the ghost rows of a packed batch aggregate nothing
 * (ghost rows), so their GraphConv output is that constant.  Needs N % 4 == 0 and 16-byte aligned c rows. */
int tsgnn_rowgemm_supported(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c, int64_t ldc, int K, int N,
                            int trans_b);
int tsgnn_rowgemm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int trans_b, const float* bias, float* c,
                      int64_t ldc, float* rinv, int64_t rows, int K, int N, int normalize, int64_t fill_rows,
                      tsgnn_stream_t stream);
/* workgroups the row panels of a `rows`-row fused layer launch take on the current device: ceil(rows / 32) 32-row panels, or — when that
 * is a few more than the device has compute units — one full panel per unit and the remaining rows as 16-row units (a unit with two
 * full panels decided the launch: 1.3-1.4 x).  For callers that size a co-resident role of the same launch (encoders.py:33-40 backward:
 * the weight-gradient slab blocks beside the input-gradient panels). */
int tsgnn_panel_blocks(int64_t rows);
/* on = 0: the fused layer launches issued by this process keep plain 32-row panels until on = 1 again (process-wide, not per thread: a
 * backward's launches come from autograd's thread).  For capacity-padded batches — ingest.IngestPipeline captures its steps under it: the
 * rows beyond one panel per unit are mostly padding there.  Not for concurrent use by two trainers of one process. */
int tsgnn_panel_split_hint(int on);
/* Aggregation fused into the product (GraphConv.forward lines encoders.py:33-40 in one launch; and its input gradient
 * dX = (A dU) W^T for a symmetric A): the A operand of tsgnn_rowgemm_f32 is replaced by
 *   z[r,:] = sum_k x[ell[r*ell_w + k], :K]      (ell = fixed-width neighbour table of tsgnn_csr_to_ell, entries < 0 skipped,
 *                                               ell_w in {4, 8, 16}) + the rows' CSR tail (tail_ptr[rows+1], tail_col; both
 *                                               NULL when no list is longer than ell_w: tsgnn_csr_tail_fill),
 * gathered chunk by chunk while it is staged for the MFMAs.  zout (nullable) receives z for rows [0, rows) (the weight
 * gradient needs it; its columns [K, roundup4(K)) get the aggregated row padding of x, which must be finite).  N <= 128;
 * everything else as tsgnn_rowgemm_f32. */
int tsgnn_gather_rowgemm_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* b, int64_t ldb, int trans_b,
                             const float* bias, float* c, int64_t ldc, float* rinv, float* zout, int64_t ldz, int64_t rows, int K,
                             int N, int normalize, int64_t fill_rows, tsgnn_stream_t stream);
/* tsgnn_gather_rowgemm_f32 (trans_b = 0, normalize = 1, 96 < N <= 128, K <= 128) for a layer that is followed by the slot
 * batch-norm (apply_bn, encoders.py:134-138) WITHOUT a launch for it (tail_ptr / tail_col as there): the epilogue adds every real row's
 * (sum_f relu(v), sum_f relu(v)^2) to sums[2 * row_slot[r]] as 64-bit fixed-point integers (2^-40 units: integer addition is
 * associative, so the totals do not depend on the order the panels finish in — bitwise reproducible) and the filler block leaves
 * the ghost row's two numbers in ghost[0..1].  sums [2 * nslots], 16-byte aligned, zero before the launch; row_slot[r] < 0: row r
 * belongs to no graph (padding of a capacity-padded batch).  Consumer: tsgnn_sage_layer_fwd_bn_f32. */
int tsgnn_gather_rowgemm_st_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* b, int64_t ldb, const float* bias,
                                float* c, int64_t ldc, float* rinv, float* zout, int64_t ldz, int64_t rows, int K, int N,
                                int64_t fill_rows, const int* row_slot, unsigned long long* sums, float* ghost, tsgnn_stream_t stream);
/* tsgnn_sage_layer_fwd[_ro]_f32 for a layer whose INPUT's slot batch-norm has no launch of its own: x = the previous layer's
 * normalised pre-activations v, sums_in / ghost_in = what its statistics epilogue left, slot_count[n] = graphs with more than n
 * nodes.  Every row-panel block turns the sums into (mean, rstd) per slot — exact from the integers, ghost copies by their
 * multiplicity B - slot_count[n], biased variance, eps 1e-5 — and gathers y_j = (relu(v_j) - mean[slot_j]) * rstd[slot_j]; the
 * readout partial does the same for the rows it scans, and its blocks of graph 0 write mean_out / rstd_out [nslots] for the
 * backward (tsgnn_slot_post_bwd_f32).  ell / tail_col: entry = slot << 20 | row, nslots <= 1024, rows < 2^20, n_ghost = nslots.
 * row_slot != NULL: this layer is followed by a batch-norm as well, its statistics go to sums_out / ghost_out (zero before);
 * packed_out != NULL: last layer, readout epilogue as in tsgnn_sage_layer_fwd_ro_f32 (not both).  ro_map (nullable): see
 * tsgnn_sage_layer_fwd_bn_plan; ignored unless ro_map_ch equals the chunk size the launch uses. */
int tsgnn_sage_layer_fwd_bn_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                                float* v, int64_t ldv, float* rinv, float* zout, int64_t ldz, int64_t rows, int K, int64_t fill_rows,
                                const int* graph_ptr, const int* slot_count, int B, int nslots, int n_ghost, unsigned long long* packed,
                                unsigned long long* packed_out, const int* row_graph, const unsigned long long* sums_in,
                                const float* ghost_in, float* mean_out, float* rstd_out, const int* row_slot,
                                unsigned long long* sums_out, float* ghost_out, const int* ro_map, int ro_map_ch, tsgnn_stream_t stream);
/* HOST function: slots per readout block (64 / 128 / 256) and row-panel blocks (filler included) of that launch — what a caller needs to
 * build ro_map: a permutation of the B * ceil(nslots / ro_ch) readout work items (graph * chunks + chunk) that puts the blocks scanning
 * a graph on the XCD whose row panels gather it (workgroups b, b + 8, ... share an XCD; the k-th readout block is workgroup n_gemm + k).
 * Speed only: any permutation gives the same results. */
int tsgnn_sage_layer_fwd_bn_plan(int64_t rows, int64_t fill_rows, int B, int nslots, int* ro_ch, int* n_gemm);
/* Forward of a hidden 128 -> 128 GraphConv layer in ONE launch together with the max-readout partial of its INPUT x (the
 * previous layer's output; both only read x): tsgnn_gather_rowgemm_f32(normalize = 1, fill_rows) + tsgnn_readout_partial_f32
 * over x into packed[B*128] (layout and ghost-row rule as there; n_real = rows). */
int tsgnn_sage_layer_fwd_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                             float* v, int64_t ldv, float* rinv, float* zout, int64_t ldz, int64_t rows, int K, int64_t fill_rows,
                             const int* graph_ptr, int B, int nslots, int n_ghost, unsigned long long* packed, tsgnn_stream_t stream);
/* The same launch when the layer is the LAST of the stack (no slot batch-norm follows, encoders.py:182-183): the product's
 * epilogue also folds the max readout of the layer's own output v into packed_out[B*128] (packed (ordered value, ~row)
 * atomicMax; real rows from the row panels, each graph's first ghost row from the filler block), so no pass over v is needed
 * for it.  packed_out: zeroed by the caller, nullable (= tsgnn_sage_layer_fwd_f32); row_graph[rows] = graph of each row. */
int tsgnn_sage_layer_fwd_ro_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                                float* v, int64_t ldv, float* rinv, float* zout, int64_t ldz, int64_t rows, int K, int64_t fill_rows,
                                const int* graph_ptr, int B, int nslots, int n_ghost, unsigned long long* packed,
                                unsigned long long* packed_out, const int* row_graph, tsgnn_stream_t stream);
/* Backward of a hidden 128 -> 128 GraphConv layer's GEMM-shaped halves in ONE launch (both consume du): the weight / bias
 * gradient slabs of tsgnn_linear_wgrad_f32 (dw == NULL form: reduce ws later with tsgnn_wgrad_reduce_multi_f32; plan with
 * tsgnn_linear_wgrad_plan(rows, 128, 128, ...)) and dxs = (A du) w^T of tsgnn_gather_rowgemm_f32 (trans_b = 1, symmetric A).
 * A CU hosts one block of each grid, so the two run side by side instead of back to back. */
int tsgnn_sage_layer_bwd_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* du, int64_t lddu, const float* w, int64_t ldw, float* dxs,
                             int64_t lddxs, const float* z, int64_t ldz, int64_t rows, int nslab, int64_t rows_per_slab,
                             int64_t bias_only_rows, float* ws, tsgnn_stream_t stream);
/* backward of the row normalisation: du = rinv * (dv - v (v.dv)) */
int tsgnn_l2norm_bwd_f32(const float* v, int64_t ldv, const float* dv, int64_t lddv, const float* rinv, float* du,
                         int64_t lddu, int64_t rows, int F, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- slot batch-norm + readout (bn_readout.hip) */

/* Rows: [0,n_real) real nodes graph after graph; [n_real, n_real+n_ghost) one "ghost" row per node slot
 * standing for the reference's padded rows of that slot (n_ghost = nmax, multiplicity B - slot_count[n])
 * or n_ghost = 0 (padded layout: every row is materialised).
 * y = (act(v) - mean[slot]) * rstd[slot], act = relu if relu; bn = 0: y = act(v).
 * Replaces self.act + apply_bn (encoders.py:179-181, 134-138): fresh BatchNorm1d(Nmax) per call. */
int tsgnn_bn_slots_fwd_f32(const int* graph_ptr, const int* slot_count, const int* row_slot, int B, int nmax,
                           int64_t n_real, int n_ghost, const float* v, int64_t ldv, int F, int relu, int bn,
                           float* mean, float* rstd, float* y, int64_t ldy, tsgnn_stream_t stream);
int tsgnn_bn_slots_bwd_f32(const int* graph_ptr, const int* slot_count, const int* row_slot, int B, int nmax,
                           int64_t n_real, int n_ghost, const float* v, int64_t ldv, const float* dy, int64_t lddy, int F,
                           int relu, int bn, const float* mean, const float* rstd, float* m1, float* m2, float* dv,
                           int64_t lddv, tsgnn_stream_t stream);

/* Per-graph statistics: the same ReLU + apply_bn when every graph is normalised as if it were alone in its batch (B = 1,
 * how tripletnet.py:36-38 calls the encoder): a per-row layer norm over the features, biased variance, eps 1e-5. */
int tsgnn_row_ln_fwd_f32(const float* v, int64_t ldv, int64_t rows, int F, int relu, float* mean, float* rstd, float* y, int64_t ldy,
                         tsgnn_stream_t stream);
int tsgnn_row_ln_bwd_f32(const float* v, int64_t ldv, const float* dy, int64_t lddy, int64_t rows, int F, int relu, const float* mean,
                         const float* rstd, float* dv, int64_t lddv, tsgnn_stream_t stream);

/* Backward of [max readout ; tsgnn_row_ln_fwd_f32 ; L2 normalise] of a hidden GraphConv layer in one pass (the row-local counterpart
 * of tsgnn_slot_post_bwd_f32 for per-graph statistics): rows [0, n_real) are real rows (graph row_graph[r]), rows [n_real, rows) ghost
 * rows (row n_real + s = the padded slot s of every graph; it can be the readout winner of any graph).  dy = dxs (real rows; nullable)
 * + the readout gradient dout[b, f] where arg[b * F + f] == r (both nullable together); ln = 0 / relu = 0 skip those stages;
 * du = rinv (dv - v <v, dv>).  F <= 256. */
int tsgnn_row_post_bwd_f32(const int* row_graph, int B, int64_t n_real, int64_t rows, const float* v, int64_t ldv, const float* dxs,
                           int64_t lddxs, const float* dout, int64_t ldo, const int* arg, int F, int relu, int ln, const float* mean,
                           const float* rstd, const float* rinv, float* du, int64_t lddu, tsgnn_stream_t stream);

/* out[b,f] = max over the nmax node slots of graph b (ghost rows included, trap T5), arg = winning row.
 * Replaces torch.max(x, dim=1) (encoders.py:183,190,197,353,383).  ONE launch.  packed_ws: tsgnn_readout_max_ws_words(B, nmax, F)
 * uint64 words (B * F packed maxima, then the graphs' arrival counters), ALL zero on entry and all zero again on return: a workspace
 * zeroed once serves every later call issued on the same stream. */
int tsgnn_readout_max_ws_words(int B, int nmax, int F);   /* <= 0: the shape is out of range */
int tsgnn_readout_max_fwd_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                              const float* x, int64_t ldx, int F, int relu, unsigned long long* packed_ws, float* out,
                              int64_t ldo, int* arg, tsgnn_stream_t stream);
/* dx[arg[b,f], f] += dout[b,f] (dx pre-initialised by the caller) */
int tsgnn_readout_max_bwd_f32(const float* dout, int64_t ldo, const int* arg, int B, int F, const float* x, int64_t ldx_in,
                              int relu, int64_t n_real, float* dx, int64_t ldx, tsgnn_stream_t stream);
/* the same as a dense pass that writes every element of dx[rows_total, F] (no zero fill, no atomics): rows [0, rows) can hold a
 * maximum (row_graph[rows] = their graphs) — all rows of a batch without ghost rows, or the real rows when the caller discards the
 * ghost rows' gradient; add (nullable): a second gradient of the same tensor, summed in the same pass.  F % 4 == 0, 16-byte rows. */
int tsgnn_readout_max_bwd_rows_f32(const float* dout, int64_t ldo, const int* arg, const int* row_graph, int F, int64_t rows,
                                   int64_t rows_total, const float* add, int64_t ldadd, float* dx, int64_t ldx, tsgnn_stream_t stream);

/* padded [B,nmax,F] (graph_sampler.py:110-114) <-> packed rows */
int tsgnn_pack_rows_f32(const float* src, int nmax, int F, const int* row_graph, const int* row_slot, int64_t n_real,
                        float* dst, int64_t ld, tsgnn_stream_t stream);
int tsgnn_unpack_rows_f32(const int* graph_ptr, int B, int nmax, int64_t n_real, int n_ghost, const float* src, int64_t ld,
                          int F, float fill, float* dst, tsgnn_stream_t stream);
int tsgnn_unpack_rows_bwd_ghost_f32(const int* graph_ptr, int B, int nmax, int64_t n_real, const float* ddst, int F,
                                    float* dsrc, int64_t ld, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- optimiser on a flat buffer (optim.hip) */

/* loss = mean_b CE(logits[b,:], label[b]) and dlogits = d loss / d logits in one launch
 * (model.loss -> F.cross_entropy, encoders.py:221-224; labels int64 as the reference passes them). */
int tsgnn_softmax_ce_f32(const float* logits, int64_t ld, const int64_t* label, int B, int C, float* loss, float* dlogits,
                         tsgnn_stream_t stream);
/* Same optimiser step when the gradient producers already left shares of |grad|^2 in parts[0..nparts) (normparts of
 * tsgnn_wgrad_reduce_multi_f32 / tsgnn_head2_bwd_f32) and advanced state[0]: one launch, no device-wide barrier.  Single GPU
 * only (after an all-reduce the local shares no longer describe the gradient). */
int tsgnn_adam_from_partials_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, float max_norm, float* state, const float* parts, int nparts,
                                 const float* poison, tsgnn_stream_t stream);
/* clip_grad_norm(max_norm) + Adam.step() of the reference loop (train.py:128-129) on one flat fp32
 * parameter / gradient buffer (the buffer RCCL all-reduces): grad is first scaled by grad_scale
 * (1/world_size).  state: 4 floats {step, grad_norm, applied scale, skipped} (zeroed before the first step).
 * poison (nullable, both optimiser entry points): one device float, the device's error word — a kernel with a bounded
 * device-wide barrier that could not complete it (tsgnn_dense_stack_*_f32) stores a non-zero value there.  The optimiser reads
 * it first and, if set, leaves parameters, moments and the step counter untouched and sets state[3] = 1: an invalid gradient is
 * never applied, and the host raises at its next synchronisation; ws >= 258 floats, 8-byte aligned, zeroed once (word 256 is the sign-off counter of the one-launch variant used
 * for n <= 131,072: every block sums the norm of the whole gradient itself, so there is NO device-wide barrier, no
 * co-residency assumption and no partial update; larger n: norm partials, norm, update = three launches). */
int tsgnn_clip_adam_step_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float max_norm, float grad_scale, float* state,
                             float* ws, const float* poison, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- pooled-level GCN stacks in one launch (dense_stack.hip) */

/* 1 if tsgnn_dense_stack_*_f32 take these shapes: B graphs of K <= 64 nodes, nstack (1 or 2) stacks of L <= 4 layers, input
 * width fin0 <= 192, hidden width, last-layer widths of the two stacks (<= 128). */
int tsgnn_dense_stack_supported(int B, int K, int nstack, int L, int fin0, int hidden, int last0, int last1);
/* GCN stack(s) of a pooled DiffPool level (encoders.py:378-380 -> gcn_forward :140-167, dense adjacency adj[B,K,K], x[B*K, fin0]),
 * forward / backward as ONE launch each, for up to two stacks that share (x, adj).  Per hidden layer: u = (A x) W + b,
 * v = u / max(|u|, 1e-12), y = BN_slot(ReLU(v)) (apply_bn, :134-138: fresh statistics over batch and features, eps 1e-5, biased
 * variance); last layer: v only; out[:, off_l : off_l + n_l] = the layer's output.  The workgroups (16 rows of one graph each,
 * all resident) meet at a device-wide barrier per hidden layer (per-slot statistics) forward, twice per layer backward; a
 * barrier that is not completed within its bound raises err[0]: the optimiser entry points skip their update while it is set
 * (`poison`) and the host raises at its next synchronisation (message_passing.check_device_errors).
 * `desc`: host array of 8-byte words (pointers / integers), layout in csrc/dense_stack.hip::ds_unpack, built by
 * two_stage_gnn_amd/dense_stack.py::_describe. */
int tsgnn_dense_stack_fwd_f32(const int64_t* desc, tsgnn_stream_t stream);
int tsgnn_dense_stack_bwd_f32(const int64_t* desc, tsgnn_stream_t stream);
/* Workgroups of the dense-stack kernels the CURRENT device keeps resident at once (compute units x the runtime's occupancy for
 * them): the grid limit tsgnn_dense_stack_supported applies.  0 without a usable device. */
int tsgnn_dense_stack_max_resident(void);
/* Test hook for the failure path of the bounded device-wide barriers: one workgroup waits for `expect` arrivals on the barrier
 * words `sync` (>= 32 words, zero).  expect = 1 passes; expect = 2 can never complete: the spin gives up at its bound (~1 s) and
 * stores 1.0 to *err, as a dense-stack launch does whose workgroups were not all resident. */
int tsgnn_dense_stack_barrier_selftest(unsigned* sync, float* err, int expect, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- edge-softmax attention (attention.hip) */

/* s[r,h] = <x[r, head h], a[h]>  — the two per-node scalars a1.h_i / a2.h_j that replace the reference's
 * [N,N,2F] pair tensor (encoders_GAT.py:35-36). */
int tsgnn_node_scores_f32(const float* x, int64_t ldx, int64_t rows, int H, int Fh, const float* a, int64_t lda, float* s,
                          tsgnn_stream_t stream);
/* both score vectors of a head in one pass over the rows: s1 = x . a1, s2 = x . a2 (encoders_GAT.py:34-36) */
int tsgnn_node_scores2_f32(const float* x, int64_t ldx, int64_t rows, int H, int Fh, const float* a1, int64_t lda1, float* s1,
                           const float* a2, int64_t lda2, float* s2, tsgnn_stream_t stream);
/* alpha[e,h] = softmax over the entries e of CSR row g of LeakyReLU(s_grp[g,h] + s_oth[col[e],h]).
 * Reference DGATHead (encoders_GAT.py:36-41, softmax over dim=1 = per column, trap T3): call on A^T with
 * s_grp = a2.h, s_oth = a1.h.  PyG GATConv: call on A (rows = targets).  mod > 0: node index = id % mod. */
int tsgnn_edge_softmax_fwd_f32(const int* rowptr, const int* col, int64_t rows, int H, const float* s_grp, const float* s_oth,
                               int mod, float slope, float* alpha, tsgnn_stream_t stream);
int tsgnn_edge_softmax_bwd_f32(const int* rowptr, const int* col, int64_t rows, int H, const float* s_grp, const float* s_oth,
                               int mod, float slope, const float* alpha, const float* dalpha, float* dt, float* ds_grp,
                               tsgnn_stream_t stream);
/* per-entry values between a CSR and its transpose: gather dst[p] = src[perm[p]] / scatter dst[perm[p]] = src[p] */
int tsgnn_edge_permute_f32(const float* src, const int* perm, int64_t n, int H, int scatter, float* dst, tsgnn_stream_t stream);
/* out[r,h] = sum of val[e,h] over the entries of row r */
int tsgnn_csr_row_sum_f32(const int* rowptr, const float* val, int64_t rows, int H, float* out, tsgnn_stream_t stream);
/* the same for values stored in another entry order: entry e's value is val[eperm[e]] */
int tsgnn_csr_row_sum_perm_f32(const int* rowptr, const float* val, const int* eperm, int64_t rows, int H, float* out,
                               tsgnn_stream_t stream);
/* y[r, head h] = sum_e alpha[e,h] * x[col[e], head h]   (h_prime = attention @ h, encoders_GAT.py:43) */
int tsgnn_csr_spmm_heads_f32(const int* rowptr, const int* col, const float* alpha, int H, int Fh, const float* x, int64_t ldx,
                             int mod, float* y, int64_t ldy, int64_t rows, tsgnn_stream_t stream);
/* the same aggregation with the element-wise passes that follow it folded into its epilogue:
 * y[r, h*Fh+f] += w1[r,h] * a1[h*lda1+f] + w2[r,h] * a2[h*lda2+f] + uscale * (uw ? uw[r,h] : 1) * u[(r / rows_per_seg)*ldu + h*Fh+f]
 * (each term optional: NULL w1 / w2 / u; row_seg (nullable) names every row's segment for ragged batches and replaces
 * r / rows_per_seg; eperm (nullable): entry e's weights are alpha[eperm[e]], i.e. alpha is stored in the entry order of the
 * transposed structure — the column softmax produces it there — and no permuted copy is made).  Forward: the uniform 1/N contribution of all-masked softmax columns
 * (encoders_GAT.py:38-41); backward: dh = A^T-aggregation + ds_row (x) a_row + ds_col (x) a_col + iso * du / N. */
int tsgnn_csr_spmm_heads_epi_f32(const int* rowptr, const int* col, const float* alpha, int H, int Fh, const float* x, int64_t ldx,
                                 int mod, float* y, int64_t ldy, int64_t rows, const float* w1, const float* a1, int64_t lda1,
                                 const float* w2, const float* a2, int64_t lda2, const float* u, int64_t ldu, const float* uw,
                                 int rows_per_seg, const int* row_seg, float uscale, const int* eperm, tsgnn_stream_t stream);
/* dalpha[e,h] = <dy[row(e), head h], x[col[e], head h]>  (sampled dense-dense product) */
int tsgnn_csr_sddmm_heads_f32(const int* rowptr, const int* col, int H, int Fh, const float* dy, int64_t lddy, const float* x,
                              int64_t ldx, int mod, float* dalpha, int64_t rows, tsgnn_stream_t stream);
/* out[s,c] = scale * sum_{r in segment s} w[r, c/Fh] * x[r,c]  (w NULL = 1; seg_ptr NULL = one segment of `rows`;
 * mean != 0 divides by the segment length; max_seg = longest segment).  ws >= ceil(max_seg/128)*nseg*H*Fh floats.
 * Also PyG global_mean_pool (Code/sag/network.py:36). */
int tsgnn_segment_wsum_f32(const float* x, int64_t ldx, const float* w, int H, int Fh, const int* seg_ptr, int nseg, int64_t rows,
                           int64_t max_seg, float scale, int mean, float* ws, float* out, int64_t ldo, tsgnn_stream_t stream);
/* two weighted sums of the same rows in one pass (the gradients of both attention vectors): out1 = sum w1 x, out2 = sum w2 x;
 * ws: twice the floats tsgnn_segment_wsum_f32 needs */
int tsgnn_segment_wsum2_f32(const float* x, int64_t ldx, const float* w1, const float* w2, int H, int Fh, const int* seg_ptr, int nseg,
                            int64_t rows, int64_t max_seg, float scale, float* ws, float* out1, float* out2, int64_t ldo,
                            tsgnn_stream_t stream);
/* out[s, :C] = scale * sum over the listed rows of segment s (idx / w entries [seg_ptr[s], seg_ptr[s+1])) of w[e] * x[idx[e], :C]:
 * the uniform softmax term only involves a graph's few edge-less columns (encoders_GAT.py:38-41) */
int tsgnn_gather_wsum_f32(const float* x, int64_t ldx, const int* idx, const float* w, const int* seg_ptr, int nseg, int C, float scale,
                          float* out, int64_t ldo, tsgnn_stream_t stream);
/* y[r,c] += scale * w[r,c/Fh] * (a ? a[(c/Fh)*lda + c%Fh] : u[(r / rows_per_seg)*ldu + c])   (w NULL = 1) */
int tsgnn_broadcast_add_f32(float* y, int64_t ldy, int64_t rows, int H, int Fh, const float* w, const float* a, int64_t lda,
                            const float* u, int64_t ldu, int rows_per_seg, float scale, tsgnn_stream_t stream);
/* The per-head parameters of a DGATLayer (attention_{i}.w [Fin, Fo], attention_{i}.a [2*Fo, 1]; encoders_GAT.py:22-25,60-62)
 * <-> the fused operands of the layer, one launch each way: W[Fin, H*Fo] = [w_0 | w_1 | ...], A[2, H, Fo] (A[0]: the halves
 * that meet h_i, A[1]: those that meet h_j, :34-36); backward gw[H, Fin, Fo], ga[H, 2*Fo] from dW / dA[0] / dA[1] (each
 * nullable = zero).  H <= tsgnn_pack_heads_max(); unused head pointers NULL. */
int tsgnn_pack_heads_max(void);
int tsgnn_pack_heads_f32(const float* w0, const float* w1, const float* w2, const float* w3, const float* w4, const float* w5,
                         const float* w6, const float* w7, const float* a0, const float* a1, const float* a2, const float* a3,
                         const float* a4, const float* a5, const float* a6, const float* a7, int H, int Fin, int Fo, float* W, float* A,
                         tsgnn_stream_t stream);
int tsgnn_unpack_heads_f32(const float* dW, const float* dA0, const float* dA1, int H, int Fin, int Fo, float* gw, float* ga,
                           tsgnn_stream_t stream);

/* ---- one GAT layer, all heads, fused (csrc/gat_fused.hip).  Replaces DGATHead.forward / DGATLayer.forward
 * (Code/sage+gat+diffpool/encoders_GAT.py:29-49, 68-84) for batches whose edge-less columns are listed per graph.
 * hp[rows, ldh] = x . W' with W' = [W_0 | .. | W_{H-1} | W_h a1_h (H columns) | W_h a2_h (H columns) | 0 pad] (Ns columns):
 * features, then the two attention scalars of every head (:35-36).  See the file header for the data flow. */
int tsgnn_gat_fused_supported(int H, int Fh);
/* y[i] = act( sum_j alpha_ij hp[j, :C] + sum_{edge-less j of i's graph} (iso_w / N) hp[j, :C] ), alpha = softmax over dim=1
 * (column-wise, :41) of the masked LeakyReLU scores; mean_heads: mean over heads before the ELU (:78-83), y is [rows, Fh].
 * (rowptr, col): A; (rp_t, col_t): A^T.  row_graph nullable (then graph = row / nmax).  iso_*: nullable together; iso_ptr[B + 1].
 * drop_p > 0: attention dropout (:42), Philox4x32-10 keyed on (seed + *drop_ctr; i, j, head); drop_ctr (nullable) is a DEVICE counter
 * read when the kernel starts, so a step replayed from a hipGraph that advances it draws a new mask every replay.
 * stat[rows, H, 2] (out): (max, 1 / sum of exponentials) of every softmax column, written by a first small launch and read by
 * the attention kernel here and by the backward. */
int tsgnn_gat_attn_fwd_f32(const float* hp, int64_t ldh, const int* rowptr, const int* col, const int* rp_t, const int* col_t,
                           int64_t rows, int H, int Fh, float slope, const int* row_graph, int nmax, const int* iso_idx,
                           const float* iso_w, const int* iso_ptr, float uscale, int mean_heads, int apply_elu, float drop_p,
                           uint64_t seed, const unsigned long long* drop_ctr, float* stat, float* y, int64_t ldy, tsgnn_stream_t stream);
/* backward, column-wise: dhp[:, :C] (features), dhp[:, C+H+h] (d s_col), zero pad columns; per-entry terms t1, t2 [nnz, H] (A^T
 * entry order) and S [rows, H] for tsgnn_gat_score_rowsum_f32, which writes dhp[:, C+h] (d s_row) and — from the partial sums
 * dupart[B, tsgnn_gat_bwd_parts(B), C] of dpre over each graph's rows — dh of the listed edge-less columns.  dy, y: the layer's
 * output gradient and output ([rows, Fh] with mean_heads).  iso_row[rows * iso_row_ld]: per-row weight of the edge-less columns.
 * With drop_p > 0 the listed columns are completed by the backward itself (every element has its own mask): pass
 * dupart = NULL to the row-sum kernel. */
int tsgnn_gat_bwd_parts(int B);
int tsgnn_gat_attn_bwd_f32(const float* hp, int64_t ldh, const float* y, int64_t ldy, const float* dy, int64_t lddy, const int* rp_t,
                           const int* col_t, int64_t rows, int H, int Fh, float slope, int mean_heads, int apply_elu,
                           const int* graph_ptr, int B, const int* iso_idx, const float* iso_w, const int* iso_ptr,
                           const float* iso_row, int iso_row_ld, float uscale, float drop_p, uint64_t seed,
                           const unsigned long long* drop_ctr, const float* stat, float* dhp, int Ns, float* t1, float* t2, float* S,
                           float* dupart, tsgnn_stream_t stream);
/* the same for a layer whose output feeds ONLY the max readout over each graph's rows (the last layer, encoders_GAT.py:189): dy NULL,
 * and instead the readout's gradient ro_dout [B, Co] (leading dimension ro_ldo), its winners ro_arg [B, Co] (rows) and row_graph
 * [rows]: dy[i, c] = (ro_arg[b, c] == i) ? ro_dout[b, c] : 0 with b = row_graph[i] is formed on the fly — neither the [rows, Co]
 * gradient tensor nor the pass that would write it (tsgnn_readout_max_bwd_rows_f32) exists.  Co = the layer's output width. */
int tsgnn_gat_attn_bwd_ro_f32(const float* hp, int64_t ldh, const float* y, int64_t ldy, const float* dy, int64_t lddy, const int* rp_t,
                              const int* col_t, int64_t rows, int H, int Fh, float slope, int mean_heads, int apply_elu,
                              const int* graph_ptr, int B, const int* iso_idx, const float* iso_w, const int* iso_ptr,
                              const float* iso_row, int iso_row_ld, float uscale, float drop_p, uint64_t seed,
                              const unsigned long long* drop_ctr, const float* stat, float* dhp, int Ns, float* t1, float* t2, float* S,
                              float* dupart, const float* ro_dout, int64_t ro_ldo, const int* ro_arg, const int* row_graph,
                              tsgnn_stream_t stream);
/* the dropout multipliers (0 or 1 / (1 - p)) of attention elements (i0 + i, j0 + j) of every head, out[ni, nj, H] — what the two
 * kernels above apply; lets a test hand the very same mask to the dense oracle */
int tsgnn_gat_dropout_mult_f32(float drop_p, uint64_t seed, const unsigned long long* drop_ctr, int64_t i0, int64_t ni, int64_t j0,
                               int64_t nj, int H, float* out, tsgnn_stream_t stream);
int tsgnn_gat_score_rowsum_f32(const int* rowptr, const int* col, const int* eperm, const float* t1, const float* t2, const float* S,
                               int64_t rows, int H, float* dhp, int64_t ldh, int C, const float* dupart, int B, const int* iso_idx,
                               const float* iso_w, const int* iso_ptr, float uscale, tsgnn_stream_t stream);
/* heads' parameters -> W' (and dW' -> the heads' gradients) for up to 4 layers in ONE launch.  desc lives in HOST memory:
 * [L, then per layer: H, Fin, Fo, Ns, W' (pack) or dW' (unpack), gw [H, Fin, Fo], ga [H, 2 Fo], w_0..w_7, a_0..a_7]
 * (tsgnn_gat_pack_desc_words() words per layer; gw / ga are read by the unpack only). */
int tsgnn_gat_pack_desc_words(void);
int tsgnn_gat_pack_f32(const int64_t* desc, tsgnn_stream_t stream);
int tsgnn_gat_unpack_f32(const int64_t* desc, tsgnn_stream_t stream);
/* ELU (encoders_GAT.py:47) / mean over heads then ELU (:78-83) */
int tsgnn_elu_heads_fwd_f32(const float* x, int64_t rows, int H, int Fh, int mean_heads, int apply_elu, float* y, tsgnn_stream_t stream);
int tsgnn_elu_heads_bwd_f32(const float* x, const float* dy, int64_t rows, int H, int Fh, int mean_heads, int apply_elu, float* dx,
                            tsgnn_stream_t stream);

/* ---------------------------------------------------------------- SAGPool graph-level head (mlp_head.hip) */

/* Code/sag/network.py:48-53 in one launch: a1 = relu(x W1^T + b1) * keep * keep_scale (dropout mask keep[B, D1] of 0 / 1,
 * NULL = no dropout), a2 = relu(a1 W2^T + b2), logp = log_softmax(a2 W3^T + b3).  W* in nn.Linear's [out, in] layout,
 * x[B, D0].  a1, a2 are kept for the backward.  D0 % 4 == 0, D1 % 4 == 0, C <= 16, 16-byte aligned W1 / W2. */
int tsgnn_mlp3_supported(int B, int D0, int D1, int D2, int C);
int tsgnn_mlp3_fwd_f32(const float* x, int64_t ldx, const float* w1, const float* b1, const float* keep, float keep_scale, const float* w2,
                       const float* b2, const float* w3, const float* b3, int B, int D0, int D1, int D2, int C, float* a1, float* a2,
                       float* logp, tsgnn_stream_t stream);
/* tsgnn_mlp3_fwd_f32 with the dropout mask (F.dropout after the first ReLU, network.py:49: probability p of dropping, survivors
 * scaled by 1 / (1 - p)) made INSIDE the launch: Philox4x32-10 keyed on (seed + the device counter state[0]; graph, hidden unit).
 * state: two uint64 words, zeroed once: [0] counts the launches — the launch advances it itself, so a step replayed from a hipGraph
 * draws a new mask at every replay and no host generator is involved —, [1] is its ticket word (zero between launches).
 * used[0] receives the counter value this launch keyed its mask with; tsgnn_mlp3_dropout_mask_f32 regenerates that mask. */
int tsgnn_mlp3_fwd_drop_f32(const float* x, int64_t ldx, const float* w1, const float* b1, float p, uint64_t seed,
                            unsigned long long* state, unsigned long long* used, const float* w2, const float* b2, const float* w3,
                            const float* b3, int B, int D0, int D1, int D2, int C, float* a1, float* a2, float* logp,
                            tsgnn_stream_t stream);
int tsgnn_mlp3_dropout_mask_f32(float p, uint64_t seed, uint64_t counter, int B, int D1, float* out, tsgnn_stream_t stream);
/* its backward in one launch ([dW1 tiles | dW2 tiles | dW3 | dX rows] blocks, each recomputing dlogits -> dz2 in LDS);
 * dlogp[B, C] = gradient w.r.t. logp; dx nullable. */
int tsgnn_mlp3_bwd_f32(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                       const float* a2, const float* logp, const float* dlogp, float keep_scale, int B, int D0, int D1, int D2, int C,
                       float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx, int64_t lddx,
                       tsgnn_stream_t stream);
/* the same backward as two short launches (rows: dlogits -> dz2 -> dz1 -> dx once per row; weights: tiles that sum over the
 * rows); ws: B * (C + D2 + D1) floats of scratch */
int tsgnn_mlp3_bwd2_f32(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                        const float* a2, const float* logp, const float* dlogp, float keep_scale, int B, int D0, int D1, int D2, int C,
                        float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx, int64_t lddx, float* ws,
                        tsgnn_stream_t stream);
/* the same with F.nll_loss(logp, label) (mean; Code/sag/train.py) folded in: dlogits = (softmax - onehot) / B is formed inside
 * the rows kernel and the loss value is written to loss[0] by the weights kernel */
int tsgnn_mlp3_bwd2_nll_f32(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                            const float* a2, const float* logp, const int64_t* label, float* loss, float keep_scale, int B, int D0,
                            int D1, int D2, int C, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx,
                            int64_t lddx, float* ws, tsgnn_stream_t stream);

/* the same two launches that also leave |grad|^2 shares: normparts[tsgnn_mlp3_bwd2_norm_blocks(D1, D2)], entry k = the sum of squares
 * of the gradient entries block k of the weights launch wrote (label != NULL: the nll form, dlogp ignored) — the head's part of the
 * norm for tsgnn_adam_from_partials_f32 when its six gradients go straight into the optimiser's flat bucket */
int tsgnn_mlp3_bwd2_norm_blocks(int D1, int D2);
int tsgnn_mlp3_bwd2_np_f32(const float* x, int64_t ldx, const float* w1, const float* w2, const float* w3, const float* a1,
                           const float* a2, const float* logp, const float* dlogp, const int64_t* label, float* loss, float keep_scale, int B,
                           int D0, int D1, int D2, int C, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* dx,
                           int64_t lddx, float* ws, float* normparts, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- DiffPool link-prediction side loss (linkpred.hip) */

/* encoders.py:416-440 for adj_hop = 1, value and gradient in one pass, no [B,N,N] tensor:
 *   loss = inv_entries * sum_b sum_{i,j in graph b} BCE(min((S S^T)_ij, clamp), a_ij),  eps = 1e-7
 * S[rows, K] (K <= 128): assignment rows of the batch, graph b = rows [graph_ptr[b], graph_ptr[b+1]) (the reference's
 * adj_mask: only pairs inside a graph's real rows count); slabs: every graph's rows cut into pieces of at most
 * tsgnn_linkpred_tile_rows() rows (slab_row_ptr[nslab+1], slab_graph[nslab]); (rowptr, col, val) the adjacency (val NULL =
 * unit), (rowptr_t, col_t, val_t) its transpose or NULLs when it is symmetric.  dS[rows, K] = d loss / d S (rows outside
 * every slab are left untouched: zero them); ws: tsgnn_linkpred_chunks() * rows * ldd floats (per-chunk partial gradients,
 * every row of every slab is written); part: tsgnn_linkpred_chunks() * nslab + 2 * ceil(rows / 4) floats; loss: 1 float.
 * clamp: the reference passes an UNINITIALISED one-element tensor (:424); 1.0 is the value it presumably meant. */
int tsgnn_linkpred_tile_rows(void);
int tsgnn_linkpred_chunks(void);
int tsgnn_linkpred_loss_f32(const float* S, int64_t lds, int K, int64_t rows, const int* slab_row_ptr, const int* slab_graph,
                            int nslab, const int* graph_ptr, const int* rowptr, const int* col, const float* val,
                            const int* rowptr_t, const int* col_t, const float* val_t, float clamp, float inv_entries, float* dS,
                            int64_t ldd, float* ws, float* part, float* loss, tsgnn_stream_t stream);
/* the same with two row operands, p_ij = <X_i, Y_j>, for adj_hop > 1 (encoders.py:419-423: sum_p (S S^T)^p = (S M) S^T):
 * dX[i] = sum_j (dL/dp_ij) Y_j over the pairs of every graph and the entries (i, j) of the given CSR; the loss value with
 * count_loss.  A second call with the operands swapped and the transposed CSR gives the other operand's gradient. */
int tsgnn_linkpred_loss_xy_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int K, int64_t rows, const int* slab_row_ptr,
                               const int* slab_graph, int nslab, const int* graph_ptr, const int* rowptr, const int* col,
                               const float* val, float clamp, float inv_entries, int count_loss, float* dX, int64_t ldd, float* ws,
                               float* part, float* loss, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- SAGPool path (pooling.hip) */

/* PyG topk(score, ratio, batch) (call site Code/sag/layers.py:20): for every graph b keep its
 * k_b = k_ptr[b+1]-k_ptr[b] highest-scoring nodes, descending; ties -> smaller node id.
 * perm[k_ptr[b] + t] = global node id.  new_id (nullable, one int per node): filter_adj's relabelling map, new_id[node] =
 * its position in perm or -1 when dropped.  Graphs of more than tsgnn_topk_max_segment() nodes are unsupported. */
int tsgnn_topk_max_segment(void);
int tsgnn_topk_segments_f32(const float* score, const int* graph_ptr, const int* k_ptr, int B, int max_seg, int* perm,
                            int* new_id, tsgnn_stream_t stream);
/* out[p,:] = x[perm[p],:] * tanh(score[perm[p]])  (Code/sag/layers.py:21); use_tanh = 0: gate = score */
int tsgnn_gather_gate_fwd_f32(const float* x, int64_t ldx, const float* score, const int* perm, int64_t K, int F, int use_tanh,
                              float* out, int64_t ldo, tsgnn_stream_t stream);
int tsgnn_gather_gate_bwd_f32(const float* x, int64_t ldx, const float* score, const int* perm, int64_t K, int F, int use_tanh,
                              const float* dout, int64_t ldo, float* dx, int64_t lddx, float* dscore, tsgnn_stream_t stream);
/* PyG filter_adj (Code/sag/layers.py:23-24): relabel by perm, keep edges with both ends kept, original order.
 * mark -> (caller: exclusive scan of flag -> pos) -> compact. edge ids int64 as in PyG's edge_index. */
int tsgnn_filter_edges_mark(const int* perm, int64_t K, int64_t N, const int64_t* src, const int64_t* dst, int64_t E,
                            int* new_id, int* flag, tsgnn_stream_t stream);
int tsgnn_filter_edges_compact(const int64_t* src, const int64_t* dst, int64_t E, const int* new_id, const int* flag,
                               const int* pos, int64_t* out_src, int64_t* out_dst, int64_t* kept_eid, tsgnn_stream_t stream);
/* ---- sync-free SAGPool level (sagpool.hip): the filtered adjacency stays a CSR whose entry count lives on the device.
 * PyG gcn_norm for unit edge weights as per-row coefficients (GCNConv call sites Code/sag/network.py:19-23, layers.py:12):
 * dinv[i] = (deg_i + 1)^-1/2, self_w[i] = dinv[i]^2 (an existing self loop is kept instead: +0, self_w = 0). */
int tsgnn_gcn_coef_f32(const int* rowptr, const int* col, int64_t n_rows, float* dinv, float* self_w, tsgnn_stream_t stream);
/* out[i] = dinv[i] * sum_{j in N(i)} dinv[j] * x'[j] + self_w[i] * x'[i] (+ bias),  x' = relu_in ? relu(x) : x
 * = (D^-1/2 (A+I) D^-1/2 x')[i].  y (nullable) receives out; w_dot (nullable): t[i] = <out[i], w_dot> (+ *dot_bias) — the
 * GCNConv(C -> 1) score layer of SAGPool (layers.py:18) without materialising its input transform.  Symmetric A: the same
 * call is the backward (A^T = A). */
int tsgnn_gcn_propagate_f32(const int* rowptr, const int* col, const float* dinv, const float* self_w, const float* x,
                            int64_t ldx, int relu_in, const float* bias, const float* w_dot, const float* dot_bias, float* y,
                            int64_t ldy, float* t, int64_t n_rows, int feat, tsgnn_stream_t stream);
/* the same with explicit row ends: row r = entries [rowptr[r], rowend[r]) (rowend NULL = rowptr[r+1]) — the CSR the per-graph
 * pooling kernel writes keeps every graph's entries at its old segment base, with slack between graphs */
int tsgnn_gcn_propagate_re_f32(const int* rowptr, const int* rowend, const int* col, const float* dinv, const float* self_w,
                               const float* x, int64_t ldx, int relu_in, const float* bias, const float* w_dot, const float* dot_bias,
                               float* y, int64_t ldy, float* t, int64_t n_rows, int feat, tsgnn_stream_t stream);
/* Narrow inputs (feat <= 8; IMDB-B has one constant column): the aggregation agg = A^ x (as tsgnn_gcn_propagate_re_f32, kept for
 * the weight gradient) AND GCNConv's transform y = agg . w + bias (w [feat, n_out] row-major; network.py:34) in one launch — at
 * feat = 1 the transform is an outer product, not worth an MFMA launch of its own. */
int tsgnn_gcn_propagate_affine_f32(const int* rowptr, const int* rowend, const int* col, const float* dinv, const float* self_w,
                                   const float* x, int64_t ldx, float* agg, int64_t ldagg, int64_t n_rows, int feat, const float* w,
                                   int64_t ldw, const float* bias, float* y, int64_t ldy, int n_out, tsgnn_stream_t stream);
/* 1 when the fused level kernels below accept feature width F (F % 4 == 0, F <= 256) */
/* mean aggregation with the coefficients taken from the row lengths (rowend nullable as tsgnn_gcn_propagate_re_f32):
 *   transpose = 0:  y[i] = (1 / max(len_i, 1)) sum_{j in row i} x[j] (+ xself[i])     PyG SAGEConv's aggregation
 *   transpose = 1:  y[i] = sum_{j in row i} x[j] / max(len_j, 1)     (+ xself[i])     its adjoint on a symmetric edge list
 * xself nullable ([n_rows, >= feat]): added with weight 1 — the input gradient of lin_l(mean_j x_j) + lin_r(x_i) in one launch */
int tsgnn_propagate_mean_f32(const int* rowptr, const int* rowend, const int* col, int transpose, const float* x, int64_t ldx,
                             const float* xself, int64_t ldxs, float* y, int64_t ldy, int64_t n_rows, int feat, tsgnn_stream_t stream);
int tsgnn_sag_supported(int F);
/* kept rows (layers.py:21): xp[p,:] = relu?(y[perm[p],:]) * tanh(score[perm[p]]); cnt[p] = kept neighbours of perm[p].
 * y = xp = NULL: count only (the transposed adjacency of a non-symmetric graph). */
int tsgnn_sag_pool_gather_f32(const float* y, int64_t ldy, const float* score, const int* perm, const int* new_id,
                              const int* rowptr, const int* col, int64_t K, int F, int relu_in, float* xp, int64_t ldo, int* cnt,
                              tsgnn_stream_t stream);
/* One launch per level for everything between the conv output y (pre-activation) and the pooled rows, for graphs of at most
 * tsgnn_sag_pool_graph_max_nodes() nodes (one workgroup per graph, intermediate results in LDS): the score layer
 * score = A^ (relu(y) w_s) + b_s (layers.py:18), top-k with the relabelling map (perm, new_id as tsgnn_topk_segments_f32),
 * xp / cnt as tsgnn_sag_pool_gather_f32(relu_in = 1), out / arg as tsgnn_sag_readout_f32.  rowend (nullable): explicit row ends
 * of the input CSR.  rowptr_new .. self_w_new (all or none): filter_adj done here too — the pooled adjacency goes to col_new
 * from each graph's old segment base (rows [rowptr_new[p], rowend_new[p]), K entries each) with the next level's
 * tsgnn_gcn_coef_f32 output, so no scan / fill launches follow.  agg_next (nullable, needs the filter outputs): the NEXT
 * level's aggregation A^' xp = tsgnn_gcn_propagate_re_f32 on the pooled rows, formed here as well (same arithmetic). */
int tsgnn_sag_pool_graph_max_nodes(void);
int tsgnn_sag_pool_graph_f32(const float* y, int64_t ldy, const int* rowptr, const int* rowend, const int* col, const float* dinv,
                             const float* self_w, const float* w_s, const float* b_s, const int* graph_ptr, const int* graph_ptr_new,
                             int B, int max_seg, int F, float* score, int* perm, int* new_id, float* xp, int64_t ldo, int* cnt,
                             float* out, int64_t ldout, int* arg, int accumulate, int* rowptr_new, int* rowend_new, int* col_new,
                             float* dinv_new, float* self_w_new, float* agg_next, int64_t ldagg, tsgnn_stream_t stream);
/* out[b, :F] (+)= max over the rows of graph b, out[b, F:2F] (+)= their mean (gmp || gap, network.py:36,40,44);
 * arg[b, f] = row holding the max (ties -> smallest row) */
int tsgnn_sag_readout_f32(const float* xp, int64_t ld, const int* graph_ptr, int B, int F, int accumulate, float* out, int64_t ldo,
                          int* arg, tsgnn_stream_t stream);
/* filter_adj on CSR (layers.py:23-24): new row p = old row perm[p], entries = kept neighbours relabelled by new_id, original
 * order; rowptr_new = exclusive scan of tsgnn_sag_pool_gather_f32's cnt.  dinv_new/self_w_new (nullable pair): the next
 * level's tsgnn_gcn_coef_f32 output. */
int tsgnn_csr_filter_fill(const int* rowptr, const int* col, const int* perm, const int* new_id, int64_t K, const int* rowptr_new,
                          int* col_new, float* dinv_new, float* self_w_new, tsgnn_stream_t stream);
/* one-launch exclusive scan for short arrays (n <= 2^20): out[0..n], out[n] = total */
int tsgnn_scan_short_i32(const int* in, int64_t n, int* out, tsgnn_stream_t stream);
/* backward of gather + readouts per OLD row r (p = new_id[r]): kept: dtot = dxp[p] (nullable) + dread[b, F:2F] / k_b +
 * [arg[b,:] == p] dread[b, :F]; dyb[r] = dtot * gate; dscore[r] = (1 - gate^2) <dtot, relu?(y[r])>.  dropped: zeros. */
int tsgnn_sag_pool_bwd_f32(const float* y, int64_t ldy, const float* score, const int* new_id, const int* row_graph_new,
                           const int* graph_ptr_new, const int* arg, const float* dxp, int64_t lddxp, const float* dread,
                           int64_t lddr, int64_t N, int F, int relu_in, float* dyb, int64_t lddy, float* dscore,
                           tsgnn_stream_t stream);
/* tsgnn_sag_pool_bwd_f32 + tsgnn_sag_du_f32 as one workgroup per graph (graphs <= tsgnn_sag_pool_graph_max_nodes() nodes,
 * symmetric adjacency, rowend nullable): du[r] = gradient w.r.t. the pre-activation conv output y[r]; part: B rows of F + 4
 * floats; dws / dbs as tsgnn_sag_du_f32.  dagg_next (nullable; then dxp must be NULL): the gradient of the NEXT level's
 * aggregation with that level's CSR (rowptr_n, rowend_n, col_n) and coefficients — dxp = A^' dagg_next is formed per kept row
 * inside the kernel instead of by a tsgnn_gcn_propagate_re_f32 launch.  dws = dbs = NULL: the partial rows are left in `part` for
 * tsgnn_linear_wgrad_du_f32 (or tsgnn_sag_du_reduce_f32) to add up. */
int tsgnn_sag_pool_graph_bwd_f32(const float* y, int64_t ldy, const float* score, const int* new_id, const int* graph_ptr,
                                 const int* graph_ptr_new, const int* arg, const float* dxp, int64_t lddxp, const float* dread,
                                 int64_t lddr, const int* rowptr, const int* rowend, const int* col, const float* dinv,
                                 const float* self_w, const float* w_s, int B, int max_seg, int F, float* du, int64_t lddu, float* part,
                                 float* dws, float* dbs, const float* dagg_next, int64_t lddagg, const int* rowptr_n,
                                 const int* rowend_n, const int* col_n, const float* dinv_n, const float* self_w_n,
                                 tsgnn_stream_t stream);
/* fixed-order sum of nb partial rows part[nb][F + 4] (columns 0..F-1: dw_s, column F: db_s) -> dws[F], dbs[1] */
int tsgnn_sag_du_reduce_f32(float* part, int nb, int F, float* dws, float* dbs, tsgnn_stream_t stream);
/* tsgnn_linear_wgrad_f32 (dw != NULL) whose reduction launch carries that sum as one extra block (nb <= 256): the conv layer's
 * weight gradient and the score layer's wait for the same producer, so they share a launch */
int tsgnn_linear_wgrad_du_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                              int64_t rows_per_slab, float* ws, float* dw, float* db, float* part, int nb, int F_du, float* dws,
                              float* dbs, tsgnn_stream_t stream);
/* dyb[r] <- (dyb[r] + dt[r] * w_s) * [y[r] > 0] with dt = A^ dscore (score layer backward folded in);
 * dws = sum_r dt[r] * relu(y[r]), dbs = sum_r dscore[r] (fixed-order block partials in `part`: tsgnn_sag_du_blocks(N, F)
 * rows of F + 4 floats, summed by a second one-block launch). */
int tsgnn_sag_du_blocks(int64_t N, int F);
int tsgnn_sag_du_f32(const int* rowptr, const int* rowend, const int* col, const float* dinv, const float* self_w, const float* dscore, const float* y,
                     int64_t ldy, const float* w_s, float* dyb, int64_t lddy, int64_t N, int F, float* part, float* dws, float* dbs,
                     tsgnn_stream_t stream);
int tsgnn_relu_fwd_f32(const float* x, int64_t n, float* y, tsgnn_stream_t stream);
int tsgnn_relu_bwd_f32(const float* y, const float* dy, int64_t n, float* dx, tsgnn_stream_t stream);
/* nn.Softmax(dim=-1) over the assignment logits (encoders.py:369) */
int tsgnn_row_softmax_fwd_f32(const float* x, int64_t ldx, int64_t rows, int C, float* y, int64_t ldy, tsgnn_stream_t stream);
int tsgnn_row_softmax_bwd_f32(const float* y, int64_t ldy, const float* dy, int64_t lddy, int64_t rows, int C, float* dx,
                              int64_t lddx, tsgnn_stream_t stream);
/* the same with the embedding mask of encoders.py:370-371 folded in: rows >= zero_from (the ghost rows of a packed batch)
 * get y = 0 forward and dx = 0 backward */
int tsgnn_row_softmax_masked_fwd_f32(const float* x, int64_t ldx, int64_t rows, int C, float* y, int64_t ldy, int64_t zero_from,
                                     tsgnn_stream_t stream);
int tsgnn_row_softmax_masked_bwd_f32(const float* y, int64_t ldy, const float* dy, int64_t lddy, int64_t rows, int C, float* dx,
                                     int64_t lddx, int64_t zero_from, tsgnn_stream_t stream);
/* PyG dense_diff_pool's assignment (north_star operator; no call site in the reference): y = softmax(x, -1) * mask (mask:
 * one float per row, nullable) and hpart[ceil(rows / 4)] = partial sums (fixed order) of the rows' entropy terms -sum_k y log(y + eps);
 * the backward adds g_ent[0] * g_scale * d(sum of the entropy terms)/dy to the gradient arriving at y (ds, nullable; g_ent: device scalar,
 * nullable) and goes through the mask and the softmax. */
int tsgnn_row_softmax_ent_fwd_f32(const float* x, int64_t ldx, int64_t rows, int C, const float* mask, float eps, float* y, int64_t ldy,
                                  float* hpart, tsgnn_stream_t stream);
int tsgnn_row_softmax_ent_bwd_f32(const float* y, int64_t ldy, const float* ds, int64_t ldds, const float* mask, const float* g_ent,
                                  float g_scale, float eps, int64_t rows, int C, float* dx, int64_t lddx, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- fused slot kernels of the GraphSage stack (sage_fused.hip) */

/* 1 if (B graphs, F features) is covered by the fused slot kernels (B <= 128, F % 4 == 0, F <= 128) */
int tsgnn_slot_fused_supported(int B, int F);
/* y = slot_bn(relu(v)) with the statistics computed in the same pass (one workgroup per node slot keeps the slot's
 * rows of all graphs in registers).  Same result as tsgnn_bn_slots_fwd_f32 (encoders.py:179-181,134-138).
 * zero_ptr (nullable): zero_n 64-bit words cleared on the side (the stack's packed max-readout buffer). */
int tsgnn_slot_bn_fwd_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                          const float* v, int64_t ldv, int F, int relu, float* mean, float* rstd, float* y, int64_t ldy,
                          unsigned long long* zero_ptr, int64_t zero_n, tsgnn_stream_t stream);
/* One-pass backward of [max readout (dout, arg; both NULL: none) + next layer's dxs (nullable) + a gradient dxs2 (nullable)
 * that reaches this layer's output rows directly (node-level outputs)] -> slot BN -> ReLU -> row L2 normalise:
 * du = gradient w.r.t. the pre-normalise GraphConv output (feeds tsgnn_linear_wgrad_f32 / tsgnn_rowgemm_f32). */
int tsgnn_slot_post_bwd_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                            const float* v, int64_t ldv, const float* dxs, int64_t lddxs, const float* dxs2, int64_t lddxs2,
                            const float* dout, int64_t ldo, const int* arg, int F, int relu, int bn, const float* mean,
                            const float* rstd, const float* rinv, float* du, int64_t lddu, tsgnn_stream_t stream);
/* Layer 0 of a stack — the layer whose dU has no consumer but its own weight / bias gradient (the input features need no gradient):
 * tsgnn_slot_post_bwd_f32 (no dxs2) AND the slabs of tsgnn_linear_wgrad_f32 in ONE launch; the rows of dU never go to memory.
 * z [rows, K_in] (16-byte rows): the layer's aggregated input kept by the forward.  ws: nblocks slabs of (K_in + 1) * 128 floats
 * (nblocks <= nmax persistent workgroups walk the slots; sum them with tsgnn_wgrad_reduce_multi_f32, nslab = nblocks).
 * B <= 32, F = 128, K_in <= 128, n_ghost = nmax. */
int tsgnn_slot_post_wgrad_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                              const float* v, int64_t ldv, const float* dxs, int64_t lddxs, const float* dout, int64_t ldo,
                              const int* arg, int F, int relu, int bn, const float* mean, const float* rstd, const float* rinv,
                              const float* z, int64_t ldz, int K_in, float* ws, int nblocks, tsgnn_stream_t stream);
/* The same for the LAST layer of a stack (relu = bn = 0, nothing above it): du from the max-readout gradient alone, one lane
 * group per row instead of one workgroup per slot.  row_graph[n_real]: graph of every real row (>= B: padding row of a
 * capacity-padded batch, du = 0); rows [0, n_real + n_ghost_rows) of du are written; a ghost row n_real + n collects the
 * graphs with exactly n nodes, in graph order. */
int tsgnn_readout_l2_bwd_f32(const int* graph_ptr, const int* row_graph, int B, int64_t n_real, int n_ghost_rows, const float* v,
                             int64_t ldv, const float* dout, int64_t ldo, const int* arg, int F, const float* rinv, float* du,
                             int64_t lddu, tsgnn_stream_t stream);
/* max readout (encoders.py:183): partial maxima of one layer into packed[B*F] (zeroed by the caller), then ONE decode
 * for all L layers: out[b, l*Fh + f], arg in packed order (layers 0..L-2 are Fh wide, the last Fl). */
int tsgnn_readout_partial_f32(const int* graph_ptr, int B, int nmax, int64_t n_real, int n_ghost, const float* x, int64_t ldx, int F,
                              unsigned long long* packed, tsgnn_stream_t stream);
int tsgnn_readout_decode_layers_f32(const unsigned long long* packed, int B, int L, int Fh, int Fl, float* out, int64_t ldo, int* arg,
                                    tsgnn_stream_t stream);

/* ---------------------------------------------------------------- graph-level head (head.hip) */

/* vec = W1 out + b1 ; y = W2 vec + b2  — the two chained nn.Linear after the readout (pre_pred_model -> pred_model,
 * or map_model -> map2_model; encoders.py:207-217).  W1 [E,P], W2 [C,E] as nn.Linear stores them.  P % 4 == 0. */
int tsgnn_head2_fwd_f32(const float* out, int64_t ldo, const float* w1, const float* b1, const float* w2, const float* b2, int B, int P,
                        int E, int C, float* vec, float* y, tsgnn_stream_t stream);
/* Tail of the GraphSage stack in ONE launch (block b = graph b): decode layers 0..L-2 of the packed max readout (layout of
 * tsgnn_readout_partial_f32 / tsgnn_readout_decode_layers_f32), take the last layer's max readout by scanning graph b's rows
 * of v_last directly (its first ghost row stands for all of them: the last layer has no slot batch-norm), write
 * out[B, (L-1)*Fh + Fl] and arg, then the head vec = W1 out + b1, y = W2 vec + b2 (encoders.py:183-217). */
int tsgnn_readout_head_fwd_f32(const unsigned long long* packed, int B, int L, int Fh, int Fl, const float* v_last, int64_t ldv,
                               const int* graph_ptr, int64_t n_real, int nslots, int n_ghost, float* out, int64_t ldo, int* arg,
                               const float* w1, const float* b1, const float* w2, const float* b2, int E, int C, float* vec, float* y,
                               tsgnn_stream_t stream);
/* The same tail when the LAST layer's readout is already in packed too (tsgnn_sage_layer_fwd_ro_f32): decode all
 * (L-1)*Fh + Fl packed maxima of each graph into out / arg, then the head.  No pass over node rows.  E <= 128. */
int tsgnn_packed_head_fwd_f32(const unsigned long long* packed, int B, int L, int Fh, int Fl, float* out, int64_t ldo, int* arg,
                              const float* w1, const float* b1, const float* w2, const float* b2, int E, int C, float* vec, float* y,
                              tsgnn_stream_t stream);
/* tsgnn_packed_head_fwd_f32 that also does the step's housekeeping: every decoded entry of packed is zeroed (ready for the next
 * step's atomicMax) and clear[0 .. clear_n) 64-bit words are zeroed (the integer sums of the fused slot batch-norms: all
 * consumed by the launches before this one). */
int tsgnn_packed_head_fwd_z_f32(unsigned long long* packed, int B, int L, int Fh, int Fl, float* out, int64_t ldo, int* arg,
                                const float* w1, const float* b1, const float* w2, const float* b2, int E, int C, float* vec, float* y,
                                unsigned long long* clear, int64_t clear_n, tsgnn_stream_t stream);
/* backward in one launch: dvt = dvec (nullable) + W2^T dy (internal) ; dout = W1^T dvt ; dW1 = dvt^T out ; db1 ; dW2 = dy^T vec ; db2.
 * normparts (nullable, ceil(E/4) + 1 floats): per weight block, the sum of squares of the gradient entries it wrote. */
int tsgnn_head2_bwd_f32(const float* out, int64_t ldo, const float* vec, const float* dy, const float* dvec, const float* w1,
                        const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo, float* dw1, float* db1,
                        float* dw2, float* db2, float* normparts, tsgnn_stream_t stream);
/* The head whose launches also make the max readout of a LAST uniform level (N <= 64 rows per graph, rows b * N + n of z; DiffPool's
 * last pooled level, encoders.py:383,388-391).  Forward: columns [c0, c0 + F) of `out` (c0 + F == P) are FILLED with
 * max_n z[b * N + n, f] first, arg[b, f] = the winning row.  Backward: the row block of graph b also writes
 * dz[b * N + n, f] = (arg[b, f] == b * N + n) ? dout[b, c0 + f] : 0 for every row of the graph; dy NULL: the cross-entropy is folded
 * in as in tsgnn_head2_bwd_ce_f32 (ce_y = the logits, ce_label, ce_loss; all NULL otherwise). */
int tsgnn_head2_fwd_ro_f32(float* out, int64_t ldo, const float* w1, const float* b1, const float* w2, const float* b2, int B, int P,
                           int E, int C, float* vec, float* y, const float* z, int64_t ldz, int N, int c0, int F, int* arg,
                           tsgnn_stream_t stream);
int tsgnn_head2_bwd_ro_f32(const float* out, int64_t ldo, const float* vec, const float* dy, const float* dvec, const float* w1,
                           const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo, float* dw1, float* db1,
                           float* dw2, float* db2, float* normparts, const int* arg, int c0, int F, int N, float* dz, int64_t lddz,
                           const float* ce_y, const int64_t* ce_label, float* ce_loss, tsgnn_stream_t stream);
/* the same with F.cross_entropy (encoders.py:221-224) folded in: the gradient of mean softmax cross-entropy of the logits
 * y[B, C] w.r.t. them is rebuilt inside the kernel (B * C values per block in LDS) and the loss value is written to loss[0] */
int tsgnn_head2_bwd_ce_f32(const float* out, int64_t ldo, const float* vec, const float* y, const int64_t* label, float* loss,
                           const float* dvec, const float* w1, const float* w2, int B, int P, int E, int C, float* dout, int64_t lddo,
                           float* dw1, float* db1, float* dw2, float* db2, float* normparts, tsgnn_stream_t stream);

/* The same launch ALSO producing dU of the stack's LAST GraphConv layer (it has no batch-norm: dU is a row-wise function of the
 * readout gradient, what tsgnn_readout_l2_bwd_f32 computes in a launch of its own): extra workgroups (graph b, chunk of 64 or 128 rows)
 * rebuild the F-wide segment [seg_off, seg_off + F) of dout[b, :] themselves (same operands, same order, same bits) and do not
 * wait for anybody.  y != NULL: cross-entropy folded in (label, loss) — else dy is given.  v / rinv: the layer's output rows and
 * 1 / norm; arg [B, F]: its max-readout winners; du rows [0, n_real) are written, and du[n_real + b] = graph b's ghost-row
 * contribution (all ghost rows of this layer are identical, and only the SUM of the rows behind the real ones is ever used —
 * the bias gradient — so pass bias_only_rows = B downstream).  n_ghost_rows: ghost rows that exist; max_nodes = a bound on
 * the largest graph.  TSGNN_EUNSUPPORTED (nothing launched): fall back to tsgnn_head2_bwd*_f32 + tsgnn_readout_l2_bwd_f32. */
int tsgnn_head2_bwd_du_f32(const float* out, int64_t ldo, const float* vec, const float* y, const int64_t* label, float* loss,
                           const float* dy, const float* dvec, const float* w1, const float* w2, int B, int P, int E, int C, float* dout,
                           int64_t lddo, float* dw1, float* db1, float* dw2, float* db2, float* normparts, const int* graph_ptr,
                           int64_t n_real, int n_ghost_rows, int max_nodes, const float* v, int64_t ldv, const float* rinv, const int* arg,
                           int seg_off, int F, float* du, int64_t lddu, tsgnn_stream_t stream);
/* the same with the dU workgroups LISTED by the host: du_map[n_map] = graph << 8 | chunk for every chunk of du_chunk (64 or 128) rows
 * that holds rows — chunk 0 of EVERY graph (it also writes the graph's ghost contribution row); graph == B with chunks 0 .. k-1 for a
 * capacity-padded batch's padding rows (zero-filled by k workgroups) —, so that an exact batch launches no workgroup that only
 * returns and the launch's duration does not depend on where the batch's large graphs sit.  du_map NULL: the dense grid above
 * (n_map = 0), or — n_map < 0, B <= 64 — the same compact order resolved INSIDE the kernel from graph_ptr (a 64-lane prefix sum of
 * the graphs' chunk counts): for batches whose sizes live on the device (capacity-padded ingest batches). */
int tsgnn_head2_bwd_du_map_f32(const float* out, int64_t ldo, const float* vec, const float* y, const int64_t* label, float* loss,
                               const float* dy, const float* dvec, const float* w1, const float* w2, int B, int P, int E, int C,
                               float* dout, int64_t lddo, float* dw1, float* db1, float* dw2, float* db2, float* normparts,
                               const int* graph_ptr, int64_t n_real, int n_ghost_rows, int max_nodes, const float* v, int64_t ldv,
                               const float* rinv, const int* arg, int seg_off, int F, float* du, int64_t lddu, const int* du_map,
                               int n_map, int du_chunk, tsgnn_stream_t stream);

/* ---- DiffPool contraction of a pooled level (dense per-graph operands small enough for LDS), one workgroup per graph
 * (csrc/contract.hip).  Replaces the three bmm's of encoders.py:374-375 and their six backward products. */
int tsgnn_contract_dense_supported(int N, int K, int F);
/* X'[b] = S_b^T Z_b (xo [B,K,F]), A'[b] = S_b^T A_b S_b (ao [B,K,K]); t [B,K,N] = S^T A is kept for the backward.
 * s [B,N,K], z [B,N,F], adj [B,N,N], all contiguous. */
int tsgnn_contract_dense_fwd_f32(const float* s, const float* z, const float* adj, int B, int N, int K, int F, float* xo, float* ao,
                                 float* t, tsgnn_stream_t stream);
/* the same + the max readout of z over each graph's N rows (encoders.py:383) out of the staged operand: ro_out [B, F] (leading
 * dimension ro_ldo), ro_arg [B, F] = winning row b * N + n (what tsgnn_readout_max_fwd_f32 returns for the uniform batch (B, N)); ro_out
 * nullable.  s_out (nullable) given: `s` holds the assignment LOGITS, S = softmax over the K clusters of every row (nn.Softmax(dim=-1),
 * encoders.py:369) is formed on the staged operand, used for the products and written to s_out [B, N, K] */
int tsgnn_contract_dense_fwd_ro_f32(const float* s, const float* z, const float* adj, int B, int N, int K, int F, float* xo, float* ao,
                                    float* t, float* ro_out, int64_t ro_ldo, int* ro_arg, float* s_out, tsgnn_stream_t stream);
/* ds [B,N,K], dz [B,N,F], dadj [B,N,N] (each nullable) from dxo [B,K,F], dao [B,K,K] */
int tsgnn_contract_dense_bwd_f32(const float* s, const float* z, const float* adj, const float* t, const float* dxo, const float* dao,
                                 int B, int N, int K, int F, float* ds, float* dz, float* dadj, tsgnn_stream_t stream);
/* the same; dz additionally takes the gradient of the max readout of z over each graph's N rows — ro_dout [B, F] (leading dimension
 * ro_ldo), ro_arg [B, F] = winning row b * N + n or -1 — i.e. the pass tsgnn_readout_max_bwd_rows_f32 would make over dz afterwards
 * (the embeddings feed the readout AND the next contraction, encoders.py:383,374); ro_dout nullable.  softmax != 0: `s` is the softmax the
 * forward formed from the logits (s_out), and ds is returned as the gradient of the LOGITS, s * (dS - rowsum(s * dS)) */
int tsgnn_contract_dense_bwd_ro_f32(const float* s, const float* z, const float* adj, const float* t, const float* dxo, const float* dao,
                                    int B, int N, int K, int F, float* ds, float* dz, float* dadj, const float* ro_dout, int64_t ro_ldo,
                                    const int* ro_arg, int softmax, tsgnn_stream_t stream);

/* backward of the ROW-layout (level 1) contraction X'[b] = S_b^T Z_b, A'[b] = S_b^T (A S)_b (diffpool.py::_ContractRows) in one
 * launch: dZ = S dX', dS = Z dX'^T + (AS) dA'^T, d(AS) = S dA' for the rows of every slab (slab_row_ptr[nslab + 1]: at most 32
 * consecutive rows of ONE graph per slab; slab_graph[nslab]); rows [zero_from, zero_to) of the outputs are cleared. */
int tsgnn_contract_rows_bwd_supported(int K, int F);
int tsgnn_contract_rows_bwd_f32(const float* S, int64_t ldS, const float* Z, int64_t ldZ, const float* AS, int64_t ldAS,
                                const float* dxo, const float* dao, const int* slab_row_ptr, const int* slab_graph, int nslab, int K,
                                int F, float* dZ, int64_t lddZ, float* dS, int64_t lddS, float* dAS, int64_t lddAS,
                                int64_t zero_from, int64_t zero_to, tsgnn_stream_t stream);
/* the same; dZ additionally takes the gradient of the max readout of Z (ro_dout [B, F], ro_ldo, ro_arg [B, F] = winning row or -1);
 * winners outside the slabs' rows (ghost rows) are dropped */
int tsgnn_contract_rows_bwd_ro_f32(const float* S, int64_t ldS, const float* Z, int64_t ldZ, const float* AS, int64_t ldAS,
                                   const float* dxo, const float* dao, const int* slab_row_ptr, const int* slab_graph, int nslab, int K,
                                   int F, float* dZ, int64_t lddZ, float* dS, int64_t lddS, float* dAS, int64_t lddAS,
                                   int64_t zero_from, int64_t zero_to, const float* ro_dout, int64_t ro_ldo, const int* ro_arg,
                                   tsgnn_stream_t stream);

/* paired forms for two stacks that share batch and shapes (grid.y = 2; sage_stack._SageStackPair): tsgnn_slot_bn_fwd_f32 without
 * the readout-buffer clear, tsgnn_slot_post_bwd_f32 without a readout gradient, and the slab reduction of up to 8 sets described
 * in HOST memory, desc = [n, n x (ws, nslab, K, N, dw, db)] */
int tsgnn_slot_bn_fwd_pair_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                               const float* v0, const float* v1, int64_t ldv, int F, int relu, float* mean0, float* mean1, float* rstd0,
                               float* rstd1, float* y0, float* y1, int64_t ldy, tsgnn_stream_t stream);
int tsgnn_slot_post_bwd_pair_f32(const int* graph_ptr, const int* slot_count, int B, int nmax, int64_t n_real, int n_ghost,
                                 const float* v0, const float* v1, int64_t ldv, const float* dxs0, const float* dxs1, int64_t lddxs,
                                 const float* dxs2_0, const float* dxs2_1, int64_t lddxs2, int F, int relu, int bn, const float* mean0,
                                 const float* mean1, const float* rstd0, const float* rstd1, const float* rinv0, const float* rinv1,
                                 float* du0, float* du1, int64_t lddu, tsgnn_stream_t stream);
int tsgnn_wgrad_reduce_sets_f32(const int64_t* desc, tsgnn_stream_t stream);

/* ---- several independent problems of the GraphConv layer kernels in one launch (csrc/multi.hip): up to two weight-gradient slab
 * problems (arguments of tsgnn_linear_wgrad_f32, dw = db = NULL) and up to two gather products (arguments of
 * tsgnn_gather_rowgemm_f32), for the two 64-wide GCN stacks of DiffPool's first level (encoders.py:352-363).  desc (HOST memory):
 *   [ntn, ng,
 *    ntn x (z, ldz, du, lddu, rows, K_in, N, nslab, rows_per_slab, bias_only_rows, ws),
 *    ng  x (ell, ell_w, tail_ptr, tail_col, x, ldx, b, ldb, trans_b, bias, c, ldc, rinv, zout, ldz, rows, K, N, normalize, fill_rows)]
 * 32 < N <= 64, K, K_in <= 128, problems of one kind share every shape; TSGNN_EUNSUPPORTED otherwise. */
int tsgnn_sage_multi_tn_words(void);
int tsgnn_sage_multi_g_words(void);
int tsgnn_sage_multi_f32(const int64_t* desc, tsgnn_stream_t stream);
/* the same launch; zero0[0..n0) / zero1[0..n1) (nullable) are cleared by the filler block of product 0 / 1 AFTER it has written its
 * fill rows: the embedding mask of a stack that returns node features (ghost rows of the concatenation := 0, encoders.py:150-151,
 * 355-356) without a launch of its own.  Needs fill_rows > 0, n % 4 == 0, 16-byte aligned regions; TSGNN_EUNSUPPORTED otherwise. */
int tsgnn_sage_multi_zero_f32(const int64_t* desc, float* zero0, int64_t n0, float* zero1, int64_t n1, tsgnn_stream_t stream);

/* ---- backward of a GAT layer's packed projection hp = x W' (csrc/gat_products.hip; encoders_GAT.py:29-36): the slab partials of
 * dW'[K_in, N] = x[:, :K_in]^T du into ws (plan: tsgnn_wgrad_blocks_plan(rows, K_in, N, ldx, lddu); reduce:
 * tsgnn_wgrad_blocks_reduce_f32) and dx[rows, K_in] = du[rows, N] . wp[K_in, N]^T in ONE launch — the same blocks as
 * tsgnn_wgrad_blocks_f32's slab launch + tsgnn_rowgemm_f32(trans_b = 1), same bits.  128 < K_in <= 512, N <= 512, N % 4 == 0. */
int tsgnn_gat_bwd_products_f32(const float* x, int64_t ldx, const float* du, int64_t lddu, int64_t rows, int K_in, int N, const float* wp,
                               int64_t ldwp, float* dx, int64_t lddx, int nslab, int64_t rows_per_slab, float* ws, tsgnn_stream_t stream);
int tsgnn_wgrad_blocks_reduce_f32(const float* ws, int nslab, int K_in, int N, float* dw, int64_t lddw, tsgnn_stream_t stream);
/* K_in, N <= 128 (one 128 x 128 set: the slab layout of tsgnn_linear_wgrad_f32): the reduction of tsgnn_linear_wgrad_du_f32 alone
 * (part nullable) — the SAGPool conv layers (Code/sag/network.py:19-23) run their slabs beside dagg = du W^T in tsgnn_gat_bwd_products_f32 */
int tsgnn_linear_wgrad_du_reduce_f32(const float* ws, int nslab, int K_in, int N, float* dw, float* db, float* part, int nb, int F_du,
                                     float* dws, float* dbs, tsgnn_stream_t stream);
/* the same pairing for a torch.nn.Linear y = x W^T + b with W [N = out, K_in = in] (DiffPool's assignment predictor, encoders.py:362-372):
 * the slab partials of (dW^T, db) from x and dy into ws and dx[rows, K_in] = dy[rows, N] . W in one launch; the reduction into
 * nn.Linear's layout dw_oi[N][K_in] + db[N] (nullable).  K_in, N <= 512, both multiples of 4. */
int tsgnn_linear_bwd_products_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int K_in, int N, const float* w,
                                  int64_t ldw, float* dx, int64_t lddx, int nslab, int64_t rows_per_slab, float* ws, tsgnn_stream_t stream);
int tsgnn_wgrad_blocks_reduce_oi_f32(const float* ws, int nslab, int K_in, int N, float* dw_oi, int64_t lddw, float* db, tsgnn_stream_t stream);
/* two such reductions in one launch (both layers of a GAT encoder's backward); the slab-only form of the blocked weight gradient:
 * tsgnn_wgrad_blocks_slabs_f32 = tsgnn_wgrad_blocks_f32 without its reduction. */
int tsgnn_wgrad_blocks_reduce2_f32(const float* ws0, int nslab0, int K0, int N0, float* dw0, int64_t lddw0, const float* ws1, int nslab1,
                                   int K1, int N1, float* dw1, int64_t lddw1, tsgnn_stream_t stream);
int tsgnn_wgrad_blocks_slabs_f32(const float* z, int64_t ldz, const float* du, int64_t lddu, int64_t rows, int K_in, int N, int nslab,
                                 int64_t rows_per_slab, float* ws, tsgnn_stream_t stream);

/* ---- tail of the 2stg triplet step (csrc/triplet.hip; Code/sage+gat+diffpool/tripletnet.py:35-45): the three graphs' embeddings
 * embed[b] = W r[b] + bias (encoders.py:217 `map_model`, nn.Linear's [E, D] layout; r[3, D] = the concatenated readouts of anchor,
 * positive, negative) and dist = (||e_a - e_p + eps||_2, ||e_a - e_n + eps||_2) (F.pairwise_distance) in one launch; the backward
 * from the gradients of the two distances (d_dp[1], d_dn[1]) and of the three embeddings (d_ea, d_ep, d_en [E]; every one
 * nullable = zero) to d_r[3, D], dW[E, D], db[E] (nullable) in one launch.
 * D % 4 == 0, E <= 512, 16-byte aligned r / W rows. */
int tsgnn_triplet_embed_fwd_f32(const float* r, int64_t ldr, const float* w, int64_t ldw, const float* b, int D, int E, float eps,
                                float* embed, float* dist, tsgnn_stream_t stream);
int tsgnn_triplet_embed_bwd_f32(const float* r, int64_t ldr, const float* w, int64_t ldw, int D, int E, float eps, const float* embed,
                                const float* dist, const float* d_dp, const float* d_dn, const float* d_ea, const float* d_ep,
                                const float* d_en, float* d_r, int64_t lddr, float* dw, int64_t lddw, float* db, tsgnn_stream_t stream);

/* torch.nn.MarginRankingLoss(margin) of the triplet loop (Code/sage+gat+diffpool/train_triplet.py:235,277) in one launch:
 * loss[0] = mean (mean != 0) or sum over i of max(0, -target[i] (x1[i] - x2[i]) + margin); coef[n] = the gradient coefficients the
 * backward scales: dx1 = g[0] coef, dx2 = -g[0] coef (either nullable). */
int tsgnn_margin_rank_fwd_f32(const float* x1, const float* x2, const float* target, int64_t n, float margin, int mean, float* loss,
                              float* coef, tsgnn_stream_t stream);
int tsgnn_margin_rank_bwd_f32(const float* g, const float* coef, int64_t n, float* dx1, float* dx2, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- torch_geometric-named fused layers (csrc/sageconv.hip)
 * BASELINE.json's north_star names PyG's SAGEConv / GATConv / SAGPooling / dense_diff_pool; the reference itself never calls them
 * (no call site; PARITY UNPINNED — the oracle is oracle/pyg_ref.py's restatement of PyG's documented formulas).
 *
 * tsgnn_sage_conv_f32: ONE launch for
 *     out[i, :] = act( [ dst_scale[i] * sum_{j in N(i)} xg[j, :K]  ||  xs[i, :K] ] . [ W_l ; W_r ] + bias )
 *   forward of SAGEConv (lin_l(mean_j x_j) + lin_r(x_i)): xg = xs = x, dst_scale = 1 / max(deg, 1), the weights packed
 *     from lin_l.weight / lin_r.weight (nn.Linear's [N = out, K = in] layout: kn = 0), bias = lin_l.bias;
 *   its input gradient for a symmetric edge list, dx = A_mean^T (du W_l) + du W_r: xg = du scaled row-wise by 1 / max(deg, 1)
 *     (tsgnn_sage_relu_readout_bwd_f32 writes it beside du), xs = du, dst_scale = NULL, the same weight tensors packed as
 *     [K = out][N = in] (kn = 1);
 *   PyG GraphConv (sum aggregation): dst_scale NULL, xg = xs.
 * wl_pk / wr_pk: FRAGMENT-MAJOR copies of the two weight matrices (16,384 floats each, tsgnn_sage_conv_pack_f32): the matrix cores'
 * B operand is read with full-width, fully coalesced loads and never staged in LDS.
 * ell / tail: the fixed-width neighbour table of tsgnn_csr_to_ell (+ CSR tail).  K, N <= 128; xg / xs / zout rows 16-byte aligned with
 * at least ceil4(K) floats.  zout (nullable): the scaled aggregate [rows, K] (operand of the weight gradient).  relu_out: ReLU in the
 * epilogue; normalize: F.normalize(out, dim = -1) (rinv nullable: 1 / max(|row|, 1e-12)).  ro_packed / ro_sums [B, N] (nullable pair,
 * zero before the launch): per-graph column maxima of the output rows as packed (ordered value, ~row) atomicMax and column sums as
 * 64-bit fixed-point integers (2^-32 units; order-independent) — global_max_pool / global_mean_pool (network.py:36) without a pass
 * of their own; decode with tsgnn_sage_readout_decode_f32. */
int tsgnn_sage_conv_supported(int K, int N);
/* desc (HOST memory): [nsets <= 16, nsets x (w, ldw, K, N, kn, out)] — out[16384] = fragment-major copy of the K x N matrix w
 * (kn = 0: w[n * ldw + k], nn.Linear's [out, in]; kn = 1: w[k * ldw + n]), zero beyond K / N; one launch for all layers of a step */
int tsgnn_sage_conv_pack_f32(const int64_t* desc, tsgnn_stream_t stream);
/* post_h (nullable; input-gradient launches): the epilogue finishes the dU of the layer BELOW without a launch of its own — with v = the product, out = ( v + [post_arg[b, c] == r] post_dread[b, c] + post_dread[b, N + c] / n_b ) *
 * [post_h[r, c] > 0] and out2 (nullable) = out * post_row_scale[r] (see tsgnn_sage_relu_readout_bwd_f32; needs ro_row_graph /
 * ro_graph_ptr, excludes bias-free options relu_out / normalize / ro_packed). */
int tsgnn_sage_conv_f32(const int* ell, int ell_w, const int* tail_ptr, const int* tail_col, const float* xg, int64_t ldxg, const float* xs,
                        int64_t ldxs, const float* dst_scale, const float* wl_pk, const float* wr_pk,
                        const float* bias, float* out, int64_t ldo, float* zout, int64_t ldz, float* rinv,
                        int64_t rows, int K, int N, int relu_out, int normalize, unsigned long long* ro_packed,
                        unsigned long long* ro_sums, const int* ro_row_graph, const int* ro_graph_ptr, const float* post_h, int64_t post_ldh,
                        const float* post_dread, int64_t post_lddr, const int* post_arg, const float* post_row_scale, float* out2,
                        int64_t ldo2, tsgnn_stream_t stream);
/* slab partials of (z[:, :K]^T du, colsum du) into ws_l and of x[:, :K]^T du into ws_r ([nslab][K + 1][N] each, the layout of
 * tsgnn_linear_wgrad_f32's dw == NULL form; plan with tsgnn_linear_wgrad_plan): both weights of a SAGEConv layer in ONE launch */
int tsgnn_sage_wgrad_pair_f32(const float* z, int64_t ldz, const float* x, int64_t ldx, const float* du, int64_t lddu, int64_t rows, int K,
                              int N, int nslab, int64_t rows_per_slab, float* ws_l, float* ws_r, tsgnn_stream_t stream);
/* du[r, c] = ( dxs[r, c] (nullable) + [arg[b, c] == r] dread[b, c] + dread[b, F + c] / n_b ) * [h[r, c] > 0] (relu != 0), b = row_graph[r]:
 * backward of h = relu(u) feeding the next layer, the max readout (arg = winning rows) and the mean readout (dread [B, 2F], nullable);
 * dus (nullable) = du scaled row-wise by row_scale[r] (1 / deg: the rows the input-gradient launch gathers) */
int tsgnn_sage_relu_readout_bwd_f32(const float* h, int64_t ldh, const float* dxs, int64_t lddxs, const float* dread, int64_t lddr,
                                    const int* arg, const int* row_graph, const int* graph_ptr, int64_t rows, int F, int relu, float* du,
                                    int64_t lddu, const float* row_scale, float* dus, int64_t lddus, tsgnn_stream_t stream);
/* read[b, :F] = sum_l max_l[b, :], read[b, F:2F] = sum_l sum_l[b, :] / n_b (network.py:36-46), arg[l, b, f] = the row that holds layer l's
 * maximum; packed / sums [L, B, F] as left by tsgnn_sage_conv_f32's epilogue, zero again afterwards */
int tsgnn_sage_readout_decode_f32(unsigned long long* packed, unsigned long long* sums, const int* graph_ptr, int B, int L, int F, float* read,
                                  int64_t ldr, int* arg, tsgnn_stream_t stream);
/* desc (HOST memory): [nsets <= 12, nsets x (ws, nslab, K, N, dw_oi, lddw, db, n_db, tail, kn)]: slab sets of tsgnn_linear_wgrad_f32
 * (dw == NULL form) summed in slab order into nn.Linear's layout dw_oi[n * lddw + k] (kn = 1: GCNConv's [in, out], dw[k * lddw + n])
 * (+ db[n < n_db], nullable; n_db = N for a weight set): all layers' weight gradients in one launch.  K = 0: a set of partial ROWS [nslab][N] only, column sums to db[0 .. n_db)
 * and column n_db to tail[0] (nullable) — the SAGPool score layer's partial rows [nb][F + 4] of tsgnn_sag_pool_graph_bwd_f32
 * (Code/sag/layers.py:18 weight / bias gradients), i.e. tsgnn_sag_du_reduce_f32 riding in this launch.
 * normparts (nullable; tsgnn_sage_wgrad_reduce_oi_blocks(desc) entries): block k's sum of squares of what it wrote; step_state
 * (nullable): step_state[0] += 1 — both as tsgnn_wgrad_reduce_multi_f32, for tsgnn_adam_from_partials_f32 */
int tsgnn_sage_wgrad_reduce_oi_blocks(const int64_t* desc);
int tsgnn_sage_wgrad_reduce_oi_f32(const int64_t* desc, float* normparts, float* step_state, tsgnn_stream_t stream);
/* desc (HOST memory): [njobs <= 16, njobs x (src, lds, rows, cols, dst, ldd, dst_cols)]: dst[r, 0 .. dst_cols) = src[r, 0 .. cols) then
 * zeros — the [W_l | W_r] images of every level of a SAGEConv stack placed and zero-padded in ONE launch (host glue of the reference's
 * network with PyG SAGEConv layers; no reference line) */
int tsgnn_copy2d_multi_f32(const int64_t* desc, tsgnn_stream_t stream);

/* ---------------------------------------------------------------- torch_geometric GATConv as fused launches (csrc/gatconv.hip)
 * Per-TARGET edge softmax (standard GAT; the reference's own DGATHead normalises per column: tsgnn_gat_attn_*).  No call site in the
 * reference (PARITY UNPINNED).  hp = x W' [rows, >= C + 2H] with W' = [W^T | W_h^T att_r_h | W_h^T att_l_h | pad] (C = H * Co):
 * features, then the node's scalar as TARGET (hp[:, C + h]) and as SOURCE (hp[:, C + H + h]).  (rowptr, col): CSR grouped by target, self
 * loops included.  H in {1, 2, 4, 8}, Co in {4, 8, 16, 32, 64}, C <= 256. */
int tsgnn_gatconv_supported(int H, int Co);
/* y[i] = act( sum_j alpha_ij hp[j, :C] [mean over heads] + bias ), alpha_ij = softmax over the sources j of row i of
 * LeakyReLU(hp[i, C + h] + hp[j, C + H + h]); stat[rows, H, 2] (out): (max, 1 / sum of exponentials) of every row, by a first small launch */
int tsgnn_gatconv_fwd_f32(const float* hp, int64_t ldh, const int* rowptr, const int* col, int64_t rows, int H, int Co, float slope,
                          int mean_heads, int apply_elu, const float* bias, float* stat, float* y, int64_t ldy, tsgnn_stream_t stream);
/* backward, target side (one wave per row): dpre[rows, C] = dy * ELU'(y) [/ H per head] (out: the transposed pass gathers it; its column
 * sums are the bias gradient), dhp[:, C + h] = d s_dst, the zero pad of dhp, per-entry alpha / t1 / t2 [nnz, H] in A's entry order and
 * S [rows, H].  The source side is two existing launches over A^T: dhp[:, :C] = tsgnn_csr_spmm_heads_epi_f32(alpha through the entry
 * map, x = dpre) and dhp[:, C + H + h] = tsgnn_gat_score_rowsum_f32(t1, t2, S; column offset C + H). */
int tsgnn_gatconv_bwd_rows_f32(const float* hp, int64_t ldh, const float* y, int64_t ldy, const float* dy, int64_t lddy, const int* rowptr,
                               const int* col, int64_t rows, int H, int Co, float slope, int mean_heads, int apply_elu, const float* stat,
                               float* dpre, int64_t lddp, float* dhp, int Ns, float* alpha, float* t1, float* t2, float* S,
                               tsgnn_stream_t stream);
/* lin_l.weight [H * Co, Fin] (nn.Linear layout) + att_r / att_l [H * Co] -> W' [Fin, Ns] for up to 4 layers in ONE launch, and dW' -> their
 * gradients in one launch.  desc (HOST memory): [L, L x (H, Fin, Co, Ns, w, ldw, att_r, att_l, wp (W' out / dW' in), gw, gar, gal)] */
int tsgnn_gatconv_pack_desc_words(void);
int tsgnn_gatconv_pack_f32(const int64_t* desc, tsgnn_stream_t stream);
int tsgnn_gatconv_unpack_f32(const int64_t* desc, tsgnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TSGNN_H */
