// GCN stack of a pooled DiffPool level (encoders.py:378-380 -> gcn_forward :140-167 on the dense, differentiable adjacency
// A' = S^T A S of :375) as ONE launch forward and ONE launch backward, for up to two stacks that share (x, adj) — the level's
// embedding stack and the next level's assignment stack.
//
// A pooled level is tiny (DD, 16 graphs: 64 nodes x 192 features, then 8 nodes): the layer-by-layer form spends its time on
// launch boundaries — per layer a batched A.x product, the transform + L2 normalise, the slot batch-norm; backward five more
// (24 launches per stack, ~80 per step for the two pooled levels of BASELINE config 5).  Here a workgroup owns 16 rows of one
// graph of one stack and walks all layers; the only thing that couples graphs is the per-slot batch-norm (apply_bn,
// encoders.py:134-138: channel = node slot, statistics over batch and features), so workgroups meet at a device-wide barrier
// once per hidden layer forward (per-row partial sums -> every workgroup finishes the statistics of all slots itself) and
// twice per layer backward (BN partials; the rows of d(A x) the other tiles of the graph produced).  The grid is at most a few
// hundred workgroups (all resident: it runs alone on its stream position inside the step's hipGraph), the barrier is an
// arrival word per barrier with a BOUNDED spin: a workgroup that gives up raises `err[0]`, which the host checks at its next
// synchronisation (message_passing.check_device_errors) instead of hanging the device.
//
// Every product is a 16-row tile on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate) with both operands in LDS: with one wave
// per SIMD nothing hides LDS latency, so what counts is instructions and LDS requests per flop (a first version on scalar FMAs
// spent 14 us in one 16 x 192 x 64 tile product; scripts/trace_dense_stack.hip).
#include "common.h"
#include <utility>
#include "../../include/tsgnn.h"

// developer timeline stamps (scripts/trace_dense_stack.hip): forward by default, backward with -DTSGNN_TRACE_BWD
#ifdef TSGNN_TRACE_BWD
#define TRF(x) do { } while (0)
#define TRB(x) TR(x)
#else
#define TRF(x) TR(x)
#define TRB(x) do { } while (0)
#endif

namespace {

constexpr int DS_TR = 16;                 // rows per tile
#ifndef TSGNN_DS_THREADS
#define TSGNN_DS_THREADS 512
#endif
// Threads of a workgroup.  The first 256 ("main": thread (r, cg) = row r of the tile, columns 4 cg + 64 j + q) run the epilogues
// and own the register tiles; with 512, waves 4-7 are HELPERS that take part in what parallelises freely — the fills, the zeroing,
// the tile products (column groups dealt over eight waves), the dW tiles, the final sums — so that two waves per SIMD interleave
// where one wave per SIMD was bound by instruction issue and LDS round trips (scripts/trace_dense_stack.hip).
constexpr int DS_NT = TSGNN_DS_THREADS;
constexpr int DS_NW = DS_NT / 64;
static_assert(DS_NT == 256 || DS_NT == 512, "main threads are the first 256");
constexpr int DS_MAXL = 4;
constexpr int DS_BIG = 15360;             // floats of the big LDS region: [K][finP + 16], [fin][nP + 16], [n][finP + 16], [fin][80] (P: padded to 64)
constexpr float DS_NORM_EPS = 1e-12f;
constexpr float DS_BN_EPS = 1e-5f;

struct DsLayer {
  const float* w; int64_t ldw; const float* bias; int fin, n, off;
  float* agg;                 // [R, fin]   A.x, kept for dW
  float* v;                   // [R, n]     normalised pre-activation (hidden layers)
  float* rinv;                // [R]
  float* mean; float* rstd;   // [K]        hidden layers
  float* dw; float* db;       // gradients (backward)
  int64_t slab_off;           // offset of this layer's (fin + 1) * n partials inside a workgroup's slab
};
struct DsStack { float* out; int64_t ldo; const float* dout; int64_t lddo; int L; DsLayer layer[DS_MAXL]; };
struct DsArgs {
  const float* x; int64_t ldx; int fin0; const float* adj; int B, K, nstack;
  DsStack st[2];
  float* stats;               // [DS_MAXL][nstack * R][2]  per-row partial sums
  float* dagg;                // backward: [nstack][R][finmax] rows of d(A x) for the sibling tiles
  float* dxn;                 // backward: [nstack][R][finmax] gradient handed to the layer below / the stack's dx
  float* slabs;               // backward: [workgroups][slab_floats]
  int64_t slab_floats; int finmax;
  float* dx; int64_t lddx; float* dadj;   // outputs (nullable)
  float* dadj_part;           // [nstack][B*K*K] when two stacks contribute
  const float* dadj_add;      // nullable [B*K*K]: the gradient ANOTHER consumer of the same adjacency already produced (the next
                              // level's contraction, through the node's adjacency pass-through), summed into dadj here
  unsigned* sync; float* err;   // sync: 32 + 256 words, zero before the first launch (barrier words, sign-off counter, per-graph words)
};

// Device-wide barrier number k of a launch: its own arrival word (words[k], zero before the launch), bounded spin.  Word k - 1 is
// re-armed by workgroup 0 once barrier k has completed (every workgroup has left barrier k - 1 by then); the last barrier's word
// by whichever workgroup signs off last (grid_finish).  Launches of different grid sizes can therefore share the words.
//
// LIGHT: for exchanges whose data travels in agent-scope ATOMIC stores / loads only (ds_put2 / ds_get2 below: the per-row
// batch-norm partial sums, 8 bytes a row).  Those accesses are coherent across the XCDs by themselves (sc1), so the barrier needs
// neither the L2 write-back of a release at agent scope nor the L2 invalidate of an acquire — which is what a device-wide barrier
// costs here (the kernel's own dirty lines flushed, the weights re-fetched afterwards).  What it DOES need is that every thread's
// own ds_put2 stores have been acknowledged before the workgroup's arrival is issued: the stores and the arrival go to different
// L2 channels, which are not ordered against each other, and on gfx950 __syncthreads() does NOT wait for outstanding global
// stores (the ISA showed global_store -> s_barrier -> global_atomic_add with no vmcnt wait: ADVICE r2).  drain_stores() is that
// wait — `s_waitcnt vmcnt(0)` by every thread ahead of the s_barrier, no cache maintenance.  The arrival and the spin are relaxed.
// Nothing else written before a LIGHT barrier may be read by another workgroup after it.
// The full barriers drain too: their agent-scope release (buffer_wbl2 sc1 + s_waitcnt vmcnt(0)) is executed by thread 0 ALONE and
// waits for thread 0's wave's stores only; the write-back covers the other waves' data once it has reached the L2, which is what
// their own vmcnt(0) ahead of the s_barrier guarantees (hip's grid.sync() has every thread fence for the same reason).
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <bool LIGHT = false>
__device__ __forceinline__ void grid_barrier(unsigned* words, int& k, unsigned nblocks, float* err) {
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (LIGHT) __hip_atomic_fetch_add(words + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(words + k, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = false;
    // spin on RELAXED loads, then ONE acquire fence: an acquire load invalidates the caches at every iteration of the spin
    for (int spin = 0; spin < (1 << 21); ++spin) {
      if (__hip_atomic_load(words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nblocks) { ok = true; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!LIGHT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok) __hip_atomic_store(err, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k > 0 && blockIdx.x == 0) __hip_atomic_store(words + k - 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ++k;
  __syncthreads();
}
// The same barrier in two halves — arrive, (work that does not depend on the others), wait — so that a phase's operands which are
// forward data (not produced by this launch) are staged into LDS while the arrivals travel.
template <bool LIGHT = false>
__device__ __forceinline__ void grid_arrive(unsigned* words, int k) {
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (LIGHT) __hip_atomic_fetch_add(words + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(words + k, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <bool LIGHT = false>
__device__ __forceinline__ void grid_wait(unsigned* words, int& k, unsigned nblocks, float* err) {
  if (threadIdx.x == 0) {
    bool ok = false;
    for (int spin = 0; spin < (1 << 21); ++spin) {
      if (__hip_atomic_load(words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nblocks) { ok = true; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!LIGHT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok) __hip_atomic_store(err, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k > 0 && blockIdx.x == 0) __hip_atomic_store(words + k - 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  ++k;
  __syncthreads();
}
// Barrier among the `n` workgroups that own the tiles of ONE graph (the exchange of dagg rows before dx = A^T dagg concerns nobody
// else): its own arrival word (sync[32 + stack * B + graph]), counted up through the launch — the j-th use waits for n * j —
// and zeroed by the graph's first tile after the launch's last device-wide barrier.  Four arrivals instead of the whole grid's.
__device__ __forceinline__ void group_barrier(unsigned* word, unsigned target, float* err) {
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = false;
    for (int spin = 0; spin < (1 << 21); ++spin) {
      if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok) __hip_atomic_store(err, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}
// a pair of floats through agent-scope atomics (one 8-byte access): see grid_barrier<LIGHT>
__device__ __forceinline__ void ds_put2(float* p, float x, float y) {
  const unsigned long long bits = ((unsigned long long)__float_as_uint(y) << 32) | (unsigned long long)__float_as_uint(x);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ds_get2(const float* p) {
  const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)(bits & 0xffffffffull)), __uint_as_float((unsigned)(bits >> 32)));
}
// after a workgroup's last barrier: sign off; the last one re-arms the last barrier's word and the sign-off counter (no waiting)
__device__ __forceinline__ void grid_finish(unsigned* words, int k, unsigned nblocks) {
  if (threadIdx.x == 0 && k > 0) {
    const unsigned prev = __hip_atomic_fetch_add(words + 31, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == nblocks - 1) {
      __hip_atomic_store(words + k - 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(words + 31, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// x / d for 0 <= x < 2^22, 0 < d <= 4096 without the ~35-instruction integer division sequence: a float reciprocal and one
// correction step (exact).  The fills compute a (row, column) pair per 16-byte element; with `idx / c4n` they were bound by the
// address arithmetic, not by memory (3-4 us a fill at one wave per SIMD: scripts/trace_dense_stack.hip).
struct FastDiv {
  float inv; int d;
  __device__ __forceinline__ explicit FastDiv(int d_) : inv(1.0f / (float)d_), d(d_) {}
  __device__ __forceinline__ int operator()(int x) const {
    int q = (int)(((float)x + 0.5f) * inv);
    const int r = x - q * d;
    q += (r >= d) - (r < 0);
    return q;
  }
};

// rows x cols floats from global (row stride ldg) into LDS through `f(value, row, col)`; element (r, c) lands at
// lds[r * lds_ld + c], or at lds[c * lds_ld + r] when TRANSPOSE.  Loads are issued eight 16-byte requests deep before anything is
// stored (a plain `lds[i] = g[i]` loop is one dependent L2 round trip per iteration).  cols % 4 == 0, ldg % 4 == 0, g 16-byte aligned.
template <bool TRANSPOSE, typename F>
__device__ __forceinline__ void fill_lds(float* lds, int lds_ld, const float* __restrict__ g, int64_t ldg, int rows, int cols, F f) {
  const int tid = threadIdx.x;
  const int c4n = cols >> 2, total = rows * c4n;
  const FastDiv by_c4n(c4n);
  for (int base = tid; base < total; base += 8 * DS_NT) {
    float4 v[8];
    int rr[8], cc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * DS_NT;
      rr[u] = by_c4n(idx); cc[u] = 4 * (idx - rr[u] * c4n);
      v[u] = idx < total ? *reinterpret_cast<const float4*>(g + (int64_t)rr[u] * ldg + cc[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (base + u * DS_NT < total) {
        const float o0 = f(v[u].x, rr[u], cc[u]), o1 = f(v[u].y, rr[u], cc[u] + 1), o2 = f(v[u].z, rr[u], cc[u] + 2), o3 = f(v[u].w, rr[u], cc[u] + 3);
        if (TRANSPOSE) {
          lds[(cc[u] + 0) * lds_ld + rr[u]] = o0; lds[(cc[u] + 1) * lds_ld + rr[u]] = o1;
          lds[(cc[u] + 2) * lds_ld + rr[u]] = o2; lds[(cc[u] + 3) * lds_ld + rr[u]] = o3;
        } else {
          *reinterpret_cast<float4*>(lds + rr[u] * lds_ld + cc[u]) = make_float4(o0, o1, o2, o3);
        }
      }
    }
  }
}
struct Ident { __device__ __forceinline__ float operator()(float v, int, int) const { return v; } };

// fill_lds in two halves: pf_load requests a matrix (rows x cols floats, <= NV * 1024) into registers, pf_store writes it to LDS
// later.  A phase's operand that does not depend on the phases before it (the weights, forward data read by the backward) is
// requested BEFORE them — its L2 round trip runs under their MFMA chains and barriers instead of after them; the LDS region it
// lands in is free by the time it is stored.  Requests are unconditional from clamped addresses.
template <int NV>
struct Pf { float4 v[NV]; };
// (index_sequence instead of a loop: every register of the set is named at compile time; a loop the compiler leaves rolled would
// put the whole set into scratch)
template <typename F, int... I>
__device__ __forceinline__ void pf_each(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int NV>
__device__ __forceinline__ void pf_load(Pf<NV>& p, const float* __restrict__ g, int64_t ldg, int rows, int cols) {
  const int c4n = cols >> 2, total = rows * c4n;
  const FastDiv by_c4n(c4n);
  pf_each([&](auto u) {
    const int idx = min(u.value * DS_NT + (int)threadIdx.x, total - 1);
    const int rr = by_c4n(idx), cc = 4 * (idx - rr * c4n);
    p.v[u.value] = *reinterpret_cast<const float4*>(g + (int64_t)rr * ldg + cc);
  }, std::make_integer_sequence<int, NV>{});
}
template <int NV, bool TRANSPOSE>
__device__ __forceinline__ void pf_store(float* lds, int lds_ld, const Pf<NV>& p, int rows, int cols) {
  const int c4n = cols >> 2, total = rows * c4n;
  const FastDiv by_c4n(c4n);
  pf_each([&](auto u) {
    const int idx = u.value * DS_NT + (int)threadIdx.x;
    if (idx < total) {
      const int rr = by_c4n(idx), cc = 4 * (idx - rr * c4n);
      const float4 v = p.v[u.value];
      if (TRANSPOSE) {
        lds[(cc + 0) * lds_ld + rr] = v.x; lds[(cc + 1) * lds_ld + rr] = v.y;
        lds[(cc + 2) * lds_ld + rr] = v.z; lds[(cc + 3) * lds_ld + rr] = v.w;
      } else {
        *reinterpret_cast<float4*>(lds + rr * lds_ld + cc) = v;
      }
    }
  }, std::make_integer_sequence<int, NV>{});
}
constexpr int DS_PFW = (15360 / 4 + DS_NT - 1) / DS_NT;   // float4 per thread of a prefetched matrix of up to DS_BIG floats
__device__ __forceinline__ void zero_lds(float* lds, int n) {
  for (int i = threadIdx.x; i < n; i += DS_NT) lds[i] = 0.f;
}
__device__ __forceinline__ int pad64(int n) { return (n + 63) & ~63; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// C[16][16 ng] = A[16 x depth] . Bm[depth x 16 ng] on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate), operands and result in LDS:
//   A(i, k) = a[i * ars + k * acs],  Bm(k, n) = bm[k * ldb + n],  C(i, n) -> c[i * ldc + n].
// Wave w takes the 16-column groups w, w + 4, ...; lane l supplies A(l % 16, k0 + l / 16) and Bm(k0 + l / 16, 16 g + l % 16) per
// step of four k and holds C(4 (l / 16) + v, 16 g + l % 16) in accumulator register v.  Sixteen steps' operands are requested
// before their MFMA chain issues.  depth % 4 == 0; ldb % 32 == 16 keeps the four k-rows of a B request on disjoint bank halves.
// All threads of the workgroup must call it (barrier-free inside; the caller synchronises before and after).
__device__ __forceinline__ void tile_mfma(float* c, int ldc, const float* a, int ars, int acs, const float* bm, int ldb, int depth, int ng) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int steps = depth >> 2;
  for (int g = w; g < ng; g += DS_NW) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if ((steps & 15) == 0) {
      // whole rounds of sixteen steps (depth 64, 192: every product of the DD levels with 64 nodes): reads without predicates in
      // large basic blocks, the next round's operands requested before the current chain issues — with predicates every read
      // was waited for on its own (same fix and measurement as contract.hip::ct_mfma: 144 -> 92 cycles per MFMA)
      const float* ap = a + i * ars + kq * acs;
      const float* bp = bm + kq * ldb + 16 * g + i;
      auto load16 = [&](int s0, float (&af)[16], float (&bf)[16]) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { af[u] = ap[4 * (s0 + u) * acs]; bf[u] = bp[4 * (s0 + u) * ldb]; }
      };
      float a0[16], b0[16], a1[16], b1[16];
      load16(0, a0, b0);
      for (int s0 = 0;; s0 += 32) {
        const bool more1 = s0 + 16 < steps;
        load16(more1 ? s0 + 16 : 0, a1, b1);               // (at the end: a harmless re-read)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], acc, 0, 0, 0);
        if (!more1) break;
        const bool more2 = s0 + 32 < steps;
        load16(more2 ? s0 + 32 : 0, a0, b0);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b1[u], acc, 0, 0, 0);
        if (!more2) break;
      }
    } else
    for (int s0 = 0; s0 < steps; s0 += 16) {
      float af[16], bf[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = 4 * (s0 + u) + kq;
        const bool ok = s0 + u < steps;
        af[u] = ok ? a[i * ars + k * acs] : 0.f;
        bf[u] = ok ? bm[k * ldb + 16 * g + i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u], bf[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) c[(4 * kq + v) * ldc + 16 * g + i] = acc[v];
  }
}
__device__ __forceinline__ void zero_acc(float (&acc)[3][4]) {
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
}
// this thread's columns (4 cg + 64 j + q) of row r of an LDS tile
__device__ __forceinline__ void read_tile(float (&acc)[3][4], const float* tile, int ld, int r, int cg, int J) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float4 v = j < J ? *reinterpret_cast<const float4*>(tile + r * ld + 4 * cg + 64 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[j][0] = v.x; acc[j][1] = v.y; acc[j][2] = v.z; acc[j][3] = v.w;
  }
}
// per-slot totals of the per-row partial sums of all graphs (same order in every workgroup: the same bits)
// (fetching the B x K entries with all 256 threads into LDS first and summing from there was slower: 5.4 k cycles against 3.7 k
// for the 64 threads that walk their slot's graphs — the relaxed atomic loads are L2 hits of ~230 cycles each)
__device__ __forceinline__ void slot_totals(const float* st, int B, int K, int n, float& t1, float& t2) {
  t1 = 0.f; t2 = 0.f;
  int bb = 0;
  for (; bb + 8 <= B; bb += 8) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ds_get2(st + ((int64_t)(bb + u) * K + n) * 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) { t1 += v[u].x; t2 += v[u].y; }
  }
  for (; bb < B; ++bb) { const float2 v = ds_get2(st + ((int64_t)bb * K + n) * 2); t1 += v.x; t2 += v.y; }
}

constexpr int DS_LDA = 68;                // row stride of the A tile [TR][K <= 64] (+4: the 16 rows of an A request spread over the banks)
constexpr int DS_PADB = 16;               // B-operand matrices have row stride (columns padded to 64) + 16

// ------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(DS_NT) __attribute__((amdgpu_waves_per_eu(DS_NW / 4, DS_NW / 4))) void dense_stack_fwd_kernel(DsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* big = smem;                        // xin [K][finP + 16]  then  W [fin][nP + 16]
  float* At = big + DS_BIG;                 // [TR][DS_LDA]
  float* aggt = At + DS_TR * DS_LDA;        // [TR][finP + 4]
  float* ut = aggt + DS_TR * 196;           // [TR][nP]
  float* mu = ut + DS_TR * 148;             // [K]
  float* rs = mu + 64;                      // [K]
  float* big2 = rs + 64;                    // W [fin][nP + 16]: its own region, so that the next layer's weights land under the barrier
  const int tid = threadIdx.x, r = (tid & 255) >> 4, cg = tid & 15;
  const bool main_t = tid < 256;                             // (helper waves shadow a main thread's addresses and store nothing)
  const int K = a.K, B = a.B, R = B * K;
  const int tiles = (K + DS_TR - 1) / DS_TR;
  const unsigned nblocks = gridDim.x;
  const int t = blockIdx.x % tiles, b = (blockIdx.x / tiles) % B, s = blockIdx.x / (tiles * B);
  const DsStack& S = a.st[s];
  const int r0 = t * DS_TR, nrows = min(DS_TR, K - r0);
  const int64_t row = (int64_t)b * K + r0 + min(r, nrows - 1);     // (threads of missing rows shadow the tile's last row: no stores)
  const bool rok = main_t && r < nrows;
  int bar = 0;
  TRF(0);
  Pf<DS_PFW> wpf;                                            // the layer's weights, on their way while the phases before their product run
  pf_load(wpf, S.layer[0].w, S.layer[0].ldw, S.layer[0].fin, S.layer[0].n);
  zero_lds(At, DS_TR * DS_LDA);
  {
    const DsLayer& L0 = S.layer[0];
    if (pad64(L0.n) != L0.n) zero_lds(big2, L0.fin * (pad64(L0.n) + DS_PADB));
  }
  __syncthreads();
  fill_lds<false>(At, DS_LDA, a.adj + ((int64_t)b * K + r0) * K, K, nrows, K, Ident());
  pf_store<DS_PFW, false>(big2, pad64(S.layer[0].n) + DS_PADB, wpf, S.layer[0].fin, S.layer[0].n);
  if (S.L > 1) pf_load(wpf, S.layer[1].w, S.layer[1].ldw, S.layer[1].fin, S.layer[1].n);       // (stored under layer 0's barrier)
  float vprev[3][4];                                         // tiles == 1: this thread's entries of the layer below's v
  zero_acc(vprev);
  for (int l = 0; l < S.L; ++l) {
    const DsLayer& Ly = S.layer[l];
    const int fin = Ly.fin, N = Ly.n, finP = pad64(fin), NP = pad64(N), Jf = finP >> 6, Jn = NP >> 6;
    const int ldx = finP + DS_PADB, ldw = NP + DS_PADB, lda = finP + 4;   // lda: A-operand tiles (16 rows of a request on distinct banks)
    const bool last = l == S.L - 1;
    // (1) the layer's input rows of graph b -> LDS: x, or BN(ReLU(v)) of the layer below with the statistics just finished
    if (finP != fin) zero_lds(big, K * ldx);
    __syncthreads();
    if (l == 0) {
      fill_lds<false>(big, ldx, a.x + (int64_t)b * K * a.ldx, a.ldx, K, fin, Ident());
    } else if (tiles == 1) {
      // the graph is this tile: its rows of v are still in this thread's registers (no trip through memory)
      const DsLayer& Lp = S.layer[l - 1];
      float* outp = S.out + row * S.ldo + Lp.off;
      if (rok) {
        const float m = mu[r0 + r], q_ = rs[r0 + r];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int c = 4 * cg + 64 * j;
          if (j < Jf && c < fin) {
            const float4 y = make_float4((fmaxf(vprev[j][0], 0.f) - m) * q_, (fmaxf(vprev[j][1], 0.f) - m) * q_,
                                         (fmaxf(vprev[j][2], 0.f) - m) * q_, (fmaxf(vprev[j][3], 0.f) - m) * q_);
            *reinterpret_cast<float4*>(big + (r0 + r) * ldx + c) = y;
            *reinterpret_cast<float4*>(outp + c) = y;
          }
        }
      }
    } else {
      const DsLayer& Lp = S.layer[l - 1];
      float* outp = S.out + (int64_t)b * K * S.ldo + Lp.off;
      const int64_t ldo = S.ldo;
      fill_lds<false>(big, ldx, Lp.v + (int64_t)b * K * fin, fin, K, fin, [&](float v, int m, int c) {
        const float y = (fmaxf(v, 0.f) - mu[m]) * rs[m];
        if (m >= r0 && m < r0 + nrows) outp[(int64_t)m * ldo + c] = y;                             // this tile's rows of the result
        return y;
      });
    }
    __syncthreads();
    if (l == 0) TRF(1);
    // (2) agg tile = A[tile rows, :] . xin   (rows beyond the tile are zero rows of At)
    tile_mfma(aggt, lda, At, DS_LDA, 1, big, ldx, K, finP >> 4);
    __syncthreads();
    float acc[3][4];
    read_tile(acc, aggt, lda, r, cg, Jf);
    if (rok) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = 4 * cg + 64 * j;
        if (j < Jf && c < fin) *reinterpret_cast<float4*>(Ly.agg + row * fin + c) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
      }
    }
    if (l == 0) TRF(2);
    // (3) u = agg . W + bias (W went to big2 under the previous barrier / at the start), row L2 normalise
    if (l == 0) TRF(3);
    // the bias, requested before the product from clamped addresses (per-element `bias ? bias[c] : 0` loads were waited for one
    // by one); no bias: the same requests go to the weights and are discarded
    float4 bias4[3];
    {
      const float* bsrc = Ly.bias ? Ly.bias : Ly.w;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = 4 * cg + 64 * j;
        bias4[j] = *reinterpret_cast<const float4*>(bsrc + ((j < Jn && c < N) ? c : 0));
        if (!Ly.bias) bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    tile_mfma(ut, NP, aggt, lda, 1, big2, ldw, fin, NP >> 4);
    __syncthreads();
    float u[3][4];
    read_tile(u, ut, NP, r, cg, Jn);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float bq[4] = {bias4[j].x, bias4[j].y, bias4[j].z, bias4[j].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * cg + 64 * j + q;
        if (j < Jn && c < N) { u[j][q] += bq[q]; ss = fmaf(u[j][q], u[j][q], ss); } else u[j][q] = 0.f;
      }
    }
    if (l == 0) TRF(4);
    ss = row16_sum(ss);                                    // the 16 lanes of a row are one DPP row
    const float ri = fminf(__builtin_amdgcn_rsqf(ss), 1.0f / DS_NORM_EPS);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = 4 * cg + 64 * j;
      const float4 v = make_float4(u[j][0] * ri, u[j][1] * ri, u[j][2] * ri, u[j][3] * ri);
      vprev[j][0] = v.x; vprev[j][1] = v.y; vprev[j][2] = v.z; vprev[j][3] = v.w;
      if (j < Jn && c < N) {
        if (rok) {
          if (last) *reinterpret_cast<float4*>(S.out + row * S.ldo + Ly.off + c) = v;
          else *reinterpret_cast<float4*>(Ly.v + row * N + c) = v;
        }
        const float p0 = fmaxf(v.x, 0.f), p1 = fmaxf(v.y, 0.f), p2 = fmaxf(v.z, 0.f), p3 = fmaxf(v.w, 0.f);
        s1 += (p0 + p1) + (p2 + p3);
        s2 = fmaf(p0, p0, s2); s2 = fmaf(p1, p1, s2); s2 = fmaf(p2, p2, s2); s2 = fmaf(p3, p3, s2);
      }
    }
    if (rok && cg == 0) Ly.rinv[row] = ri;
    if (!last) {
      s1 = row16_sum(s1); s2 = row16_sum(s2);
      float* st = a.stats + ((int64_t)l * a.nstack * R + (int64_t)s * R) * 2;
      if (rok && cg == 0) ds_put2(st + row * 2, s1, s2);
      if (l == 0) TRF(5);
      // one tile per graph: only the partial sums cross workgroups here; else the sibling tiles' rows of v do too.  While the
      // arrivals travel, the next layer's weights (in registers since a layer ago) go to big2 and the ones after are requested
      if (tiles == 1) grid_arrive<true>(a.sync, bar);
      else grid_arrive(a.sync, bar);
      {
        const DsLayer& Ln = S.layer[l + 1];
        const int ldn = pad64(Ln.n) + DS_PADB;
        if (pad64(Ln.n) != Ln.n) { zero_lds(big2, Ln.fin * ldn); __syncthreads(); }
        pf_store<DS_PFW, false>(big2, ldn, wpf, Ln.fin, Ln.n);
        if (l + 2 < S.L) pf_load(wpf, S.layer[l + 2].w, S.layer[l + 2].ldw, S.layer[l + 2].fin, S.layer[l + 2].n);
      }
      if (tiles == 1) grid_wait<true>(a.sync, bar, nblocks, a.err);
      else grid_wait(a.sync, bar, nblocks, a.err);
      if (l == 0) TRF(6);
      // every workgroup finishes the statistics of all K slots
      if (tid < K) {
        float t1, t2;
        slot_totals(st, B, K, tid, t1, t2);
        const float cnt = (float)B * (float)N;
        const float m = t1 / cnt;
        const float var = fmaxf(t2 / cnt - m * m, 0.f);
        const float rstd = 1.0f / sqrtf(var + DS_BN_EPS);
        mu[tid] = m; rs[tid] = rstd;
        if (b == 0 && t == 0) { Ly.mean[tid] = m; Ly.rstd[tid] = rstd; }
      }
      __syncthreads();
      if (l == 0) TRF(7);
      if (l == 1) TRF(8);
    }
  }
  TRF(9);
#ifndef TSGNN_TRACE_BWD
  TR_END();
#endif
  grid_finish(a.sync, bar, nblocks);
}

// ------------------------------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(DS_NT) __attribute__((amdgpu_waves_per_eu(DS_NW / 4, DS_NW / 4))) void dense_stack_bwd_kernel(DsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* big = smem;                        // W^T [n][finP + 16] / xin^T [fin][80] / dagg of the graph [K][finP + 16]
  float* At = big + DS_BIG;                 // [K][TR] columns of A
  float* aggt = At + DS_TR * DS_LDA;        // [TR][finP]   agg tile, then dagg tile, then the dx tile
  float* dut = aggt + DS_TR * 196;          // [TR][nP + 20]  du tile ; later the dA tile [TR][64]
  float* m1s = dut + DS_TR * 148;           // [K]
  float* m2s = m1s + 64;                    // [K]
  float* big2 = m2s + 64;                   // xin^T [fin][80]: a second big region, so that it is staged while W^T is still in use
  const int tid = threadIdx.x, r = (tid & 255) >> 4, cg = tid & 15;
  const bool main_t = tid < 256;                             // (helper waves shadow a main thread's addresses and store nothing)
  const int lane = tid & 63, wv = tid >> 6;
  const int K = a.K, B = a.B, R = B * K;
  const int tiles = (K + DS_TR - 1) / DS_TR;
  const unsigned nblocks = gridDim.x;
  const int t = blockIdx.x % tiles, b = (blockIdx.x / tiles) % B, s = blockIdx.x / (tiles * B);
  const DsStack& S = a.st[s];
  const int r0 = t * DS_TR, nrows = min(DS_TR, K - r0);
  const bool rok = main_t && r < nrows;
  const int64_t row = (int64_t)b * K + r0 + min(r, nrows - 1);
  float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_floats;
  float* daggS = a.dagg + (int64_t)s * R * a.finmax;
  float* dxnS = a.dxn + (int64_t)s * R * a.finmax;
  float dadj_acc[4] = {0.f, 0.f, 0.f, 0.f};               // this thread's entries (r, 4 cg + q) of the graph's dA tile
  int bar = 0, p5_uses = 0;
  // the adjacency columns of this tile (P5's A operand) are the same for every layer: staged once
  zero_lds(At, K * DS_TR);
  __syncthreads();
  if ((nrows & 3) == 0) fill_lds<false>(At, DS_TR, a.adj + (int64_t)b * K * K + r0, K, K, nrows, Ident());   // At[rr][mm] = A[rr][r0 + mm]
  else for (int i = tid; i < K * nrows; i += DS_NT) { const int rr = i / nrows, mm = i - rr * nrows; At[rr * DS_TR + mm] = a.adj[((int64_t)b * K + rr) * K + r0 + mm]; }
  TRB(0);
  for (int l = S.L - 1; l >= 0; --l) {
    const bool trl = l == S.L - 2;                           // the stamped layer: one with a batch-norm barrier
    const DsLayer& Ly = S.layer[l];
    const int fin = Ly.fin, N = Ly.n, finP = pad64(fin), NP = pad64(N), Jf = finP >> 6, Jn = NP >> 6;
    if (trl) TRB(1);
    const int ldu = NP + 20, ldf = finP + DS_PADB, lda = finP + 4;    // ldu: the du tile is a B operand (P2) and an A operand (P3)
    const bool last = l == S.L - 1;
    // ---- P1: dy = dout block (+ the gradient from the layer above), then back through BN / ReLU / L2 normalise -> du
    float dy[3][4], vv[3][4];
    {
      // every operand of the phase requested at once from clamped addresses, masked afterwards (loads inside `if (rok && ...)`
      // were waited for one by one: nine dependent L2 round trips per layer)
      const float* vsrc = last ? S.out + row * S.ldo + Ly.off : Ly.v + row * N;
      const float* esrc = last ? vsrc : dxnS + row * a.finmax;
      float4 d4[3], e4[3], v4[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = 4 * cg + 64 * j;
        const int cc = (j < Jn && c < N) ? c : 0;
        d4[j] = *reinterpret_cast<const float4*>(S.dout + row * S.lddo + Ly.off + cc);
        e4[j] = *reinterpret_cast<const float4*>(esrc + cc);
        v4[j] = *reinterpret_cast<const float4*>(vsrc + cc);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = 4 * cg + 64 * j;
        const bool ok = rok && j < Jn && c < N;
        float4 d = d4[j];
        if (!last) { d.x += e4[j].x; d.y += e4[j].y; d.z += e4[j].z; d.w += e4[j].w; }
        dy[j][0] = ok ? d.x : 0.f; dy[j][1] = ok ? d.y : 0.f; dy[j][2] = ok ? d.z : 0.f; dy[j][3] = ok ? d.w : 0.f;
        vv[j][0] = ok ? v4[j].x : 0.f; vv[j][1] = ok ? v4[j].y : 0.f; vv[j][2] = ok ? v4[j].z : 0.f; vv[j][3] = ok ? v4[j].w : 0.f;
      }
    }
    float dv[3][4];
    if (!last) {
      const int rs_ = r0 + min(r, nrows - 1);                // (clamped: unconditional requests)
      const float mean_l = Ly.mean[rs_], rstd_l = Ly.rstd[rs_];
      const float mean = rok ? mean_l : 0.f, rstd = rok ? rstd_l : 1.f;
      float p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xh = (fmaxf(vv[j][q], 0.f) - mean) * rstd;
          if (rok && j < Jn && 4 * cg + 64 * j + q < N) { p1 += dy[j][q]; p2 = fmaf(dy[j][q], xh, p2); }
        }
      p1 = row16_sum(p1); p2 = row16_sum(p2);
      float* st = a.stats + ((int64_t)l * a.nstack * R + (int64_t)s * R) * 2;
      if (rok && cg == 0) ds_put2(st + row * 2, p1, p2);
      if (trl) TRB(2);
      grid_arrive<true>(a.sync, bar);                      // (only the partial sums cross workgroups here)
    } else {
      __syncthreads();                                     // (the layer above's readers of aggt / big are done)
    }
    // under the barrier's wait: the forward data this layer's products read — the agg tile (P2), W^T (P3), xin^T (P4) — go to
    // LDS now
    zero_lds(aggt, DS_TR * lda);
    if (finP != fin) zero_lds(big, N * ldf);
    if (a.dadj && K != 64) zero_lds(big2, fin * 80);
    __syncthreads();
    fill_lds<false>(aggt, lda, Ly.agg + ((int64_t)b * K + r0) * fin, fin, nrows, fin, Ident());
    fill_lds<true>(big, ldf, Ly.w, Ly.ldw, fin, N, Ident());
    if (a.dadj) {
      if (l == 0) fill_lds<true>(big2, 80, a.x + (int64_t)b * K * a.ldx, a.ldx, K, fin, Ident());
      else fill_lds<true>(big2, 80, S.out + (int64_t)b * K * S.ldo + S.layer[l - 1].off, S.ldo, K, fin, Ident());
    }
    if (!last) {
      const int rs_ = r0 + min(r, nrows - 1);
      const float mean_l = Ly.mean[rs_], rstd_l = Ly.rstd[rs_];
      const float mean = rok ? mean_l : 0.f, rstd = rok ? rstd_l : 1.f;
      float* st = a.stats + ((int64_t)l * a.nstack * R + (int64_t)s * R) * 2;
      if (trl) TRB(3);
      grid_wait<true>(a.sync, bar, nblocks, a.err);
      if (trl) TRB(4);
      if (tid < K) {
        float t1, t2;
        slot_totals(st, B, K, tid, t1, t2);
        const float cnt = (float)B * (float)N;
        m1s[tid] = t1 / cnt; m2s[tid] = t2 / cnt;
      }
      __syncthreads();
      const float m1 = rok ? m1s[r0 + r] : 0.f, m2 = rok ? m2s[r0 + r] : 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xh = (fmaxf(vv[j][q], 0.f) - mean) * rstd;
          float g = rstd * (dy[j][q] - m1 - xh * m2);
          if (!(vv[j][q] > 0.f)) g = 0.f;
          dv[j][q] = (rok && j < Jn && 4 * cg + 64 * j + q < N) ? g : 0.f;
        }
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) dv[j][q] = dy[j][q];
    }
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) dot = fmaf(vv[j][q], dv[j][q], dot);
    dot = row16_sum(dot);
    const float ri_l = Ly.rinv[row];                       // (row is clamped)
    const float ri = rok ? ri_l : 0.f;
    if (ri >= 0.999e12f) dot = 0.f;                        // clamped norm: F.normalize passes no norm gradient
#pragma unroll
    for (int j = 0; j < 3; ++j)                            // (dut's readers of the layer above are behind the barriers since)
      if (main_t && j < Jn)
        *reinterpret_cast<float4*>(dut + r * ldu + 4 * cg + 64 * j) =
            rok ? make_float4(ri * (dv[j][0] - vv[j][0] * dot), ri * (dv[j][1] - vv[j][1] * dot), ri * (dv[j][2] - vv[j][2] * dot),
                              ri * (dv[j][3] - vv[j][3] * dot))
                : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (trl) TRB(5);
    // ---- P2: this tile's share of dW = agg^T du (MFMA: M = fin, N = n, depth = the tile's 16 rows) and db = colsum(du) -> the
    //          workgroup's slab (the slabs are summed in a fixed order at the end)
    {
      float* sl = slab + Ly.slab_off;
      const int i = lane & 15, kq = lane >> 4;
      const int nkb = (fin + 15) >> 4, ncg = (N + 15) >> 4;
      for (int tl = wv; tl < nkb * ncg; tl += DS_NW) {
        const int kb = tl / ncg, g = tl - kb * ncg;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float af[4], bf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = 4 * u + kq;
          af[u] = aggt[rr * lda + 16 * kb + i];
          bf[u] = dut[rr * ldu + 16 * g + i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u], bf[u], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int k = 16 * kb + 4 * kq + v, c = 16 * g + i;
          if (k < fin && c < N) sl[(int64_t)k * N + c] = acc[v];
        }
      }
      for (int c = tid; c < N; c += DS_NT) {
        float sacc = 0.f;
#pragma unroll
        for (int rr = 0; rr < DS_TR; ++rr) sacc += dut[rr * ldu + c];
        sl[fin * N + c] = sacc;
      }
    }
    if (trl) TRB(6);
    // ---- P3: dagg tile = du . W^T   (W^T [n][finP + 16] went to LDS under the barrier: the product has the forward's shape)
    __syncthreads();                                       // (P2's readers of aggt are done)
    tile_mfma(aggt, lda, dut, ldu, 1, big, ldf, N, finP >> 4);
    __syncthreads();
    float dg[3][4];
    read_tile(dg, aggt, lda, r, cg, Jf);
    if (rok) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int k = 4 * cg + 64 * j;
        if (j < Jf && k < fin) *reinterpret_cast<float4*>(daggS + row * a.finmax + k) = make_float4(dg[j][0], dg[j][1], dg[j][2], dg[j][3]);
      }
    }
    if (trl) TRB(7);
    // ---- P4: dA[tile rows, m] += dagg[r, :] . xin[m, :]   (xin^T [fin][80] in LDS; the dA tile lands in the du tile's place)
    if (a.dadj) {                                          // (xin^T went to big2 under the barrier)
      tile_mfma(dut, 64, aggt, lda, 1, big2, 80, fin, 4);
      __syncthreads();
      const float4 v = *reinterpret_cast<const float4*>(dut + r * 64 + 4 * cg);
      dadj_acc[0] += v.x; dadj_acc[1] += v.y; dadj_acc[2] += v.z; dadj_acc[3] += v.w;
    }
    if (trl) TRB(8);
    // ---- P5: dx[tile rows m, :] = sum_r A[r, m] dagg[r, :] over ALL rows r of the graph: the sibling tiles' dagg first
    if (l > 0 || a.dx) {
      if (tiles == 1) __syncthreads();                       // the graph is this tile: its dagg rows are the workgroup's own writes
      else group_barrier(a.sync + 32 + s * B + b, (unsigned)tiles * (unsigned)(++p5_uses), a.err);
      if (trl) TRB(9);
      if (finP != fin) zero_lds(big, K * ldf);
      __syncthreads();
      fill_lds<false>(big, ldf, daggS + (int64_t)b * K * a.finmax, a.finmax, K, fin, Ident());
      __syncthreads();
      tile_mfma(aggt, lda, At, 1, DS_TR, big, ldf, K, finP >> 4);
      __syncthreads();
      float dxv[3][4];
      read_tile(dxv, aggt, lda, r, cg, Jf);
      if (rok) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int c = 4 * cg + 64 * j;
          if (j < Jf && c < fin) *reinterpret_cast<float4*>(dxnS + row * a.finmax + c) = make_float4(dxv[j][0], dxv[j][1], dxv[j][2], dxv[j][3]);
        }
      }
    }
  }
  TRB(10);
  // ---- dA tile of this stack
  if (a.dadj) {
    float* dst = a.nstack == 1 ? a.dadj : a.dadj_part + (int64_t)s * R * K;
    if (rok) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * cg + q < K)
          dst[row * K + 4 * cg + q] = dadj_acc[q] + ((a.nstack == 1 && a.dadj_add) ? a.dadj_add[row * K + 4 * cg + q] : 0.f);
    }
  }
  grid_barrier(a.sync, bar, nblocks, a.err);
  if (tid == 0 && t == 0 && tiles > 1) __hip_atomic_store(a.sync + 32 + s * B + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
  TRB(11);
  // ---- weight / bias gradients: every element is the sum of the stack's workgroups' slabs, in workgroup order.  The elements of
  // ALL layers of both stacks are one index space (a slab holds its layers back to back), so every thread of the grid has its
  // element at once; layer after layer, each loop kept a fifth of the threads busy with eight dependent rounds of loads.
  {
    const int per_stack = tiles * B;
    int used[2] = {0, 0};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
      if (ss < a.nstack)
#pragma unroll
        for (int l = 0; l < DS_MAXL; ++l)
          if (l < a.st[ss].L) used[ss] = (int)a.st[ss].layer[l].slab_off + (a.st[ss].layer[l].fin + 1) * a.st[ss].layer[l].n;
    const int total = used[0] + used[1];
    for (int g = (int)blockIdx.x * DS_NT + tid; g < total; g += (int)nblocks * DS_NT) {
      const int ss = g >= used[0] ? 1 : 0;
      const int e = g - (ss ? used[0] : 0);                 // offset inside the slab
      const float* p = a.slabs + (int64_t)ss * per_stack * a.slab_floats + e;
      float sacc = 0.f;
      int w = 0;
      for (; w + 16 <= per_stack; w += 16) {               // sixteen slabs in flight, added in workgroup order
        float v[16];
#pragma unroll
        for (int u_ = 0; u_ < 16; ++u_) v[u_] = p[(int64_t)(w + u_) * a.slab_floats];
#pragma unroll
        for (int u_ = 0; u_ < 16; ++u_) sacc += v[u_];
      }
      for (; w < per_stack; ++w) sacc += p[(int64_t)w * a.slab_floats];
      // which layer's dw / db entry this is (compile-time walk over the descriptions: no dynamic indexing of the arguments)
      float* dst = nullptr;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int l = 0; l < DS_MAXL; ++l) {
          if (s2 < a.nstack && l < a.st[s2].L && s2 == ss) {
            const DsLayer& Ly = a.st[s2].layer[l];
            const int o = e - (int)Ly.slab_off, nw = Ly.fin * Ly.n;
            if (o >= 0 && o < nw + Ly.n) dst = o < nw ? (Ly.dw ? Ly.dw + o : nullptr) : (Ly.db ? Ly.db + (o - nw) : nullptr);
          }
        }
      if (dst) *dst = sacc;
    }
  }
  TRB(12);
  // ---- the stacks' input gradients add up (two stacks: both read the same x and adjacency)
  if (a.dx) {
    const int f0 = a.fin0;
    const FastDiv by_f0(f0);                               // (R * f0 < 2^22: ds_check)
    for (int e = (int)blockIdx.x * DS_NT + tid; e < R * f0; e += (int)nblocks * DS_NT) {
      const int64_t rr = by_f0(e); const int c = (int)(e - rr * f0);
      float v = a.dxn[rr * a.finmax + c];
      if (a.nstack == 2) v += a.dxn[(int64_t)R * a.finmax + rr * a.finmax + c];
      a.dx[rr * a.lddx + c] = v;
    }
  }
  if (a.dadj && a.nstack == 2) {
    for (int64_t e = (int64_t)blockIdx.x * DS_NT + tid; e < (int64_t)R * K; e += (int64_t)nblocks * DS_NT)
      a.dadj[e] = (a.dadj_part[e] + a.dadj_part[(int64_t)R * K + e]) + (a.dadj_add ? a.dadj_add[e] : 0.f);
  }
  TRB(13);
#ifdef TSGNN_TRACE_BWD
  TR_END();
#endif
  grid_finish(a.sync, bar, nblocks);
}

constexpr size_t ds_lds_bytes() { return sizeof(float) * (2 * DS_BIG + DS_TR * 68 + DS_TR * 196 + DS_TR * 148 + 128); }   // 150 KB: one workgroup per CU

// Test hook (tsgnn_dense_stack_barrier_selftest): ONE workgroup waits at a device-wide barrier for `expect` arrivals.  expect = 1
// completes; expect = 2 can never complete — the bounded spin gives up and raises the error word, exactly what a launch whose
// workgroups are not co-resident does.
__global__ __launch_bounds__(64) void ds_barrier_selftest_kernel(unsigned* sync, float* err, unsigned expect) {
  int bar = 0;
  grid_barrier(sync, bar, expect, err);
  grid_finish(sync, bar, 1u);
}

// How many workgroups of these kernels the CURRENT device keeps resident at once — what the spin barriers rest on.  From the
// device itself (compute units x the occupancy the runtime reports for each kernel at DS_NT threads and ds_lds_bytes() of LDS),
// not a literal: a partition or a part with fewer compute units gets the limit that is true for it.  Also raises the kernels'
// dynamic-LDS limit, once per DEVICE (the attribute is per device; a process-wide flag left the second device of a process
// without it).  0 when no device is usable: the callers then report "unsupported" and the layer-by-layer path runs.
int ds_max_resident() {
  constexpr int MAXDEV = 64;
  static int cache[MAXDEV];                   // 0 = not queried yet, -1 = unusable, > 0 = the limit
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) { (void)hipGetLastError(); return 0; }
  if (cache[dev] != 0) return cache[dev] > 0 ? cache[dev] : 0;
  int cus = 0, occ_f = 0, occ_b = 0;
  bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(dense_stack_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ds_lds_bytes()) == hipSuccess;
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(dense_stack_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ds_lds_bytes()) == hipSuccess;
  ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_f, dense_stack_fwd_kernel, DS_NT, ds_lds_bytes()) == hipSuccess;
  ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, dense_stack_bwd_kernel, DS_NT, ds_lds_bytes()) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  const int occ = occ_f < occ_b ? occ_f : occ_b;
  cache[dev] = (ok && cus > 0 && occ > 0) ? cus * occ : -1;
  return cache[dev] > 0 ? cache[dev] : 0;
}

int ds_check(const DsArgs& a) {
  if (!a.x || !a.adj || !a.stats || !a.sync || !a.err || a.B <= 0 || a.K <= 0 || a.nstack < 1 || a.nstack > 2) return TSGNN_EINVAL;
  if (a.K > 64 || (a.K % 4) || (a.ldx % 4) || (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.adj) & 15)) return TSGNN_EUNSUPPORTED;
  for (int s = 0; s < a.nstack; ++s) {
    const DsStack& S = a.st[s];
    if (S.L < 1 || S.L > DS_MAXL || !S.out) return TSGNN_EINVAL;
    int prev = a.fin0;
    for (int l = 0; l < S.L; ++l) {
      const DsLayer& y = S.layer[l];
      if (!y.w || !y.agg || !y.rinv || y.fin != prev || y.n <= 0) return TSGNN_EINVAL;
      if (l < S.L - 1 && (!y.v || !y.mean || !y.rstd)) return TSGNN_EINVAL;
      const int finP = (y.fin + 63) & ~63, nP = (y.n + 63) & ~63;
      if (y.fin > 192 || y.n > 128 || (y.fin % 4) || (y.n % 4) || (y.ldw % 4) || (reinterpret_cast<uintptr_t>(y.w) & 15) ||
          (int64_t)a.K * (finP + 16) > DS_BIG || (int64_t)y.fin * (nP + 16) > DS_BIG || (int64_t)y.n * (finP + 16) > DS_BIG || (int64_t)y.fin * 80 > DS_BIG)
        return TSGNN_EUNSUPPORTED;
      prev = y.n;
    }
  }
  if (a.nstack == 2 && a.st[0].L != a.st[1].L) return TSGNN_EUNSUPPORTED;      // the workgroups of both stacks meet at the same barriers
  const int tiles = (a.K + DS_TR - 1) / DS_TR;
  if ((int64_t)tiles * a.B * a.nstack > ds_max_resident()) return TSGNN_EUNSUPPORTED;       // all workgroups must be resident
  return TSGNN_OK;
}

}  // namespace

extern "C" {

/* 1 if the one-launch pooled-level stack kernels take these shapes (B graphs of K nodes, up to two stacks of <= 4 layers) */
int tsgnn_dense_stack_supported(int B, int K, int nstack, int L, int fin0, int hidden, int last0, int last1) {
  if (B <= 0 || K <= 0 || K > 64 || (K % 4) || nstack < 1 || nstack > 2 || L < 1 || L > DS_MAXL) return 0;
  const int tiles = (K + DS_TR - 1) / DS_TR;
  if ((int64_t)tiles * B * nstack > ds_max_resident()) return 0;
  const int widths[4] = {fin0, hidden, last0, last1};
  for (int i = 0; i < 4; ++i)
    if (widths[i] <= 0 || (widths[i] % 4)) return 0;
  const int fmax = fin0 > hidden ? fin0 : hidden;
  const int nmax_ = hidden > last0 ? (hidden > last1 ? hidden : last1) : (last0 > last1 ? last0 : last1);
  const int fP = (fmax + 63) & ~63, nP = (nmax_ + 63) & ~63;
  if (fmax > 192 || nmax_ > 128 || (int64_t)K * (fP + 16) > DS_BIG || (int64_t)fmax * (nP + 16) > DS_BIG ||
      (int64_t)nmax_ * (fP + 16) > DS_BIG || (int64_t)fmax * 80 > DS_BIG)
    return 0;
  return 1;
}

/* Pooled-level GCN stacks, forward, ONE launch (dense_stack.hip).  `desc` is a host array of 8-byte words describing the
 * problem (built by two_stage_gnn_amd/dense_stack.py::_describe):
 *   [0] x  [1] ldx  [2] fin0  [3] adj  [4] B  [5] K  [6] nstack  [7] stats  [8] sync  [9] err
 *   [10] dagg  [11] dxn  [12] slabs  [13] slab_floats  [14] finmax  [15] dx  [16] lddx  [17] dadj  [18] dadj_part
 *   then per stack (2 x): out, ldo, dout, lddo, L, and per layer (4 x): w, ldw, bias, fin, n, off, agg, v, rinv, mean, rstd,
 *   dw, db, slab_off; then [141] dadj_add (nullable).
 * Per hidden layer: u = (A x) W + b, v = u / max(|u|, 1e-12), y = BN_slot(ReLU(v)) (fresh statistics over batch and features,
 * eps 1e-5, biased variance); last layer: v only (encoders.py:140-167).  out[:, off_l : off_l + n_l] = the layer's output. */
int tsgnn_dense_stack_fwd_f32(const int64_t* desc, tsgnn_stream_t stream);
int tsgnn_dense_stack_bwd_f32(const int64_t* desc, tsgnn_stream_t stream);

static void ds_unpack(const int64_t* d, DsArgs& a) {
  auto P = [&](int i) { return reinterpret_cast<float*>(static_cast<uintptr_t>(d[i])); };
  a.x = P(0); a.ldx = d[1]; a.fin0 = (int)d[2]; a.adj = P(3); a.B = (int)d[4]; a.K = (int)d[5]; a.nstack = (int)d[6];
  a.stats = P(7); a.sync = reinterpret_cast<unsigned*>(static_cast<uintptr_t>(d[8])); a.err = P(9);
  a.dagg = P(10); a.dxn = P(11); a.slabs = P(12); a.slab_floats = d[13]; a.finmax = (int)d[14];
  a.dx = P(15); a.lddx = d[16]; a.dadj = P(17); a.dadj_part = P(18);
  int o = 19;
  for (int s = 0; s < 2; ++s) {
    DsStack& S = a.st[s];
    S.out = P(o); S.ldo = d[o + 1]; S.dout = P(o + 2); S.lddo = d[o + 3]; S.L = (int)d[o + 4];
    o += 5;
    for (int l = 0; l < DS_MAXL; ++l) {
      DsLayer& y = S.layer[l];
      y.w = P(o); y.ldw = d[o + 1]; y.bias = P(o + 2); y.fin = (int)d[o + 3]; y.n = (int)d[o + 4]; y.off = (int)d[o + 5];
      y.agg = P(o + 6); y.v = P(o + 7); y.rinv = P(o + 8); y.mean = P(o + 9); y.rstd = P(o + 10); y.dw = P(o + 11); y.db = P(o + 12);
      y.slab_off = d[o + 13];
      o += 14;
    }
  }
  a.dadj_add = P(o);
}

int tsgnn_dense_stack_fwd_f32(const int64_t* desc, tsgnn_stream_t stream) {
  if (!desc) return TSGNN_EINVAL;
  DsArgs a;
  ds_unpack(desc, a);
  const int rc = ds_check(a);
  if (rc != TSGNN_OK) return rc;
  const int tiles = (a.K + DS_TR - 1) / DS_TR;
  TSGNN_KNAME("dense_stack_fwd_kernel");
  dense_stack_fwd_kernel<<<(unsigned)(tiles * a.B * a.nstack), DS_NT, ds_lds_bytes(), stream>>>(a);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

int tsgnn_dense_stack_bwd_f32(const int64_t* desc, tsgnn_stream_t stream) {
  if (!desc) return TSGNN_EINVAL;
  DsArgs a;
  ds_unpack(desc, a);
  const int rc = ds_check(a);
  if (rc != TSGNN_OK) return rc;
  if (!a.dagg || !a.dxn || !a.slabs || a.slab_floats <= 0 || a.finmax <= 0 || (a.dadj && a.nstack == 2 && !a.dadj_part)) return TSGNN_EINVAL;
  for (int s = 0; s < a.nstack; ++s)
    if (!a.st[s].dout) return TSGNN_EINVAL;
  const int tiles = (a.K + DS_TR - 1) / DS_TR;
  TSGNN_KNAME("dense_stack_bwd_kernel");
  dense_stack_bwd_kernel<<<(unsigned)(tiles * a.B * a.nstack), DS_NT, ds_lds_bytes(), stream>>>(a);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* Test hook for the failure path of the bounded device-wide barriers: one workgroup waits for `expect` arrivals on the barrier
 * words `sync` (>= 32 words, zero).  expect = 1 passes; expect = 2 times out after the barrier's bound (~1 s) and stores 1.0 to
 * *err — what a dense-stack launch whose workgroups were not all resident does.  tests/test_gpu_encoders.py uses it to prove
 * that the optimiser then skips its update and the host raises. */
int tsgnn_dense_stack_barrier_selftest(unsigned* sync, float* err, int expect, tsgnn_stream_t stream) {
  if (!sync || !err || expect < 1) return TSGNN_EINVAL;
  ds_barrier_selftest_kernel<<<1, 64, 0, stream>>>(sync, err, (unsigned)expect);
  TSGNN_CHECK_LAUNCH();
  return TSGNN_OK;
}

/* workgroups of the dense-stack kernels the current device keeps resident (compute units x occupancy); 0 without a device */
int tsgnn_dense_stack_max_resident(void) { return ds_max_resident(); }

}  // extern "C"
