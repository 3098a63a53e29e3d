// The blocked weight gradient's slab role (gemm.hip: wgrad_blocks_kernel; gat_products.hip: beside the input-gradient product).
#pragma once
#include "tn_rows_body.h"

namespace {

constexpr int WB_MAXSETS = 16;
constexpr int64_t WB_SET_FLOATS = 129 * 128;            // per slab: [kc + 1][nc] <= 129 x 128
struct WgradBlocks {
  TnArgs g;                                             // z, du, rows, rows_per_slab; K_in / N hold the TOTAL widths
  int NB, nsets;                                        // column blocks; sets = row blocks x column blocks
  int64_t set_stride;                                   // floats between the slab arrays of two sets
};

// block (bx = slab, by_all in [0, NY * nsets)) of the blocked weight gradient
template <int NY>
__device__ __forceinline__ void wgrad_blocks_role(const WgradBlocks& w, float* tn_smem, unsigned bx, unsigned by_all, unsigned nslab) {
  const int set = (int)by_all / NY, by = (int)by_all % NY;
  const int kb = set / w.NB, nb = set % w.NB;
  TnArgs g = w.g;
  g.z += 128 * kb;
  g.du += 128 * nb;
  g.K_in = min(128, w.g.K_in - 128 * kb);
  g.N = min(128, w.g.N - 128 * nb);
  g.slabs = w.g.slabs + (int64_t)set * w.set_stride;
  // (row tiles of 32: a 92-row block — the GAT input projection — runs three of them, not four)
  if (g.N <= 32) {                                                                   // a narrow last column block (the 2H score columns)
    if (g.K_in <= 96) tn_rows_body<3, 1, NY>(g, tn_smem, bx, by, nslab);
    else tn_rows_body<4, 1, NY>(g, tn_smem, bx, by, nslab);
  } else {
    if (g.K_in <= 96) tn_rows_body<3, 4, NY>(g, tn_smem, bx, by, nslab);
    else tn_rows_body<4, 4, NY>(g, tn_smem, bx, by, nslab);
  }
}

}  // namespace
