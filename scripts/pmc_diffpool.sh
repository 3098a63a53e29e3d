#!/bin/bash
# MFMA-busy / wave-cycle counters (one counter per rocprofv3 pass) of the DiffPool step's MFMA kernels: the level-1 contraction
# (ragged_tn_direct_kernel — gemm_tn_rows_kernel before round 3 —: S^T Z, S^T (A S)) and the pooled-level stack kernels (dense_stack_fwd / bwd).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_dp_$c -- python3 scripts/prof_diffpool.py > /dev/null 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob
for kern in ("gemm_tn_rows", "ragged_tn_direct", "dense_stack_fwd", "dense_stack_bwd", "contract_rows_bwd", "contract_dense_bwd", "sage_multi_kernel<2, 2>", "sage_multi_kernel<0, 1>"):
    out = {}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_WAIT_ANY"):
        f = glob.glob("gpurun_out/pmc_dp_%s/*/*_counter_collection.csv" % c)[0]
        v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == c)
        out[c] = v[len(v) // 2] if v else None
    w = out["SQ_WAVES"] or 1
    if out["SQ_WAVE_CYCLES"]:
        print("%-22s waves %6.0f  wave cycles %9.0f (x4 quad)  MFMA busy %9.0f  wait %9.0f  -> MFMA pipe busy %.1f %% of a wave's lifetime, waiting %.0f %%"
              % (kern, w, out["SQ_WAVE_CYCLES"], out["SQ_VALU_MFMA_BUSY_CYCLES"] or 0, out["SQ_WAIT_ANY"] or 0,
                 100 * (out["SQ_VALU_MFMA_BUSY_CYCLES"] or 0) / (4 * out["SQ_WAVE_CYCLES"]), 100 * (out["SQ_WAIT_ANY"] or 0) / out["SQ_WAVE_CYCLES"]))
PY
