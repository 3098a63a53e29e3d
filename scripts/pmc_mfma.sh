#!/bin/bash
# MFMA-busy / wave-cycle counters (one counter per pass) of the two-group gather kernel and of the DiffPool contraction kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_m_g_$c -- python3 scripts/prof_one.py gather > /dev/null 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_m_c_$c -- python3 scripts/prof_contract.py > /dev/null 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for tag, kern in (("g", "rowgemm_gather_ks2"), ("c", "gemm_tn_rows"), ("c", "tn_rows_reduce"), ("c", "spmm_")):
    out = {}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_WAIT_ANY"):
        f = glob.glob("gpurun_out/pmc_m_%s_%s/*/*_counter_collection.csv" % (tag, c))[0]
        v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == c)
        out[c] = v[len(v) // 2] if v else None
    print(kern, out)
PY
