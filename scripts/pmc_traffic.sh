#!/bin/bash
# HBM traffic of the aggregation kernel from PMC counters (separate passes, as MI355X_MICROARCH.md §HBM prescribes).
# Run on the GPU box:  bash scripts/pmc_traffic.sh   -> gpurun_out/pmc_*/  then  python scripts/pmc_traffic.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 scripts/prof_one.py ell > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 scripts/prof_one.py ell > /dev/null 2>&1 &&
B=2048 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f2k -- python3 scripts/prof_one.py spmm > /dev/null 2>&1 &&
B=2048 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w2k -- python3 scripts/prof_one.py spmm > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_gf -- python3 scripts/prof_one.py gather > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_gw -- python3 scripts/prof_one.py gather > /dev/null 2>&1
