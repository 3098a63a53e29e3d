#!/bin/bash
# HBM traffic of every kernel of the timed step from PMC counters (separate passes, as MI355X_MICROARCH.md §HBM prescribes).
# Run on the GPU box:  bash scripts/pmc_step.sh   -> gpurun_out/pmc_step_{f,w}/  then parsed into profiles/r03/step_traffic.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
ARGS="--no-graph --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-kernels --no-pyg --no-seeds"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_step_f -- python3 bench.py $ARGS > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_step_w -- python3 bench.py $ARGS > /dev/null 2>&1 &&
python3 scripts/pmc_step.py
