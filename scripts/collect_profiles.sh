#!/bin/bash
# Collects the round's measurements on the GPU box (run through gpurun); outputs under gpurun_out/$R/ (R = TSGNN_ROUND, default r04),
# copied to profiles/$R/ afterwards.
#   bash scripts/collect_profiles.sh [bench|rocprof|pmc|ingest|configs|diffpool_pmc|diffpool_stats|diffpool_replay|gat_stats|gat_replay|bench_replay|sagpool_replay|triplet_replay ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${TSGNN_ROUND:-r04}; export TSGNN_ROUND=$R
O=gpurun_out/$R; mkdir -p $O profiles/$R
for what in "$@"; do
  case $what in
    bench)   python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1 ;;
    rocprof) rm -rf $O/rocprof_bench
             rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_bench -- python3 bench.py --no-cpu-baseline --no-seeds --no-pyg --steps-per-graph 1 > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
             cp $O/rocprof_bench/*/*kernel_stats.csv $O/bench_b32_kernel_stats.csv ;;
    pmc)     bash scripts/pmc_step.sh > $O/pmc_step.log 2>&1 || exit 1
             cp profiles/$R/step_traffic.json $O/step_traffic.json ;;
    ingest)  python3 bench.py --ingest --no-cpu-baseline --no-sweep --no-kernels --no-seeds > $O/bench_ingest.json 2> $O/ingest.err || exit 1 ;;
    configs) python3 scripts/config_bench.py > $O/secondary_configs.txt 2>&1; python3 scripts/diffpool_step.py >> $O/secondary_configs.txt 2>&1; python3 scripts/gat_step.py >> $O/secondary_configs.txt 2>&1 ;;
    diffpool_pmc) bash scripts/pmc_diffpool.sh > $O/diffpool_mfma_pmc.txt 2>&1 ;;
    diffpool_stats) rm -rf $O/rocprof_dp
             rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_dp -- python3 scripts/prof_diffpool.py > /dev/null 2>&1 && cp $O/rocprof_dp/*/*kernel_stats.csv $O/diffpool_kernel_stats.csv ;;
    diffpool_replay) rm -rf $O/rocprof_dpr
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_dpr -- python3 scripts/diffpool_step.py > /dev/null 2>&1 &&
             python3 scripts/replay_trace.py $(ls $O/rocprof_dpr/*/*kernel_trace.csv | head -1) tn_rows_reduce_multi > $O/diffpool_replay_timeline.txt; rm -rf $O/rocprof_dpr ;;
    gat_replay) rm -rf $O/rocprof_gr
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_gr -- python3 scripts/gat_step.py > /dev/null 2>&1 &&
             python3 scripts/replay_trace.py $(ls $O/rocprof_gr/*/*kernel_trace.csv | head -1) gat_unpack_kernel > $O/gat_replay_timeline.txt; rm -rf $O/rocprof_gr ;;
    bench_replay) rm -rf $O/rocprof_br
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_br -- python3 bench.py --no-cpu-baseline --no-seeds --no-sweep --no-kernels --no-pyg --steps-per-graph 1 > /dev/null 2>&1 &&
             python3 scripts/replay_trace.py $(ls $O/rocprof_br/*/*kernel_trace.csv | head -1) adam_from_partials > $O/bench_replay_timeline.txt; rm -rf $O/rocprof_br ;;
    sagpool_replay) rm -rf $O/rocprof_sr
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_sr -- python3 scripts/sagpool_step.py > /dev/null 2>&1 &&
             python3 scripts/replay_trace.py $(ls $O/rocprof_sr/*/*kernel_trace.csv | head -1) splitk_reduce_kernel > $O/sagpool_replay_timeline.txt; rm -rf $O/rocprof_sr
             python3 scripts/sagpool_step.py >> $O/sagpool_replay_timeline.txt 2>/dev/null ;;
    triplet_replay) rm -rf $O/rocprof_tr
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_tr -- python3 scripts/triplet_step.py > /dev/null 2>&1 &&
             python3 scripts/replay_trace.py $(ls $O/rocprof_tr/*/*kernel_trace.csv | head -1) adam_update > $O/triplet_replay_timeline.txt; rm -rf $O/rocprof_tr
             python3 scripts/triplet_step.py >> $O/triplet_replay_timeline.txt 2>/dev/null
             TRIPLET_CRITERION=torch python3 scripts/triplet_step.py >> $O/triplet_replay_timeline.txt 2>/dev/null ;;
    pyg)     python3 scripts/pyg_bench.py > $O/pyg_surface.txt 2>&1; python3 scripts/pyg_bench.py GAT >> $O/pyg_surface.txt 2>&1
             python3 scripts/pyg_bench.py SAGPOOL >> $O/pyg_surface.txt 2>&1 ;;
    pyg_replay) for cfg in "DD 0" "DD 6" "PROTEINS 1" "MUTAG 0"; do set -- $cfg; rm -rf $O/rocprof_pr
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_pr -- python3 scripts/pyg_step.py $1 $2 50 > $O/pyg_step_$1_$2.log 2>&1 &&
             python3 scripts/replay_trace.py $(ls $O/rocprof_pr/*/*kernel_trace.csv | head -1) adam > $O/pyg_sage_$1_seed$2_replay_timeline.txt
             grep "us/step" $O/pyg_step_$1_$2.log >> $O/pyg_sage_$1_seed$2_replay_timeline.txt; rm -f $O/pyg_step_$1_$2.log; done
             rm -rf $O/rocprof_pr
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_pr -- python3 scripts/pyg_gat_step.py > $O/pyg_gat.log 2>&1 &&
             python3 scripts/replay_trace.py $(ls $O/rocprof_pr/*/*kernel_trace.csv | head -1) adam > $O/pyg_gat_replay_timeline.txt
             grep "us/step" $O/pyg_gat.log >> $O/pyg_gat_replay_timeline.txt; rm -rf $O/rocprof_pr $O/pyg_gat.log ;;
    sagpool_sage) for c in gcn sage; do rm -rf $O/rocprof_sp
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_sp -- python3 scripts/sagpool_sage_step.py $c 50 > $O/sp_$c.log 2>&1 &&
             python3 scripts/replay_trace.py $(ls $O/rocprof_sp/*/*kernel_trace.csv | head -1) adam > $O/sagpool_${c}conv_replay_timeline.txt
             grep "us/step" $O/sp_$c.log >> $O/sagpool_${c}conv_replay_timeline.txt; rm -f $O/sp_$c.log; done; rm -rf $O/rocprof_sp ;;
    pyg_stats) rm -rf $O/rocprof_ps
             rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_ps -- python3 scripts/pyg_step.py DD 0 400 > /dev/null 2>&1 &&
             cp $O/rocprof_ps/*/*kernel_stats.csv $O/pyg_sage_dd_b32_kernel_stats.csv; rm -rf $O/rocprof_ps ;;
    pyg_pmc) rm -rf gpurun_out/pmc_pyg_f gpurun_out/pmc_pyg_w
             PYG_EAGER=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_pyg_f -- python3 scripts/pyg_step.py DD 0 30 > /dev/null 2>&1 &&
             PYG_EAGER=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_pyg_w -- python3 scripts/pyg_step.py DD 0 30 > /dev/null 2>&1 &&
             python3 scripts/pmc_step.py pmc_pyg pyg_step_traffic.json > $O/pmc_pyg.log 2>&1; cp profiles/$R/pyg_step_traffic.json $O/ ;;
    traj)    TSGNN_TRAJ_STEPS=200 python3 -m pytest tests/test_gpu_fullsize.py -k trajectory -q -m gpu -s > $O/trajectories_raw.txt 2>&1
             grep -E "step +[0-9]+: loss|passed|failed" $O/trajectories_raw.txt > $O/trajectories.txt; rm -f $O/trajectories_raw.txt ;;
    fullsize_log) python3 -m pytest tests/test_gpu_fullsize.py -k "timed_step or config4" -q -m gpu -s > $O/fullsize_raw.txt 2>&1
             grep -E "step [0-9]: loss|graphs counted|of 128 graphs|passed|failed" $O/fullsize_raw.txt > $O/fullsize_parity.txt; rm -f $O/fullsize_raw.txt ;;
    sageconv_trace) hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTSGNN_TRACE -DTSGNN_TRACE_WPB=8 scripts/trace_sageconv.hip -o /tmp/trace_sageconv 2>/dev/null &&
             (/tmp/trace_sageconv 8151 128 128; /tmp/trace_sageconv 9191 128 128; /tmp/trace_sageconv 8151 89 89) > $O/trace_sageconv.txt 2>&1 ;;
    gat_stats) rm -rf $O/rocprof_gat
             GAT_EAGER=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_gat -- python3 scripts/gat_step.py > /dev/null 2>&1 && cp $O/rocprof_gat/*/*kernel_stats.csv $O/gat_b32_kernel_stats.csv ;;
  esac
  echo "$what done"
done
