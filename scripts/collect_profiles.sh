#!/bin/bash
# Collects the round's measurements on the GPU box (run through gpurun); outputs under gpurun_out/r03/, copied to profiles/r03/ afterwards.
#   bash scripts/collect_profiles.sh [bench|rocprof|pmc|ingest|configs|diffpool_pmc|diffpool_stats|diffpool_replay|gat_stats|gat_replay|bench_replay|sagpool_replay|triplet_replay ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03; mkdir -p $O
for what in "$@"; do
  case $what in
    bench)   python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1 ;;
    rocprof) rm -rf $O/rocprof_bench
             rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_bench -- python3 bench.py --no-cpu-baseline --no-seeds --steps-per-graph 1 > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
             cp $O/rocprof_bench/*/*kernel_stats.csv $O/bench_b32_kernel_stats.csv ;;
    pmc)     bash scripts/pmc_step.sh > $O/pmc_step.log 2>&1 || exit 1
             cp profiles/r03/step_traffic.json $O/step_traffic.json ;;
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
             rocprofv3 --kernel-trace --output-format csv -d $O/rocprof_br -- python3 bench.py --no-cpu-baseline --no-seeds --no-sweep --no-kernels --steps-per-graph 1 > /dev/null 2>&1 &&
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
    gat_stats) rm -rf $O/rocprof_gat
             GAT_EAGER=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_gat -- python3 scripts/gat_step.py > /dev/null 2>&1 && cp $O/rocprof_gat/*/*kernel_stats.csv $O/gat_b32_kernel_stats.csv ;;
  esac
  echo "$what done"
done
