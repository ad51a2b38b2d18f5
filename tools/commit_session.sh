#!/bin/bash
# copies the summaries of a tools/gpu_session.sh run from gpurun_out/<TAG>/ (scratch) into profiles/ (tracked):  tools/commit_session.sh r06_s2
T=$1; O=gpurun_out/$T
for f in bench.json bench_2p22.json bench_2p24.json bench_group_2x_same_gpu.json bench_under_rocprof.json bench_under_rocprof_serialised.json dist_probe.json env.log \
         kernel_stats_proofs_only.csv kernel_stats_proofs_only_serialised.csv msm_probe.json node_bench.json node_bench_group.json pmc_calibration.json \
         pmc_sq_counters.json proof_issue_budget.json pytest_gpu.txt shard_probe_2p20.json smoke.txt issue_classes.json; do
  [ -s "$O/$f" ] && cp "$O/$f" "profiles/${T}_$f"
done
[ -s "$O/pmc_traffic.json" ] && cp "$O/pmc_traffic.json" "profiles/${T}_pmc_traffic_table.json"
ls profiles | grep -c "^${T}_"
