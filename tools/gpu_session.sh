#!/bin/bash
# One gpurun call: environment probe, GPU parity tests, bench, rocprof kernel stats, PMC traffic.
#   TAG=r02_s1 [SKIP="tests pmc"] [PYTEST_K="expr for -k"] tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r02}
O=gpurun_out/$TAG
mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
  echo "== env"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
  nproc; free -g | head -2; node --version 2>&1; python --version; rocm-smi --showid 2>/dev/null | grep -c "GPU\[" 
} > "$O/env.log" 2>&1
skip() { [[ " $SKIP " == *" $1 "* ]]; }
# VALU issue rate per instruction class on THIS box (wsnark_peak_probe 6..28): prices the issue floors of the PMC summaries below
timeout 300 python tools/issue_probe.py 3 > "$O/issue_classes.json" 2>> "$O/env.log"
if ! skip tests; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider --durations=12 ${PYTEST_K:+-k "$PYTEST_K"} > "$O/pytest_gpu.txt" 2>&1
  echo "pytest rc=$?" >> "$O/env.log"
fi
if ! skip bench; then
  timeout 1500 python bench.py $BENCH_ARGS > "$O/bench.json" 2> "$O/bench.err"
  echo "bench rc=$?" >> "$O/env.log"
fi
# rocprofv3 --kernel-trace --stats of the bench command, PROOFS ONLY (no extras: every msm_accumulate launch then has the shape
# bench.py's roofline is quoted on), twice: the shipped two-queue schedule (in situ), and WSNARK_PROVE_OVERLAP=0 (one queue: every
# kernel alone -- where the profiler and the bench's HIP events must agree).  The bench line of each profiled run is kept beside
# its summary (the events of THAT run are the ones to compare with its csv).
PROF_CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
if ! skip prof; then
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o prof -- $PROF_CMD --no-alone-pass ) > "$O/prof.log" 2>&1
  echo "rocprof rc=$?" >> "$O/env.log"
  find "$O/prof" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats_proofs_only.csv" \; 2>/dev/null
  grep '^{"metric"' "$O/prof.log" > "$O/bench_under_rocprof.json"
  find "$O/prof" -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
  ( cd /tmp && export TMPDIR=/tmp WSNARK_PROVE_OVERLAP=0 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_serial" -o prof -- $PROF_CMD ) > "$O/prof_serial.log" 2>&1
  echo "rocprof serialised rc=$?" >> "$O/env.log"
  find "$O/prof_serial" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats_proofs_only_serialised.csv" \; 2>/dev/null
  grep '^{"metric"' "$O/prof_serial.log" > "$O/bench_under_rocprof_serialised.json"
  find "$O/prof_serial" -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
fi
# whole-proof instruction budget: SQ_INSTS_VALU over exactly P proofs between two marker launches (tools/proof_counters.py)
if ! skip issue; then
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/issue/pmc_issue" -o pmc -- python "$GRAFT_REPO_ROOT/tools/proof_counters.py" 20 4 ) > "$O/issue.log" 2>&1
  echo "pmc issue rc=$?" >> "$O/env.log"
  python tools/pmc_proof_budget.py "$O/issue" 4 --rates "$O/issue_classes.json" > "$O/proof_issue_budget.json" 2>> "$O/env.log"
  find "$O/issue" -name "*.csv" -size +2M -delete 2>/dev/null
fi
# HBM traffic counters: separate --pmc passes, kernel-trace only (never combined with sys/hip/hsa traces)
if ! skip pmc; then
  for CTR in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_$CTR" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras ) > "$O/pmc_$CTR.log" 2>&1
    echo "pmc $CTR rc=$?" >> "$O/env.log"
  done
  python tools/pmc_summary.py "$O" > "$O/pmc_traffic.json" 2>> "$O/env.log"
  find "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE" -name "*.csv" -size +2M -delete 2>/dev/null
fi
# SQ / GRBM counters behind the issue-bound claim: two passes, proofs only, kernel-trace only
if ! skip sq; then
  I=0
  for GROUP in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
    I=$((I+1))
    ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/sq/pmc_sq$I" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --extras msm,ntt ) > "$O/pmc_sq$I.log" 2>&1
    echo "pmc sq$I rc=$?" >> "$O/env.log"
  done
  python tools/pmc_counters.py "$O/sq" --rates "$O/issue_classes.json" > "$O/pmc_sq_counters.json" 2>> "$O/env.log"
  find "$O/sq" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
fi
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (64-B gathers out of a 1 GiB table; a 16-B-per-lane stream)
if ! skip calib; then
  for CTR in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/calib/pmc_$CTR" -o pmc -- python "$GRAFT_REPO_ROOT/tools/pmc_calibrate.py" ) > "$O/calib_$CTR.log" 2>&1
    echo "calib $CTR rc=$?" >> "$O/env.log"
  done
  python tools/pmc_counters.py "$O/calib" > "$O/pmc_calibration.json" 2>> "$O/env.log"
fi
[ -n "$EXTRA_CMD" ] && ( eval "$EXTRA_CMD" ) > "$O/extra.log" 2>&1
tail -15 "$O/pytest_gpu.txt" 2>/dev/null; head -c 6000 "$O/bench.json" 2>/dev/null; echo; tail -3 "$O/bench.err" 2>/dev/null; cat "$O/env.log"
