#!/bin/bash
# One gpurun call: environment probe, instruction-rate microbench, GPU parity tests, bench, rocprof.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
  echo "== env"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
  nproc; free -g | head -2; node --version 2>&1; python --version
} > gpurun_out/env.log 2>&1
timeout 300 ./tools/microbench > gpurun_out/microbench.jsonl 2>&1
echo "microbench rc=$?" >> gpurun_out/env.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/env.log
timeout 1200 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" >> gpurun_out/env.log
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r01 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --extras ntt ) > gpurun_out/prof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/env.log
# HBM traffic counters: separate --pmc passes, kernel-trace only (never combined with sys/hip/hsa traces)
for CTR in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$CTR" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --extras ntt ) > gpurun_out/pmc_$CTR.log 2>&1
  echo "pmc $CTR rc=$?" >> gpurun_out/env.log
done
python tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary.json 2>> gpurun_out/env.log
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +2M -delete 2>/dev/null
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.json | head -c 3000; cat gpurun_out/env.log
