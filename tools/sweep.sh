#!/bin/bash
# A/B template for one gpurun call: runs the headline bench under a few environment settings and prints
# Mpoints/s, ms per MSM and the per-kernel breakdown for each.   gpurun -- 'bash tools/sweep.sh'
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"; }
{
run X=default
run WSNARK_MSM_SORT=cub
run WSNARK_MSM_LO_BITS=7
run WSNARK_FIELD=32
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
