#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $*" ; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(d['value'], d['ms_per_step'], k)"; }
{
run X=1
run WSNARK_MSM_C=15
run WSNARK_MSM_C=14
run WSNARK_MSM_CHUNK=4
run WSNARK_MSM_CHUNK=16
run X=2
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
