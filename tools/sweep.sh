#!/bin/bash
# parameter sweep on the headline workload (G1 MSM 2^20)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $1" ; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"; }
{
run "default" X=1
run "persist 3 blocks/CU" WSNARK_MSM_PERSIST=3
run "persist 4 blocks/CU" WSNARK_MSM_PERSIST=4
run "persist 6 blocks/CU" WSNARK_MSM_PERSIST=6
run "chunk4" WSNARK_MSM_CHUNK=4
run "chunk16" WSNARK_MSM_CHUNK=16
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
