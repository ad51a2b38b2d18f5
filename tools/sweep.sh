#!/bin/bash
# parameter sweep on the headline workload (G1 MSM 2^20): window size, task cap, accumulate occupancy
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $1" ; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"; }
{
run "default (waves4,c16,lmax2x)" X=1
run "c15" WSNARK_MSM_C=15
run "c14" WSNARK_MSM_C=14
run "c13" WSNARK_MSM_C=13
run "lmax 1x" WSNARK_MSM_LMAX_X4=4
run "lmax 1.5x" WSNARK_MSM_LMAX_X4=6
run "lmax 3x" WSNARK_MSM_LMAX_X4=12
run "lmax 4x" WSNARK_MSM_LMAX_X4=16
run "field32" WSNARK_FIELD=32
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
