#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $1" ; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"; }
{
run "tail kernels with inlined products" X=1
run "tail kernels with called products" WSNARK_LIB=$GRAFT_REPO_ROOT/tools/alt/libwsnark_noinline_tail.so
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
