#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$1', d['value'], d['ms_per_step'], k['msm_accumulate_g1'], k['msm_chunks'], k['msm_tree'])"; }
{
for i in 1 2 3 4 5 6; do
run X=inline
run WSNARK_LIB=$GRAFT_REPO_ROOT/tools/alt/libwsnark_tail_calls.so
done
} > gpurun_out/tailvar.txt 2>&1
cat gpurun_out/tailvar.txt
