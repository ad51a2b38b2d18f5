#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $*" ; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(d['value'], d['ms_per_step'], k['msm_accumulate_g1'], k['msm_chunks'], k['msm_tree'])"; }
{
run X=1
run WSNARK_LIB=$GRAFT_REPO_ROOT/tools/alt/libwsnark_mulsub_inline.so
run X=2
run WSNARK_LIB=$GRAFT_REPO_ROOT/tools/alt/libwsnark_mulsub_inline.so
} > gpurun_out/inl.txt 2>&1
cat gpurun_out/inl.txt
