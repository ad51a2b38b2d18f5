#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
{
for i in 1 2 3; do $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2
timeout 300 python tools/g2bench.py 2>&1 | tail -1
} > gpurun_out/conv.txt 2>&1
cat gpurun_out/conv.txt
