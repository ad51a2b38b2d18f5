#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 300 python tools/g2bench.py 2>&1 | tail -1
timeout 300 python tools/trace_prove.py 2>&1 | grep -E "prove ms" | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2
} > gpurun_out/g2.txt 2>&1
cat gpurun_out/g2.txt
