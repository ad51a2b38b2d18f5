#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "concurrent" 2>&1 | tail -2; done > gpurun_out/conc.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2 >> gpurun_out/conc.txt
timeout 300 python tools/trace_prove.py 2>&1 | grep "prove ms" | tail -2 >> gpurun_out/conc.txt
cat gpurun_out/conc.txt
