#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/hostpath.py > gpurun_out/hostpath.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2 >> gpurun_out/hostpath.txt
tail -8 gpurun_out/hostpath.txt
