#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "adversarial" --durations=5 > gpurun_out/adv.txt 2>&1
tail -25 gpurun_out/adv.txt
