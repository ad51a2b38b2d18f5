#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python tools/prove_big.py 22 24 > gpurun_out/prove_big.txt 2> gpurun_out/prove_big.err
cat gpurun_out/prove_big.txt; tail -5 gpurun_out/prove_big.err
