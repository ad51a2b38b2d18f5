#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
python tools/g2bench.py
for f in tools/alt/*.so; do WSNARK_LIB=$GRAFT_REPO_ROOT/$f python tools/g2bench.py; done
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
