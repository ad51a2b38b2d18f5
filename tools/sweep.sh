#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/overlap.txt
for ov in 1 3 2 0 3 1; do
  echo "== OVERLAP=$ov" >> gpurun_out/overlap.txt
  WSNARK_PROVE_OVERLAP=$ov timeout 300 python tools/trace_prove.py 2>&1 | grep -E "prove ms" | tail -2 >> gpurun_out/overlap.txt
done
cat gpurun_out/overlap.txt
