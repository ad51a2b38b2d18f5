#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/fold.txt
for f in 1 2 4 8 16; do
 for m1 in 0 1; do
  echo "== FOLD=$f M1=$m1" >> gpurun_out/fold.txt
  WSNARK_PROVE_FOLD=$f WSNARK_FOLD_M1=$m1 timeout 300 python tools/trace_prove.py 2>&1 | grep -E "prove ms" | tail -2 >> gpurun_out/fold.txt
 done
done
cat gpurun_out/fold.txt
