#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/overlap2.txt
for cfg in "1 0" "3 1" "1 1" "3 0" "2 1" "3 1" "1 0"; do
  set -- $cfg
  echo "== OVERLAP=$1 S2_PRIO=$2" >> gpurun_out/overlap2.txt
  WSNARK_PROVE_OVERLAP=$1 WSNARK_S2_PRIO=$2 timeout 300 python tools/trace_prove.py 2>&1 | grep -E "prove ms" | tail -2 >> gpurun_out/overlap2.txt
done
cat gpurun_out/overlap2.txt
