#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/overlap3.txt
for cfg in "1 1" "4 1" "4 0" "1 1" "4 1"; do
  set -- $cfg
  echo "== OVERLAP=$1 S2_PRIO=$2" >> gpurun_out/overlap3.txt
  WSNARK_PROVE_OVERLAP=$1 WSNARK_S2_PRIO=$2 timeout 300 python tools/trace_prove.py 2>&1 | grep -E "prove ms" | tail -2 >> gpurun_out/overlap3.txt
done
WSNARK_PROVE_OVERLAP=4 WSNARK_TIMELINE=1 timeout 300 python tools/trace_prove.py 2>&1 | grep -E "timeline" | grep -v ntt_pass > gpurun_out/tl4.txt
cat gpurun_out/overlap3.txt; cat gpurun_out/tl4.txt
