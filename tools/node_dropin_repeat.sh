# Runs tests/node_dropin_check.js 60 times per configuration on the GPU box and counts the runs that fail: the suite loads many small
# keys back to back and proves while their table rows are still being built -- what exposed the wrong proofs of the stream-ordered
# allocator version (prove.hip: 3 to 36 failures in 60; with plain allocations 0 in 120).
python -c "
import sys; sys.path.insert(0,'tests')
import test_node_dropin as t
t._build_addon()
" 2>&1 | tail -1
# CONFIGS: space-separated VAR=value settings, one run series each ("default" = none);  RUNS: runs per series
for cfg in ${CONFIGS:-default WSNARK_TABLE_STEPPED=0}; do
  [ "$cfg" = default ] && cfg=WSNARK_NOOP=1
  ok=0; bad=0
  for i in $(seq 1 ${RUNS:-60}); do
    env $cfg timeout 120 node tests/node_dropin_check.js > /tmp/nd.out 2>&1; rc=$?
    if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "[$cfg] run $i rc=$rc $(grep -o 'NODE_DROPIN_FAIL.*' /tmp/nd.out | head -1 | cut -c1-160)"; fi
  done
  echo "[$cfg] ok=$ok bad=$bad"
done
