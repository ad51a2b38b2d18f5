cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_c1; O=gpurun_out/r05_c1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/sched_ab.py 3 > $O/sched_ab.jsonl 2> $O/sched_ab.err
for CFG in "0 0 1" "3 3 5"; do set -- $CFG
  export WSNARK_TAIL_PRIO=$1 WSNARK_PLAN_PRIO=$2 WSNARK_PROVE_ORDER=$3
  T=$O/trace_t$1_o$3
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$T -o tr -- python $GRAFT_REPO_ROOT/tools/proof_counters.py 20 4 ) > $T.log 2>&1
  python tools/trace_timeline.py $T 1 4 > $T.timeline.txt 2>&1
  find $T -name "*.csv" -size +1M -delete
done
unset WSNARK_TAIL_PRIO WSNARK_PLAN_PRIO WSNARK_PROVE_ORDER
WSNARK_TRACE=1 timeout 300 python tools/trace_prove.py > $O/trace_prove.txt 2>&1
cat $O/sched_ab.jsonl; tail -5 $O/sched_ab.err
