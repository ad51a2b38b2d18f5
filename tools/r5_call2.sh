cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "msm_golden or msm_vs_oracle or all_same_point or grouping_variants or proofs_golden or adversarial or prove_small or degenerate or toxic_waste" > $O/pytest_subset.txt 2>&1
tail -5 $O/pytest_subset.txt
C="PROVE_CALCH_FIRST=0,COMBINE_FUSED=0,TAIL_FUSE_ROWS=0,PROVE_ORDER=1"
C="$C;PROVE_CALCH_FIRST=1,COMBINE_FUSED=0,TAIL_FUSE_ROWS=0,PROVE_ORDER=1"
C="$C;PROVE_CALCH_FIRST=1,COMBINE_FUSED=1,TAIL_FUSE_ROWS=0,PROVE_ORDER=1"
C="$C;PROVE_CALCH_FIRST=1,COMBINE_FUSED=1,TAIL_FUSE_ROWS=1,PROVE_ORDER=1"
C="$C;PROVE_CALCH_FIRST=1,COMBINE_FUSED=1,TAIL_FUSE_ROWS=1,PROVE_ORDER=4"
C="$C;PROVE_CALCH_FIRST=1,COMBINE_FUSED=1,TAIL_FUSE_ROWS=1,PROVE_ORDER=5"
C="$C;PROVE_CALCH_FIRST=1,COMBINE_FUSED=1,TAIL_FUSE_ROWS=1,PROVE_ORDER=5,TAIL_PRIO=0,PLAN_PRIO=0"
C="$C;PROVE_CALCH_FIRST=1,COMBINE_FUSED=1,TAIL_FUSE_ROWS=1,PROVE_ORDER=2"
timeout 900 python tools/sched_ab.py 3 "$C" > $O/sched_ab.jsonl 2> $O/sched_ab.err
for ORD in 1 5; do
  export WSNARK_PROVE_ORDER=$ORD
  T=$O/trace_o$ORD
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$T -o tr -- python $GRAFT_REPO_ROOT/tools/proof_counters.py 20 4 ) > $T.log 2>&1
  python tools/trace_timeline.py $T 1 4 > $T.timeline.txt 2>&1
  find $T -name "*.csv" -size +1M -delete
done
unset WSNARK_PROVE_ORDER
cat $O/sched_ab.jsonl; tail -5 $O/sched_ab.err
