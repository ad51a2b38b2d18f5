cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c7; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
C="PROVE_ORDER=1;PROVE_ORDER=7,PROVE_GATE=0;PROVE_ORDER=7,PROVE_GATE=1;PROVE_ORDER=7,PROVE_GATE=2"
timeout 600 python tools/sched_ab.py 4 "$C" 2> $O/err.txt | tee $O/sched_ab.jsonl
for G in 1 2; do
export WSNARK_PROVE_ORDER=7 WSNARK_PROVE_GATE=$G
T=$O/trace_o7g$G
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$T -o tr -- python $GRAFT_REPO_ROOT/tools/proof_counters.py 20 4 ) > $T.log 2>&1
python tools/trace_timeline.py $T 1 4 > $T.timeline.txt 2>&1
find $T -name "*.csv" -size +1M -delete
done
