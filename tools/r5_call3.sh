cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
C="PROVE_ORDER=1;PROVE_ORDER=5;PROVE_ORDER=5,TAIL_PRIO=0,PLAN_PRIO=0;PROVE_ORDER=4"
for P in "0 0" "1 0" "0 1" "1 1"; do set -- $P
  WSNARK_S2_PRIO=$1 WSNARK_S3_PRIO=$2 timeout 600 python tools/sched_ab.py 2 "$C" 2> $O/err_$1$2.txt | sed "s/^/s2=$1 s3=$2 /" | tee -a $O/sched_ab.txt
done
export WSNARK_S2_PRIO=1 WSNARK_S3_PRIO=1 WSNARK_PROVE_ORDER=5
T=$O/trace_p11_o5
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$T -o tr -- python $GRAFT_REPO_ROOT/tools/proof_counters.py 20 4 ) > $T.log 2>&1
python tools/trace_timeline.py $T 1 4 > $T.timeline.txt 2>&1
find $T -name "*.csv" -size +1M -delete
