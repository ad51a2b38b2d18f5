cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c18; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 1 0 1 0; do WSNARK_DIST_CALCH_FIRST=$v timeout 300 python tools/dist_probe.py 20 > $O/dist_$v.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('$O/dist_$v.json')); print('first=$v', d['one_call_prove_ms'], d['native_dist_prover_world1_ms'], d['native_minus_one_call_ms'], d['proofs_ok'])"; done
