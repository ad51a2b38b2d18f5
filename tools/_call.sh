cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c19; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 1 2 3; do timeout 300 python tools/dist_probe.py 20 > $O/dist_$v.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('$O/dist_$v.json')); print('run $v', d['one_call_prove_ms'], d['native_dist_prover_world1_ms'], d['native_minus_one_call_ms'], d['proofs_ok'], d['native_kernel_ms_per_proof'])"; done
timeout 900 python -m pytest tests -m gpu -q -x -k "dist or multi or group or config5" 2>&1 | tail -3
