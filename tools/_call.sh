cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c28; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
timeout 300 python tools/lib_ab.py >> $O/ab.txt 2>> $O/err.txt
timeout 300 python tools/lib_ab.py wasmsnark_amd/libwsnark_b.so >> $O/ab.txt 2>> $O/err.txt
done
python - <<'PY'
import json
for l in open('gpurun_out/r05_c28/ab.txt'):
    d=json.loads(l); print(d["lib"], d["ok"], "2q", d["two_queues_ms"], "1q", d["one_queue_ms"], "g1 acc", d["alone_ms_per_proof"]["msm_accumulate_g1"])
PY
