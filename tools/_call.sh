cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c15; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
P="timeout 300 python tools/msm_probe.py"
$P > $O/probe.txt 2> $O/err.txt
$P --set PS_ONCE=0 >> $O/probe.txt 2>> $O/err.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "resident_bases or msm" 2>&1 | tail -3 > $O/pytest.txt
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2>> $O/err.txt
tail -5 $O/err.txt; cat $O/pytest.txt
python - <<'PY'
import json
for l in open('gpurun_out/r05_c15/probe.txt'):
    d=json.loads(l)
    for k in ("per_call","resident"):
        print(d["tag"],k,d[k]["ms"],d[k]["same"],d[k]["kernels_us"])
d=json.load(open('gpurun_out/r05_c15/bench.json'))
print(d["value"], d.get("serialised_one_queue_ms_per_proof"), d.get("kernel_ms_per_proof"))
PY
