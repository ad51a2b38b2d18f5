cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c20; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
P="timeout 200 python tools/msm_probe.py --reps 20"
$P > $O/probe.txt 2> $O/err.txt
for tb in 11 12 13 14; do for ch in 4 8; do $P --set TAIL_BITS=$tb --set MSM_CHUNK=$ch >> $O/probe.txt 2>> $O/err.txt; done; done
python - <<'PY'
import json
for l in open('gpurun_out/r05_c20/probe.txt'):
    d=json.loads(l); r=d["resident"]; k=r["kernels_us"]
    print(d["tag"], "per_call", d["per_call"]["ms"], "resident", r["ms"], r["same"], {x:k[x] for x in ("msm_chunks","msm_tree","msm_rows") if x in k})
PY
