cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c13; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_c13/bench.json'))
for k in ("value","drop_in_call_ms","js_drop_in_call_ms","serialised_one_queue_ms_per_proof"): print(k, d.get(k))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","avg_launch_ms")})
print("roofline_proof", d.get("roofline_proof",{}) and {k:d["roofline_proof"].get(k) for k in ("frac","issue_ms")})
print("roofline_c3", d.get("roofline_c3",{}).get("frac"), d.get("roofline_c3",{}).get("ms"))
print("kernel_ms_per_proof", d.get("kernel_ms_per_proof"))
ex=d.get("extras",{})
print("msm", {k:v for k,v in ex.get("g1_msm_2p20",{}).items() if not isinstance(v,dict)})
print("msm kernels", ex.get("g1_msm_2p20",{}).get("kernel_ms_per_msm"))
print("node", ex.get("node_drop_in"))
print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k in ("value","cores","eight_threads_sample")})
print("cold", {k:d["cold"].get(k) for k in ("key_load_plus_first_proof_ms","first_proof_ms")} if d.get("cold") else None)
print("table_memory", d.get("table_memory"))
PY
