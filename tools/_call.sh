cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c29; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do timeout 400 python tools/node_bench.py 20 20 > $O/node_$i.json 2>> $O/err.txt; python -c "
import json; n=json.load(open('$O/node_$i.json')); print({k:n['node'][k] for k in ('key_bytes_call_ms','key_bytes_call_trusted_ms','key_handle_call_ms','pinned_witness_call_ms','first_call_ms')}, n['ctypes_host_witness_ms'])"; done
