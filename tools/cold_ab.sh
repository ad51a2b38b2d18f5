# Cold path A/B on the GPU box: the Node and fresh-Python first calls (tools/node_bench.py), then the bench line's `cold` object with the
# table rows built in short launches (default) and by the one long kernel per section (WSNARK_TABLE_STEPPED=0).
timeout 300 python tools/node_bench.py 20 5 > gpurun_out/nb3.json 2>gpurun_out/nb3.err
python -c "
import json
d=json.load(open('gpurun_out/nb3.json')); n=d.get('node', {}); print({k:n.get(k) for k in ('first_call_ms','first_call_phases_ms','next_calls_ms','key_bytes_call_ms')}); print(d.get('fresh_python_process')); print(d.get('fresh_python_process_rocm_hip_runtime_like_node')); print(d.get('node_error','')[:300])"
timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 > gpurun_out/b3.json 2>gpurun_out/b3.err
python -c "
import json
d=json.load(open('gpurun_out/b3.json')); print(d['value'], d['proofs_match_toxic_waste_closed_form'], json.dumps(d['cold']))"
WSNARK_TABLE_STEPPED=0 timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 > gpurun_out/b3o.json 2>gpurun_out/b3o.err
python -c "
import json
d=json.load(open('gpurun_out/b3o.json')); print('one-kernel', d['value'], json.dumps(d['cold']))"
