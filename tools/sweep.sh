#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "prove or proof or shard" --timeout 300 -p no:cacheprovider > gpurun_out/pytest_prove.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --extras prove > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/ptrace" -o pt -- python "$GRAFT_REPO_ROOT/tools/trace_prove.py" ) > gpurun_out/ptrace.log 2>&1
python tools/prove_timeline.py gpurun_out/ptrace > gpurun_out/prove_timeline.txt 2>&1
WSNARK_TRACE=1 timeout 300 python tools/trace_prove.py > gpurun_out/host_trace.txt 2>&1
find gpurun_out/ptrace -name "*.csv" -size +1M -delete
tail -3 gpurun_out/pytest_prove.log; python -c "
import json; d=json.load(open('gpurun_out/bench_quick.json')); print(d['value'], d['ms_per_step'], d['extras'].get('prove_ms'), d['extras'].get('prove_matches_toxic_waste_closed_form'))"
tail -3 gpurun_out/prove_timeline.txt; tail -25 gpurun_out/host_trace.txt
