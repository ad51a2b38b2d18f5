#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
( time python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" ) 2>&1 | tail -5
( time python bench.py > gpurun_out/bench_default.json ) 2>&1 | tail -4
python -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['extras']['prove_ms'], d['cpu_baseline']['value'])"
} > gpurun_out/smoke.txt 2>&1
cat gpurun_out/smoke.txt
