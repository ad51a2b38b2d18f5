#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 600 python tools/worstcase.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --extras skewed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['extras']['msm_circuit_like_scalars_ms'])"
} > gpurun_out/worst.txt 2>&1
cat gpurun_out/worst.txt
