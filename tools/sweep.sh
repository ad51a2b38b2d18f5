#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "ntt or fft or calc or prove" 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --extras ntt,prove 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['extras']; print(d['value'], d['ms_per_step'], e['ntt_2p22_fwd_plus_inv_ms'], e['ntt_kernel_ms'], e['prove_ms'])"
} > gpurun_out/ntt.txt 2>&1
cat gpurun_out/ntt.txt
