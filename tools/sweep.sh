#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $*" ; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(d['value'], d['ms_per_step'], 'count', k.get('msm_presort_count'), 'scatter', k.get('msm_presort_scatter'), 'bins', k.get('msm_presort_bins'), 'acc', k.get('msm_accumulate_g1'))"; }
{
run WSNARK_MSM_STAGE=1
run WSNARK_MSM_STAGE=0
run WSNARK_MSM_STAGE=1 WSNARK_MSM_LO_BITS=8
run WSNARK_MSM_STAGE=0 WSNARK_MSM_LO_BITS=8
run WSNARK_MSM_STAGE=1 WSNARK_MSM_LO_BITS=9
run WSNARK_MSM_STAGE=1 WSNARK_MSM_TILE=512 WSNARK_MSM_TILE_THREADS=512
run WSNARK_MSM_STAGE=1 WSNARK_MSM_TILE=1536 WSNARK_MSM_TILE_THREADS=768 WSNARK_MSM_LO_BITS=8
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "msm or multiexp or prove" 2>&1 | tail -2
} > gpurun_out/stage.txt 2>&1
cat gpurun_out/stage.txt
