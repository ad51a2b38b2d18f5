#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $*" ; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(d['value'], d['ms_per_step'], 'count', k.get('msm_presort_count'), 'scatter', k.get('msm_presort_scatter'), 'bins', k.get('msm_presort_bins'), 'acc', k.get('msm_accumulate_g1'))"; }
{
for lo in 7 8 9 10; do
 for bthr in 256 512 1024; do
   run WSNARK_MSM_LO_BITS=$lo WSNARK_MSM_TILE=1024 WSNARK_MSM_TILE_THREADS=1024 WSNARK_MSM_BIN_THREADS=$bthr
 done
done
run WSNARK_MSM_LO_BITS=9 WSNARK_MSM_TILE=2048 WSNARK_MSM_TILE_THREADS=1024
run WSNARK_MSM_LO_BITS=8 WSNARK_MSM_TILE=512 WSNARK_MSM_TILE_THREADS=512
run WSNARK_MSM_LO_BITS=7 WSNARK_MSM_TILE=512 WSNARK_MSM_TILE_THREADS=512
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
