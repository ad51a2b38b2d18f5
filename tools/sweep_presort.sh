#!/bin/bash
# grouping-pass geometry sweep (G1 / G2 MSM 2^20 via tools/ab_msm.py): scatter tile size x low bucket bits per bin
cd "$(dirname "$0")/.."
for tile in 1024 2048 4096; do for lo in 8 9 10; do
  echo -n "TILE=$tile LO_BITS=$lo : "
  WSNARK_MSM_TILE=$tile WSNARK_MSM_LO_BITS=$lo python tools/ab_msm.py x=wasmsnark_amd/libwsnark.so | python -c "
import sys,json
r=json.loads(sys.stdin.read())['r']['g1']; k=r['kernel_ms']
print(r['ms'], 'count', k['msm_presort_count'], 'scatter', k['msm_presort_scatter'], 'bins', k['msm_presort_bins'], 'plan', k['msm_plan'], 'acc', k['msm_accumulate_g1'])"
done; done
