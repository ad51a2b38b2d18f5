# prove 2^20 on the table key: scatter tile and task-cap sweeps
run() {
  echo -n "$* : "
  env "$@" python bench.py --no-extras --no-cpu-baseline --steps 15 $BARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_proof']; print(d['value'], d['roofline']['avg_launch_ms'], d['msm_accumulate_g2_avg_launch_ms'], 'scatter', k['msm_presort_scatter'], 'bins', k['msm_presort_bins'], 'combine', k['msm_combine'], d['proofs_match_toxic_waste_closed_form'])"
}
run WSNARK_MSM_TILE=1024
run WSNARK_MSM_TILE=2048
run WSNARK_MSM_TILE=4096
run WSNARK_MSM_TILE=8192
run WSNARK_MSM_LMAX_X4=6
run WSNARK_MSM_LMAX_X4=12
run WSNARK_MSM_LMAX_X4=16
run WSNARK_MSM_TILE=1024
