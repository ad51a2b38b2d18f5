# prove 2^20: first-queue order with the table key (1 = B2, A+B1, C; 2 = B2, A+B1+C), two repetitions; then 2^22
run() {
  echo -n "$* : "
  env "$@" python bench.py --no-extras --no-cpu-baseline --steps 15 $BARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['msm_accumulate_g2_avg_launch_ms'], d.get('parity') or d.get('check'))"
}
for rep in 1 2; do
run WSNARK_PROVE_ORDER=1
run WSNARK_PROVE_ORDER=2
done
BARGS="--prove-log-domain 22 --steps 8 --warmup 2"
run WSNARK_PROVE_ORDER=1
run WSNARK_KEY_TABLE=0
