# prove 2^20: resident key as fixed-base window tables (default) and the order of the sums on the first queue, against
# the plain sections / the round-1 order
run() {
  echo -n "$* : "
  env "$@" python bench.py --no-extras --no-cpu-baseline --steps 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['msm_accumulate_g2_avg_launch_ms'])"
}
for rep in 1 2; do
run WSNARK_KEY_TABLE=1
run WSNARK_PROVE_ORDER=0
run WSNARK_KEY_TABLE=0
run WSNARK_KEY_TABLE=0 WSNARK_PROVE_ORDER=0
done
run WSNARK_MSM_CHUNK=4
run WSNARK_MSM_CHUNK=16
run WSNARK_TAIL_BITS=16
