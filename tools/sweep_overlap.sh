for rep in 1 2; do
for p in 0 1; do for o in 1 2; do
  echo -n "rep=$rep S2_PRIO=$p OVERLAP=$o : "
  WSNARK_S2_PRIO=$p WSNARK_PROVE_OVERLAP=$o python bench.py --no-extras --no-cpu-baseline --steps 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['msm_accumulate_g2_avg_launch_ms'])"
done; done; done
