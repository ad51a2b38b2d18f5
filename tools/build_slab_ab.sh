# Cold path with smaller slabs of the background table build (fewer workgroups of the build resident at a time): proofs beside it vs its duration
for l in ${LANES:-65536 262144}; do
  WSNARK_TABLE_SLAB_LANES=$l timeout 150 python bench.py --no-extras --no-cpu-baseline --no-alone-pass --steps 5 --warmup 2 2>/dev/null > /tmp/slab_$l.json
  python - "$l" <<'P'
import json,sys
l=sys.argv[1]
d=json.loads(open('/tmp/slab_%s.json'%l).read().strip().splitlines()[-1]); c=d["cold"]
print(json.dumps({"slab_lanes": int(l), "steady_ms": d["value"], "load_ms": c["key_load_ms"]["total"], "first_proof_ms": c["first_proof_ms"], "next_proofs_ms": c["next_proofs_ms"], "tables_ready_ms_after_load": c["tables_ready_ms_after_the_load_returned"], "build_ms": c["key_load_ms"]["table_build"]}))
P
done
