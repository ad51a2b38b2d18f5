#!/bin/bash
# stand-alone G1 MSM 2^20 (BASELINE configs[1]) with and without the split plan, and the window-width / task-shape variants
# of the accumulation A/B (small windows = many entries per bucket, buckets cut into lane tasks, partial sums folded by
# wavefront LDS trees: the "LDS-staged segmented reduction" shape) -> gpurun_out/$TAG/
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r03_split}; mkdir -p "$O"
run() { name=$1; shift; env "$@" timeout 300 python bench.py --workload msm --no-cpu-baseline --steps 30 --warmup 10 > "$O/$name.json" 2> "$O/$name.err"; python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][0])
    print(sys.argv[2], "ms", d["ms_per_step"], "Mpts/s", d["value"], "acc_ms", d["roofline"]["avg_launch_ms"] if d.get("roofline") else None, "int_frac", d["roofline_int_alu"]["frac"] if d.get("roofline_int_alu") else None)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run split1 WSNARK_MSM_SPLIT=1
run split0 WSNARK_MSM_SPLIT=0
run split1_again WSNARK_MSM_SPLIT=1
for C in 8 10 12 14; do
  run shape_c${C}_lmax32 WSNARK_MSM_SPLIT=0 WSNARK_MSM_C=$C WSNARK_MSM_LMAX=32
  run shape_c${C}_lmax64 WSNARK_MSM_SPLIT=0 WSNARK_MSM_C=$C WSNARK_MSM_LMAX=64
done
