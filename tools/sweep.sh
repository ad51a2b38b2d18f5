#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/split.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider > gpurun_out/pytest_prove.log 2>&1
B="python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline"
run() { echo "== $*" ; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['achieved'], d['kernel_ms'])"; }
{
run WSNARK_MSM_SPLIT=1
run WSNARK_MSM_SPLIT=0
run WSNARK_MSM_SPLIT=1
run WSNARK_MSM_SPLIT=0
for sp in 0 1; do echo "== g2 split=$sp"; WSNARK_MSM_SPLIT=$sp timeout 300 python tools/g2bench.py 2>&1 | tail -3; done
} > gpurun_out/split.txt 2>&1
tail -3 gpurun_out/pytest_prove.log; cat gpurun_out/split.txt
