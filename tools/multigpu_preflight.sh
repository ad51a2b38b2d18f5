#!/bin/bash
# One command for the first box with >= 2 GPUs: everything about N > 1 that has never run on RCCL, with its
# records under gpurun_out/<TAG>/ (copy what should be judged into profiles/).
#   TAG=r05_multi tools/multigpu_preflight.sh            (from the repository root; needs torch.distributed + RCCL)
# 1. the two nccl tests that skip themselves on a one-GPU box (sharded MSM + prove over RCCL; entry points from another thread
#    on device 1)
# 2. bench.py --gpus 2 / 4 / 8 at 2^20 (strong scaling of the headline proof: points-sharded key, distributed CALC_H,
#    wsnark_groth16_prove_dist), one JSON line each -- the line says which orchestration ran and whether one fell through
# 3. BASELINE config 5: bench.py --prove-log-domain 24 on every GPU of the box, and on ONE GPU for the ratio
# 4. the all-gather / all-to-all transport on its own (tools/nccl_allgather_check.py)
# 5. ONE process over all GPUs (wsnark_group_*): tests/test_gpu_group.py, bench.py --single-process, the Node drop-in (NODE_BENCH_DEVICES)
# 6. (round 6) key FILES: tests/test_gpu_key_file.py (2^22 through the WSNARK64 container: whole, group, shards, Node), config 5 with
#    --key-file auto (rank 0 writes the 9.4 GB container once, every rank maps it and reads its 1 / N), the Node drop-in proving 2^24
#    from the file over all devices; summary.txt ends with one line comparing per-process ranks, one process, and Node at 2^20
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
TAG=${TAG:-multi}
O=gpurun_out/$TAG
mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NG=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "GPUs visible: $NG" | tee "$O/env.log"
if [ "$NG" -lt 2 ]; then echo "needs at least two GPUs" | tee -a "$O/env.log"; exit 2; fi
timeout 1800 python -m pytest tests/test_gpu_multi.py -m gpu -q --timeout 900 -p no:cacheprovider > "$O/pytest_gpu_multi.txt" 2>&1
echo "pytest multi rc=$?" | tee -a "$O/env.log"
PORT=29700
run_bench() {   # run_bench <n> <extra args...>
  local n=$1; shift
  PORT=$((PORT + 1))
  if [ "$n" -eq 1 ]; then timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 "$@"
  else timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus "$n" --steps 20 --warmup 5 "$@"
  fi
}
run_bench 1 > "$O/bench_n1.json" 2> "$O/bench_n1.err"; echo "bench n=1 rc=$?" | tee -a "$O/env.log"
for n in 2 4 8; do
  [ "$n" -le "$NG" ] || continue
  run_bench $n > "$O/bench_n$n.json" 2> "$O/bench_n$n.err"; echo "bench n=$n rc=$?" | tee -a "$O/env.log"
done
run_bench "$NG" --prove-log-domain 24 --no-extras --key-file auto > "$O/bench_2p24_n$NG.json" 2> "$O/bench_2p24_n$NG.err"; echo "bench 2^24 n=$NG (key file) rc=$?" | tee -a "$O/env.log"
run_bench 1 --prove-log-domain 24 --no-extras --no-cpu-baseline > "$O/bench_2p24_n1.json" 2> "$O/bench_2p24_n1.err"; echo "bench 2^24 n=1 rc=$?" | tee -a "$O/env.log"
# 5. (round 5) several GPUs in ONE process: the group tests on real peers, bench.py --single-process, and the Node drop-in over all devices
timeout 1800 python -m pytest tests/test_gpu_group.py -m gpu -q --timeout 900 -p no:cacheprovider > "$O/pytest_gpu_group.txt" 2>&1
echo "pytest group rc=$?" | tee -a "$O/env.log"
for n in 2 4 8; do
  [ "$n" -le "$NG" ] || continue
  timeout 1800 python bench.py --gpus "$n" --single-process --steps 20 --warmup 5 > "$O/bench_group_n$n.json" 2> "$O/bench_group_n$n.err"; echo "bench group n=$n rc=$?" | tee -a "$O/env.log"
done
DEVS=$(python -c "print(','.join(str(i) for i in range($NG)))")
NODE_BENCH_DEVICES=$DEVS timeout 900 python tools/node_bench.py 20 20 > "$O/node_bench_group.json" 2> "$O/node_bench_group.err"; echo "node group rc=$?" | tee -a "$O/env.log"
# 6. key files: the 2^22 container tests, then 2^24 from the file through Node over all devices
timeout 1800 python -m pytest tests/test_gpu_key_file.py -m gpu -q --timeout 1200 -p no:cacheprovider > "$O/pytest_gpu_key_file.txt" 2>&1
echo "pytest key file rc=$?" | tee -a "$O/env.log"
NODE_BENCH_DEVICES=$DEVS timeout 1800 python tools/prove_big.py --node 24 > "$O/prove_big_node_2p24_all_devices.json" 2> "$O/prove_big_node.err"; echo "node 2^24 from the key file rc=$?" | tee -a "$O/env.log"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((PORT + 7)) tools/nccl_allgather_check.py > "$O/nccl_allgather_check.txt" 2>&1
echo "transport check rc=$?" | tee -a "$O/env.log"
grep -h '^{"metric"' "$O"/bench_n*.json "$O"/bench_group_n*.json "$O"/bench_2p24_*.json 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print(d['metric'], 'n_gpus', d['n_gpus'], 'ms', d['value'], '|', d['config']['parallelism'][:140])
" | tee "$O/summary.txt"
# one line: the three hosts of the same 2^20 proof over all GPUs of the box
python - "$O" "$NG" <<'PY' | tee -a "$O/summary.txt"
import json, sys
o, ng = sys.argv[1], sys.argv[2]
def val(f, key="value"):
    try:
        for l in open(f):
            if l.startswith("{"):
                return json.loads(l).get(key)
    except Exception:
        return None
per_proc, one_proc = val("%s/bench_n%s.json" % (o, ng)), val("%s/bench_group_n%s.json" % (o, ng))
node = (val("%s/node_bench_group.json" % o, "node") or {}).get("key_bytes_call_ms")
print("2^20 proof over %s GPUs: per-process ranks (RCCL) %s ms | one process (wsnark_group_*) %s ms | Node drop-in over all devices %s ms | one GPU %s ms"
      % (ng, per_proc, one_proc, node, val("%s/bench_n1.json" % o)))
PY
