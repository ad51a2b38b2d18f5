# the probes a full session adds behind tools/gpu_session.sh:  TAG=r04_s7 EXTRA_CMD="TAG=r04_s7 bash tools/session_extras.sh" bash tools/gpu_session.sh
O=gpurun_out/${TAG:-extras}
mkdir -p $O
timeout 400 python tools/node_bench.py 20 20 > $O/node_bench.json 2> $O/node_bench.err
timeout 400 python tools/shard_probe.py 20 8 > $O/shard_probe_2p20.json 2> $O/shard_probe.err
timeout 300 python tools/dist_probe.py 20 > $O/dist_probe.json 2> $O/dist_probe.err
timeout 600 python bench.py --prove-log-domain 22 --no-extras --no-cpu-baseline --steps 10 > $O/bench_2p22.json 2> $O/bench_2p22.err
timeout 900 python bench.py --prove-log-domain 24 --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_2p24.json 2> $O/bench_2p24.err
# round 5: one sum on resident bases beside the per-call sum (per-kernel times), and two contexts of ONE process on the one GPU
timeout 300 python tools/msm_probe.py > $O/msm_probe.json 2> $O/msm_probe.err
timeout 600 python bench.py --gpus 2 --single-process --group-devices 0,0 --steps 10 --warmup 3 > $O/bench_group_2x_same_gpu.json 2> $O/bench_group.err
NODE_BENCH_DEVICES=0,0 timeout 400 python tools/node_bench.py 20 10 > $O/node_bench_group.json 2> $O/node_bench_group.err
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
