# the probes a full session adds behind tools/gpu_session.sh:  TAG=r04_s7 EXTRA_CMD="TAG=r04_s7 bash tools/session_extras.sh" bash tools/gpu_session.sh
O=gpurun_out/${TAG:-extras}
mkdir -p $O
timeout 400 python tools/node_bench.py 20 20 > $O/node_bench.json 2> $O/node_bench.err
timeout 400 python tools/shard_probe.py 20 8 > $O/shard_probe_2p20.json 2> $O/shard_probe.err
timeout 300 python tools/dist_probe.py 20 > $O/dist_probe.json 2> $O/dist_probe.err
timeout 600 python bench.py --prove-log-domain 22 --no-extras --no-cpu-baseline --steps 10 > $O/bench_2p22.json 2> $O/bench_2p22.err
timeout 900 python bench.py --prove-log-domain 24 --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_2p24.json 2> $O/bench_2p24.err
