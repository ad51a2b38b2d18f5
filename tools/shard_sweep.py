#!/usr/bin/env python3
"""A rank's share of an 8-way points-sharded 2^20 key (rank 3: the four witness sums), swept over how the sums are laid out on
the chip: G1 accumulations batched in one launch or back to back, task length cap, queue order.   python tools/shard_sweep.py"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, 20, n_public=5, seed=1)
sec, _ = circ.build_sections()
wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
k = bn.load_key(sections=sec, shard=(3, 8))
NAMES = ("MSM_BATCH_ACC", "MSM_LMAX", "PROVE_ORDER", "MSM_CHUNK", "TAIL_BITS")
f = lambda: bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(3, 8), skip_h=True)
ref = f()
def t(n=30):
    for _ in range(3): out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 4), out
for cfg in ({}, {"MSM_BATCH_ACC": 1}, {"MSM_LMAX": 16}, {"MSM_LMAX": 16, "MSM_BATCH_ACC": 1}, {"MSM_LMAX": 10}, {"MSM_LMAX": 10, "MSM_BATCH_ACC": 1}, {"MSM_LMAX": 8, "MSM_BATCH_ACC": 1},
            {"PROVE_ORDER": 1}, {"TAIL_BITS": 13}, {"TAIL_BITS": 13, "MSM_BATCH_ACC": 1}, {"TAIL_BITS": 13, "MSM_LMAX": 16, "MSM_BATCH_ACC": 1}, {}):
    for n in NAMES: bn.lib.tune(n, cfg.get(n))
    ms, out = t()
    print(json.dumps({"tuning": cfg, "four_witness_sums_ms": ms, "ok": out == ref}), flush=True)
