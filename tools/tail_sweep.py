#!/usr/bin/env python3
"""Reduction-tail geometry A/B in ONE process (the library's wsnark_tuning_set switches): chunk size, piece size, rows folded on the
GPU or on the host, G2 tail on lane pairs or one lane per point -- on the workloads where the tail shows: a rank's share of an
8-way points-sharded 2^20 key, a 2^16 proof, the stand-alone G1 / G2 sums, and the whole 2^20 proof.  One JSON line per setting.
    python tools/tail_sweep.py [reps]"""
import json
import os
import sys
import time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import wasmsnark_amd
from wasmsnark_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bn = wasmsnark_amd.build(device=0)
NAMES = ("MSM_CHUNK", "TAIL_BITS", "TAIL_BITS_W", "TAIL_REDUCE", "G2_TAIL_PAIR")
OLD = {"MSM_CHUNK": 8, "TAIL_BITS": 15, "TAIL_REDUCE": 0, "G2_TAIL_PAIR": 0}       # the round-3 tail


def apply(cfg):
    for n in NAMES:
        bn.lib.tune(n, cfg.get(n))


def t(f, n=reps):
    for _ in range(3):
        out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = f()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 4), out


def sweep(what, f, want, cfgs):
    for name, cfg in cfgs:
        apply(cfg)
        ms, out = t(f)
        print(json.dumps({"workload": what, "setting": name, "tuning": cfg, "ms": ms, "ok": bool(want is None or out == want)}), flush=True)
    apply({})


r, s = bytes(range(32)), bytes(range(32, 64))
# ---- a rank's share of an 8-way points-sharded 2^20 key (rank 3), the four witness sums alone and with the whole CALC_H
circ = synth.NativeCircuit(bn.lib, 20, n_public=5, seed=1)
sec, _ = circ.build_sections()
wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
k = bn.load_key(sections=sec, shard=(3, 8))
base = None
shard_cfgs = [("round-3 tail", OLD), ("round-4 default", {}), ("default, G2 one lane per point", {"G2_TAIL_PAIR": 0}),
              ("default, rows folded on the host", {"TAIL_REDUCE": 0}), ("chunks of 8", {"MSM_CHUNK": 8}), ("chunks of 2", {"MSM_CHUNK": 2}),
              ("pieces of 2^10", {"TAIL_BITS": 10}), ("pieces of 2^12", {"TAIL_BITS": 12}), ("pieces of 2^13", {"TAIL_BITS": 13}),
              ("chunks of 2, pieces of 2^10", {"MSM_CHUNK": 2, "TAIL_BITS": 10})]
apply(OLD)
ref = bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(3, 8), skip_h=True)
sweep("rank 3 of 8, 2^20 key: four witness sums", lambda: bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(3, 8), skip_h=True), ref, shard_cfgs)
k.free()
# ---- the whole 2^20 proof
key = bn.load_key(sections=sec)
want = circ.expected_proof(r, s)
sweep("prove 2^20", lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s), want,
      [("round-3 tail", OLD), ("round-4 default", {}), ("default, G2 one lane per point", {"G2_TAIL_PAIR": 0}), ("default, rows folded on the host", {"TAIL_REDUCE": 0}),
       ("pieces of 2^13", {"TAIL_BITS": 13}), ("pieces of 2^12", {"TAIL_BITS": 12}), ("chunks of 4", {"MSM_CHUNK": 4}), ("round-3 tail (again)", OLD), ("round-4 default (again)", {})])
key.free(); del d_w
# ---- a 2^16 proof
c16 = synth.NativeCircuit(bn.lib, 16, n_public=5, seed=1)
sec16, _ = c16.build_sections()
w16 = c16.witness_bin()
d16 = torch.frombuffer(bytearray(w16), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
k16 = bn.load_key(sections=sec16)
sweep("prove 2^16", lambda: bn.groth16GenProof_dev(d16.data_ptr(), len(w16), k16, r=r, s=s), c16.expected_proof(r, s),
      [("round-3 tail", OLD), ("round-4 default", {}), ("default, G2 one lane per point", {"G2_TAIL_PAIR": 0}), ("chunks of 2", {"MSM_CHUNK": 2}), ("pieces of 2^10", {"TAIL_BITS": 10}),
       ("chunks of 8", {"MSM_CHUNK": 8})])
k16.free()
# ---- stand-alone sums (caller-supplied points, per-window plans)
rng = np.random.default_rng(1234)
for g, logn in ((1, 20), (2, 18)):
    n = 1 << logn
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x1F
    pts = bn.mul_base(g, ks.tobytes())
    d_s = torch.from_numpy(sc.reshape(-1)).cuda()
    d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
    call = (lambda: bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)) if g == 1 else (lambda: bn.g2_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n))
    apply(OLD)
    refm = call()
    sweep("G%d MSM 2^%d" % (g, logn), call, refm,
          [("round-3 tail", OLD), ("round-4 default", {}), ("windows cut into 2^12-bucket pieces", {"TAIL_BITS_W": 12}), ("... and chunks of 4", {"TAIL_BITS_W": 12, "MSM_CHUNK": 4}),
           ("2^11-bucket pieces, chunks of 4", {"TAIL_BITS_W": 11, "MSM_CHUNK": 4}), ("2^13-bucket pieces, chunks of 4", {"TAIL_BITS_W": 13, "MSM_CHUNK": 4}),
           ("chunks of 4, whole windows", {"MSM_CHUNK": 4})] + ([("default, G2 one lane per point", {"G2_TAIL_PAIR": 0})] if g == 2 else []))
