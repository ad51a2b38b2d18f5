#!/usr/bin/env python3
"""Generic one-process A/B over library switches on three workloads (prove 2^20, prove 2^16, a rank's share of an 8-way 2^20 key).
   python tools/quick_ab.py NAME=v1,v2,... [NAME2=...]     e.g. TREE_SLOTS=0,256,128"""
import itertools, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
axes = [(a.split("=")[0], [int(x) for x in a.split("=")[1].split(",")]) for a in sys.argv[1:]]
bn = wasmsnark_amd.build(device=0)
r, s = bytes(range(32)), bytes(range(32, 64))
def t(f, n=20):
    for _ in range(3): out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 4), out
def sweep(name, f, want):
    for combo in list(itertools.product(*[v for _, v in axes])) * 2:
        for (k, _), v in zip(axes, combo): bn.lib.tune(k, v if v else None)
        ms, out = t(f)
        print(json.dumps({"workload": name, "tuning": dict(zip([k for k, _ in axes], combo)), "ms": ms, "ok": bool(want is None or out == want)}), flush=True)
    for k, _ in axes: bn.lib.tune(k, None)
circ = synth.NativeCircuit(bn.lib, 20, n_public=5, seed=1)
sec, _ = circ.build_sections(); wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
k = bn.load_key(sections=sec, shard=(3, 8))
f = lambda: bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(3, 8), skip_h=True)
sweep("rank 3 of 8 sums", f, f()); k.free()
key = bn.load_key(sections=sec)
sweep("prove 2^20", lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s), circ.expected_proof(r, s)); key.free()
c16 = synth.NativeCircuit(bn.lib, 16, n_public=5, seed=1); s16, _ = c16.build_sections(); w16 = c16.witness_bin()
d16 = torch.frombuffer(bytearray(w16), dtype=torch.uint8).cuda(); k16 = bn.load_key(sections=s16)
sweep("prove 2^16", lambda: bn.groth16GenProof_dev(d16.data_ptr(), len(w16), k16, r=r, s=s), c16.expected_proof(r, s))
