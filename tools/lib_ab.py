#!/usr/bin/env python3
"""A/B of two BUILDS of the library on whole 2^20 proofs (one process per build; run the builds alternately so that box drift shows):
   python tools/lib_ab.py [path/to/other/libwsnark.so]
Prints ms per proof on two queues and on one (every kernel alone), and the per-kernel means of the one-queue proofs."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth, bn128, _lib

if len(sys.argv) > 1:
    class Other(_lib.Lib):          # (a tool-side subclass: the product's binding takes no path)
        SO = os.path.abspath(sys.argv[1])
    bn = bn128.Bn128(lib=Other())
else:
    bn = wasmsnark_amd.build(device=0)
r, s = bytes(range(32)), bytes(range(32, 64))
circ = synth.NativeCircuit(bn.lib, 20, n_public=5, seed=1)
sec, _ = circ.build_sections(); wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
key = bn.load_key(sections=sec)
want = circ.expected_proof(r, s)
f = lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
def run(n):
    for _ in range(3): out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 4), out
res = {"lib": os.path.basename(bn.lib.path), "two_queues_ms": [], "one_queue_ms": []}
ok = True
for rd in range(3):
    bn.lib.tune("PROVE_OVERLAP", None); ms, out = run(20); res["two_queues_ms"].append(ms); ok &= out == want
    bn.lib.tune("PROVE_OVERLAP", 0); ms, out = run(10); res["one_queue_ms"].append(ms); ok &= out == want
bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
for _ in range(5): f()
torch.cuda.synchronize(); bn.lib.c.wsnark_timing_enable(0)
bn.lib.tune("PROVE_OVERLAP", None)
res["alone_us_per_launch"] = {k: round(v[0] / v[1] * 1e3, 1) for k, v in sorted(bn.lib.timing_report().items()) if v[1]}
res["alone_ms_per_proof"] = {k: round(v[0] / 5, 3) for k, v in sorted(bn.lib.timing_report().items()) if v[1]}
res["ok"] = bool(ok)
print(json.dumps(res), flush=True)
