#!/usr/bin/env python3
"""Round 5: wavefront priorities x queue orders on whole 2^20 proofs, A/B in ONE process (wsnark_tuning_set).
   python tools/sched_ab.py [rounds] ["NAME=v,NAME2=v;NAME=v;..."]   (a combo list; default: the priority / order matrix)
Each combo: 3 warm-up + 20 timed proofs, resident witness, proofs checked against the closed form; the list is run `rounds`
times in alternation so that box drift shows."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 2:
    combos = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv) for c in sys.argv[2].split(";")]
else:
    combos = [
        {"TAIL_PRIO": 0, "PLAN_PRIO": 0, "PROVE_ORDER": 1},      # round 4
        {"TAIL_PRIO": 3, "PLAN_PRIO": 0, "PROVE_ORDER": 1},
        {"TAIL_PRIO": 3, "PLAN_PRIO": 3, "PROVE_ORDER": 1},
        {"TAIL_PRIO": 0, "PLAN_PRIO": 0, "PROVE_ORDER": 4},
        {"TAIL_PRIO": 3, "PLAN_PRIO": 3, "PROVE_ORDER": 4},
        {"TAIL_PRIO": 0, "PLAN_PRIO": 0, "PROVE_ORDER": 5},
        {"TAIL_PRIO": 3, "PLAN_PRIO": 3, "PROVE_ORDER": 5},
        {"TAIL_PRIO": 3, "PLAN_PRIO": 3, "PROVE_ORDER": 2},
        {"TAIL_PRIO": 3, "PLAN_PRIO": 3, "PROVE_ORDER": 3},
    ]
logd = int(os.environ.get("LOGD", "20"))
bn = wasmsnark_amd.build(device=0)
r, s = bytes(range(32)), bytes(range(32, 64))
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
sec, _ = circ.build_sections(); wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
key = bn.load_key(sections=sec)
want = circ.expected_proof(r, s)
f = lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
names = sorted({k for c in combos for k in c})
def run(n):
    for _ in range(3): out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 4), out
for rd in range(rounds):
    for c in combos:
        for k in names: bn.lib.tune(k, c.get(k))
        ms, out = run(20)
        print(json.dumps({"round": rd, "tuning": c, "ms": ms, "ok": bool(out == want)}), flush=True)
for k in names: bn.lib.tune(k, None)
