#!/usr/bin/env python3
"""A fresh process's first calls on a key read from a file, one by one (what tools/node_bench.js sees, without Node): load the
key without waiting for the table rows, then six proofs from a host witness, each timed; optionally with a thread hashing the
key bytes beside the load (the JS edge takes a whole-buffer digest off the event loop at load time).
    python tools/cold_probe.py <proving_key.bin> <witness.bin> [--hash] [--wait] [--no-torch]    -> one JSON line"""
import hashlib
import json
import os
import sys
import threading
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
t_start = time.perf_counter()
if "--no-torch" in sys.argv:
    # the library then resolves libamdhip64 to /opt/rocm's (what the Node addon gets) instead of the one PyTorch bundles and
    # loads first (wasmsnark_amd/_lib.py): the two hosts otherwise run DIFFERENT HIP runtimes
    sys.modules["torch"] = None
import wasmsnark_amd
r, s = bytes(range(32)), bytes(range(32, 64))
ms = lambda t0: round((time.perf_counter() - t0) * 1e3, 2)
t0 = time.perf_counter()
bn = wasmsnark_amd.build(device=0)        # (first the library, then the inputs: wsnark_init's helper threads work while the files are read)
t_init = ms(t0)
pkey = open(sys.argv[1], "rb").read()
wit = open(sys.argv[2], "rb").read()
out = {"hip_runtime": "/opt/rocm (no PyTorch in the process)" if "--no-torch" in sys.argv else "PyTorch's bundled libamdhip64", "hash_beside_load": "--hash" in sys.argv, "wait_tables": "--wait" in sys.argv, "init_ms": t_init}
th = None
if "--hash" in sys.argv:
    th = threading.Thread(target=lambda: out.__setitem__("hash_ms", (lambda t: (hashlib.blake2b(pkey).digest(), ms(t))[1])(time.perf_counter())))
    th.start()
t0 = time.perf_counter()
key = bn.load_key(pkey, wait_tables=("--wait" in sys.argv))
out["load_ms"] = ms(t0)
calls = []
proofs = []
for i in range(6):
    t0 = time.perf_counter()
    proofs.append(bn.groth16GenProof(wit, key, r=r, s=s))
    calls.append(ms(t0))
out["calls_ms"] = calls
out["all_equal"] = all(p == proofs[0] for p in proofs)
if th:
    th.join()
key.wait_tables()
out["load_ms_by_phase"] = {k: round(v, 2) for k, v in key.load_ms.items()}
t0 = time.perf_counter()
for i in range(10):
    bn.groth16GenProof(wit, key, r=r, s=s)
out["steady_ms"] = round(ms(t0) / 10, 3)
out["pi_a0"] = proofs[0]["pi_a"][0]
print("COLD_PROBE " + json.dumps(out))
