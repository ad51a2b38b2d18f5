#!/usr/bin/env python3
"""Writes the benchmark's 2^20 key and witness to a scratch directory, times the Node drop-in on them (tools/node_bench.js) and adds
the ctypes figures of the same calls from this process, so that the JS edge can be read against the C ABI it sits on.
    python tools/node_bench.py [log_domain] [reps]    -> one JSON line"""
import json
import os
import subprocess
import sys
import tempfile
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import wasmsnark_amd
from wasmsnark_amd import synth
logd = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
sec, _ = circ.build_sections()
pkey = synth.sections_to_pkey(sec)
wit = circ.witness_bin()
r, s = bytes(range(32)), bytes(range(32, 64))
want = circ.expected_proof(r, s)
d = tempfile.mkdtemp(prefix="wsnark_node_bench_")
kp, wp = os.path.join(d, "proving_key.bin"), os.path.join(d, "witness.bin")
open(kp, "wb").write(pkey)
open(wp, "wb").write(wit)
key = bn.load_key(pkey)


def t(f, n=reps):
    for _ in range(3):
        out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = f()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3), out


d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
res = {"log_domain": logd, "key_bytes": len(pkey)}
res["ctypes_resident_witness_ms"], p = t(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s))
res["ctypes_host_witness_ms"], p2 = t(lambda: bn.groth16GenProof(wit, key, r=r, s=s))
res["ctypes_proofs_match_closed_form"] = bool(p == want and p2 == want)
key.free()
bn.lib.shutdown()          # the Node process gets the GPU to itself
out = subprocess.run(["node", os.path.join(ROOT, "tools", "node_bench.js"), kp, wp, str(reps)], capture_output=True, text=True, timeout=900)
if os.environ.get("NODE_BENCH_STDERR"):
    sys.stderr.write(out.stderr[:6000])
line = [l for l in out.stdout.splitlines() if l.startswith("NODE_BENCH ")]
if not line:
    res["node_error"] = (out.stdout + out.stderr)[-1500:]
else:
    js = json.loads(line[0][len("NODE_BENCH "):])
    res["node"] = js
    res["node_proof_matches_closed_form"] = js.get("proof_pi_a0") == want["pi_a"][0]
    res["js_key_bytes_call_over_ctypes_host_witness_ms"] = round(js["key_bytes_call_ms"] - res["ctypes_host_witness_ms"], 3)
# the same first calls from a fresh PYTHON process (no Node, no digest; then with a thread hashing the key beside the load)
for label, extra in (("fresh_python_process", []), ("fresh_python_process_rocm_hip_runtime_like_node", ["--no-torch"])):
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cold_probe.py"), kp, wp] + extra, capture_output=True, text=True, timeout=600)
    if os.environ.get("NODE_BENCH_STDERR"):
        sys.stderr.write("---- " + label + "\n" + cp.stderr[:6000])
    ln = [l for l in cp.stdout.splitlines() if l.startswith("COLD_PROBE ")]
    res[label] = json.loads(ln[0][len("COLD_PROBE "):]) if ln else {"error": (cp.stdout + cp.stderr)[-800:]}
for f in (kp, wp):
    os.remove(f)
os.rmdir(d)
print(json.dumps(res))
