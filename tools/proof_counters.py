#!/usr/bin/env python3
"""P proofs of the benchmark's own 2^20 circuit between two marker launches, for rocprofv3 --pmc passes (tools/gpu_session.sh,
pass "issue"): everything the profiler sees between the markers belongs to exactly P warm proofs from a resident witness -- no key
generation, no table build, no bench scaffolding.  The marker is wsnark_peak_probe(5) (kernel probe_inverse_kernel: nothing else
launches it).  tools/pmc_proof_budget.py turns the pass into the per-proof instruction budget behind bench.py's `roofline_proof`.
    python tools/proof_counters.py [log_domain] [proofs]"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import wasmsnark_amd
from wasmsnark_amd import synth
logd = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1, style="columns")
sec, _ = circ.build_sections()
key = bn.load_key(sections=sec)
wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
r, s = bytes(range(32)), bytes(range(32, 64))
want = circ.expected_proof(r, s)
for _ in range(3):
    ok = bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s) == want
torch.cuda.synchronize()
v = C.c_double(0)
bn.lib.check(bn.lib.c.wsnark_peak_probe(5, C.byref(v)))      # ---- marker
torch.cuda.synchronize()
for _ in range(P):
    ok = ok and bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s) == want
torch.cuda.synchronize()
bn.lib.check(bn.lib.c.wsnark_peak_probe(5, C.byref(v)))      # ---- marker
torch.cuda.synchronize()
print(json.dumps({"log_domain": logd, "proofs_between_markers": P, "proofs_match_closed_form": bool(ok), "n_vars": circ.n_vars,
                  "nnz": int(circ.nnz), "marker_kernel": "probe_inverse_kernel"}))
