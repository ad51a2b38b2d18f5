"""WSNARK_TRACE=1 over a few warm 2^20 proofs: the host's marks (where the calling thread waits, what it does after the last kernel)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, 20, n_public=5, seed=1)
sec, _ = circ.build_sections(); wit = circ.witness_bin()
key = bn.load_key(sections=sec); bn.lib.c.wsnark_pkey_wait_tables(key._h)
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
r, s = bytes(range(32)), bytes(range(32, 64))
f = lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
for _ in range(5): f()
bn.lib.tune("TRACE", 1)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); print("== proof %d: %.3f ms (python call)" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
bn.lib.tune("TRACE", None)
