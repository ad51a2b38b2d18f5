import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, 20, n_public=5, seed=1)
sec, _ = circ.build_sections(); key = bn.load_key(sections=sec); wit = circ.witness_bin()
print("key load", key.load_ms)
r, s = bytes(range(32)), bytes(range(32, 64))
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
for _ in range(5): bn.groth16GenProof(wit, key, r=r, s=s)
def t(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("dev", round(t(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)), 3), "host", round(t(lambda: bn.groth16GenProof(wit, key, r=r, s=s)), 3))
print("dev", round(t(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)), 3), "host", round(t(lambda: bn.groth16GenProof(wit, key, r=r, s=s)), 3))
os.environ["WSNARK_TRACE"] = "1"
