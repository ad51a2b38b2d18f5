"""The distributed formulation with a world of one on the GPU (default 2^20): what it costs by itself -- batched four-step
transforms, pack / unpack, no exchange -- next to the one-call prover.  Three orchestrations of the same algorithm:
wsnark_groth16_prove_dist (native: one C call per proof, csrc/dist.hip), DistProver (Python between the kernels, round 2),
and the one-call prover on the whole key."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wasmsnark_amd
from wasmsnark_amd import dist as wd, synth
bn = wasmsnark_amd.build(device=0)
logd = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
sec, _ = circ.build_sections()
key = bn.load_key(sections=sec)
wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
npv = wd.NativeDistProver(bn, sec, device=dev)
dp = wd.DistProver(bn, key, bytes(sec["pointsH"]), device=dev)
r, s = bytes(range(32)), bytes(range(32, 64))
want = circ.expected_proof(r, s)
def timeit(f, n=20):
    for _ in range(5): out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
t1, p1 = timeit(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s))
t0, p0 = timeit(lambda: npv.prove(d_w.data_ptr(), len(wit), r=r, s=s))
t2, p2 = timeit(lambda: dp.prove(d_w.data_ptr(), len(wit), r=r, s=s))
t1b, _ = timeit(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s))
bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
for _ in range(2):
    npv.prove(d_w.data_ptr(), len(wit), r=r, s=s)
torch.cuda.synchronize(); bn.lib.c.wsnark_timing_enable(0)
kt = {k: round(v[0] / 2, 4) for k, v in sorted(bn.lib.timing_report().items())}
print(json.dumps({"log_domain": logd, "one_call_prove_ms": round(min(t1, t1b), 3), "native_dist_prover_world1_ms": round(t0, 3),
                  "python_dist_prover_world1_ms": round(t2, 3), "native_minus_one_call_ms": round(t0 - min(t1, t1b), 3),
                  "proofs_ok": p1 == want and p0 == want and p2 == want, "native_kernel_ms_per_proof": kt}))
