"""DistProver with a world of one on the GPU at 2^20: what the distributed formulation costs by itself (batched four-step
transforms + layout permutes + Python orchestration, no exchange) next to the one-call prover."""
import os, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wasmsnark_amd
from wasmsnark_amd import dist as wd, synth
bn = wasmsnark_amd.build(device=0)
logd = int(sys.argv[1]) if len(sys.argv) > 1 else 20
circ = synth.make_circuit(logd, n_public=5, seed=1); S = synth.setup(circ, seed=2)
pkey, _ = synth.build_key(circ, S, bn.mul_base)
key = bn.load_key(pkey)
wit = synth.witness_bin(circ)
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
dp = wd.DistProver(bn, key, pkey[struct.unpack_from("<I", pkey, 36)[0]:], device=torch.device("cuda", 0))
r, s = bytes(range(32)), bytes(range(32, 64))
want = synth.expected_proof(circ, S, r, s, bn.mul_base)
def timeit(f, n=10):
    for _ in range(3): out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
t1, p1 = timeit(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s))
t2, p2 = timeit(lambda: dp.prove(d_w.data_ptr(), len(wit), r=r, s=s))
t3, _ = timeit(lambda: dp._calc_h_local(d_w.data_ptr(), len(wit)))
t4, _ = timeit(lambda: bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), key, shard=(0, 1), skip_h=True))
print({"log_domain": logd, "one_call_prove_ms": round(t1, 3), "dist_prover_world1_ms": round(t2, 3), "calc_h_distributed_form_alone_ms": round(t3, 3),
       "four_sums_skip_h_alone_ms": round(t4, 3), "proofs_ok": p1 == want and p2 == want})
