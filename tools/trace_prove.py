import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
circ = synth.make_circuit(20, n_public=5, seed=1); S = synth.setup(circ, seed=2)
pkey, _ = synth.build_key(circ, S, bn.mul_base)
t0 = time.perf_counter(); key = bn.load_key(pkey); print('load_key s', time.perf_counter() - t0, file=sys.stderr); wit = synth.witness_bin(circ)
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
r32, s32 = bytes(range(32)), bytes(range(32, 64))
for i in range(3):
    t0 = time.perf_counter(); bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32); print("prove ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
if os.environ.get("WSNARK_TIMELINE") == "1":
    bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
    bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
    bn.lib.c.wsnark_timing_enable(0); bn.lib.timing_report()
