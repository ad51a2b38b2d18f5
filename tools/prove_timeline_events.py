"""Kernel timeline of ONE proof on the whole key (WSNARK_TIMELINE=1: HIP events per kernel bracket on the queue it ran on, common device clock)."""
import os, sys
os.environ["WSNARK_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
logd = int(sys.argv[1]) if len(sys.argv) > 1 else 20
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
sec, _ = circ.build_sections(); wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
k = bn.load_key(sections=sec)
r, s = bytes(range(32)), bytes(range(32, 64))
for _ in range(5):
    bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), k, r=r, s=s)
bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), k, r=r, s=s)
torch.cuda.synchronize(); bn.lib.c.wsnark_timing_enable(0)
bn.lib.timing_report()
