"""Kernel timeline (WSNARK_TIMELINE=1: HIP events per kernel bracket, both / all three queues on the device clock) of ONE
partial record of rank 0 of an 8-way points-sharded 2^logd key, witness sums only (WSNARK_PARTIAL_SKIP_H) and with CALC_H + H."""
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
k = bn.load_key(sections=sec, shard=(0, 8))
for skip in (True, False):
    for _ in range(3):
        bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(0, 8), skip_h=skip)
    bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
    bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(0, 8), skip_h=skip)
    torch.cuda.synchronize(); bn.lib.c.wsnark_timing_enable(0)
    sys.stderr.write("==== skip_h=%s\n" % skip); sys.stderr.flush()
    bn.lib.timing_report()
