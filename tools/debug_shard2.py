import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
logd, world, rk = 20, 8, 3
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
sec, _ = circ.build_sections()
wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
k = bn.load_key(sections=sec, shard=(rk, world))
print("key loaded", k.table, flush=True)
for n, v in (("MSM_CHUNK", 8), ("TAIL_BITS", 15), ("TAIL_REDUCE", 0), ("G2_TAIL_PAIR", 0)):
    if os.environ.get("DBG_OLD") == "1": bn.lib.tune(n, v)
bn.lib.c.wsnark_timing_enable(1)
for it in range(2):
    rec = bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(rk, world), skip_h=(os.environ.get("DBG_SKIPH") == "1"))
    print("iteration", it, "done", flush=True)
