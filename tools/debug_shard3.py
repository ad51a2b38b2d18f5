import os, sys, hashlib, faulthandler
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
cfg = dict(x.split("=") for x in os.environ.get("DBG_TUNE", "").split(",") if x)
for n, v in cfg.items(): bn.lib.tune(n, int(v))
h = lambda b: hashlib.sha1(b).hexdigest()[:6]
out = []
for it in range(6):
    rec = bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(rk, world), skip_h=(os.environ.get("DBG_SKIPH") == "1"))
    out.append("A:%s B1:%s C:%s H:%s B2:%s" % (h(rec[0:96]), h(rec[96:192]), h(rec[192:288]), h(rec[288:384]), h(rec[384:576])))
print(os.environ.get("DBG_TUNE", "default"), "skip_h=" + os.environ.get("DBG_SKIPH", "0"))
for o in out: print("   ", o)
