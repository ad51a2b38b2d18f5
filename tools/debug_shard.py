import hashlib, json, os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
logd = int(sys.argv[1]) if len(sys.argv) > 1 else 20
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
sec, _ = circ.build_sections()
wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
r, s = bytes(range(32)), bytes(range(32, 64))
want = circ.expected_proof(r, s)
print("setup done", flush=True)
NAMES = ("MSM_CHUNK", "TAIL_BITS", "TAIL_REDUCE", "G2_TAIL_PAIR", "PROVE_ORDER")
def apply(cfg):
    for n in NAMES: bn.lib.tune(n, cfg.get(n))
h = lambda b: hashlib.sha1(b).hexdigest()[:8]
def parts(rec): return [h(rec[0:96]), h(rec[96:192]), h(rec[192:288]), h(rec[384:576])]   # A B1 C B2
OLD = {"MSM_CHUNK": 8, "TAIL_BITS": 15, "TAIL_REDUCE": 0, "G2_TAIL_PAIR": 0}
keys = [bn.load_key(sections=sec, shard=(rk, world)) for rk in range(world)]
print("keys loaded", flush=True)
for name, cfg in (("old", OLD), ("old serial", dict(OLD, PROVE_ORDER=1)), ("default", {}), ("default serial", {"PROVE_ORDER": 1}), ("default nopair", {"G2_TAIL_PAIR": 0}),
                  ("default noreduce", {"TAIL_REDUCE": 0}), ("chunk8", {"MSM_CHUNK": 8}), ("bits15", {"TAIL_BITS": 15})):
    apply(cfg)
    runs = []
    for it in range(4):
        recs = b"".join(bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), keys[rk], shard=(rk, world), skip_h=False) for rk in range(world))
        ok = bn.groth16_prove_finish(keys[0], recs, r=r, s=s) == want
        runs.append((ok, parts(recs[3 * 576:4 * 576])))
    print(json.dumps({"cfg": name, "runs": runs}), flush=True)
apply({})
