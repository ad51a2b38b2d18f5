"""Per-kernel times of the reduction tail, every kernel ALONE (one queue), over tuning combos given as "A=1,B=2;A=3".  2^20 proofs."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, int(os.environ.get("LOGD", "20")), n_public=5, seed=1)
sec, _ = circ.build_sections()
key = bn.load_key(sections=sec)
bn.lib.c.wsnark_pkey_wait_tables(key._h)
wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
r, s = bytes(range(32)), bytes(range(32, 64))
want = circ.expected_proof(r, s)
f = lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
combos = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv) for c in sys.argv[1].split(";")]
names = sorted({k for c in combos for k in c})
bn.lib.tune("PROVE_OVERLAP", 0)
for rd in range(2):
    for c in combos:
        for k in names: bn.lib.tune(k, c.get(k))
        for _ in range(3): out = f()
        n = 6
        bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): out = f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
        bn.lib.c.wsnark_timing_enable(0)
        kt = {k: round(v[0] / n, 4) for k, v in sorted(bn.lib.timing_report().items()) if v[1] and k.startswith("msm_") and "accumulate" not in k and "presort" not in k}
        print(json.dumps({"round": rd, "tuning": c, "ms_with_brackets": round(dt, 3), "ok": bool(out == want), "tail_ms_per_proof": kt, "tail_sum": round(sum(kt.values()), 4)}), flush=True)
