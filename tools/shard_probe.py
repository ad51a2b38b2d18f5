"""What ONE rank of an 8-way points-sharded proof costs, measured on the one GPU the build has: for a 2^logd key the eight shard
handles are loaded one after the other (wsnark_pkey_load_shard) and each one's partial record is timed (prove_partial: the rank's four
witness sums + its H sum over the contiguous hExps share, CALC_H computed in full -- the replicated form; and with WSNARK_PARTIAL_SKIP_H:
the four witness sums alone), next to the one-call proof on the whole key.  Not a scaling measurement (no exchange, no second GPU):
it shows the balance of the split and the compute a rank is left with.
    python tools/shard_probe.py 20 [world]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wasmsnark_amd
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
def timeit(f, n):
    for _ in range(3): out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
reps = 10 if logd <= 22 else 4
key = bn.load_key(sections=sec)
t_whole, p = timeit(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s), reps)
out = {"log_domain": logd, "world": world, "whole_key": {"prove_ms": round(t_whole, 3), "table_bytes": key.table["bytes"], "rows": key.table["rows_w"], "ok": p == want,
                                                          "key_load_ms": round(key.load_ms["total"], 1)}}
key.free()
ranks, recs = [], b""
for rank in range(world):
    k = bn.load_key(sections=sec, shard=(rank, world))
    t_all, rec = timeit(lambda: bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(rank, world)), reps)
    t_sums, _ = timeit(lambda: bn.groth16_prove_partial_dev(d_w.data_ptr(), len(wit), k, shard=(rank, world), skip_h=True), reps)
    ranks.append({"rank": rank, "pairs": k.shard["n_signals"], "rows": k.table["rows_w"], "table_bytes": k.table["bytes"], "key_load_ms": round(k.load_ms["total"], 1),
                  "partial_ms_with_full_calc_h": round(t_all, 3), "four_witness_sums_ms": round(t_sums, 3)})
    recs += rec
    if rank < world - 1:
        k.free()
out["combined_ok"] = bn.groth16_prove_finish(k, recs, r=r, s=s) == want
out["ranks"] = ranks
ts = [x["four_witness_sums_ms"] for x in ranks]
out["summary"] = {"max_rank_sums_ms": max(ts), "min_rank_sums_ms": min(ts), "max_rank_partial_ms": max(x["partial_ms_with_full_calc_h"] for x in ranks),
                  "whole_over_max_rank_partial": round(t_whole / max(x["partial_ms_with_full_calc_h"] for x in ranks), 2)}
print(json.dumps(out))
