#!/usr/bin/env python3
"""What the key's resident memory buys: the 2^20 proof on keys whose sections are resident as fixed-base tables
for ALL five sections (the default), for the hExps alone, for A / B1 / B2 / C alone, for none (plain sections), and with other table
window widths.  One process, every key loaded afresh, 2 x 20 proofs each after the table rows are built; ms per proof, resident GiB,
ms gained per GiB over the plain sections.    python tools/table_sweep.py [log_domain]"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
logd = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
sec, _ = circ.build_sections(); wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
r, s = bytes(range(32)), bytes(range(32, 64))
want = circ.expected_proof(r, s)
plain_gib = (circ.n_vars * 320 + circ.domain * 64) / 2.0**30
rows = []
cfgs = [("all five sections (default)", {}), ("none: plain sections", {"KEY_TABLE": 0}), ("hExps only", {"KEY_TABLE": 2}), ("A, B1, B2, C only", {"KEY_TABLE": 3}),
        ("all, window 19", {"TABLE_C": 19}), ("all, window 18", {"TABLE_C": 18}), ("all, window 21", {"TABLE_C": 21}), ("all five sections (default), again", {})]
for label, cfg in cfgs:
    for k, v in cfg.items(): bn.lib.tune(k, v)
    try:
        key = bn.load_key(sections=sec)
        f = lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
        best, ok = [], True
        for _ in range(2):
            for _ in range(3): out = f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): out = f()
            torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / 20 * 1e3)
            ok = ok and out == want
        t = key.table
        gib = (t["bytes"] if (t["rows_w"] > 1 or t["rows_h"] > 1) else 0) / 2.0**30
        # bytes reported are rows x sections for BOTH groups; a group kept plain has one row
        gib = (circ.n_vars * 320 * t["rows_w"] + circ.domain * 64 * t["rows_h"]) / 2.0**30
        rows.append({"key": label, "ms": [round(x, 3) for x in best], "resident_GiB": round(gib, 2), "rows_w_h": [t["rows_w"], t["rows_h"]],
                     "window_bits_w_h": [t["c_w"], t["c_h"]], "proofs_match_closed_form": bool(ok)})
        key.free()
    except Exception as ex:  # noqa: BLE001
        rows.append({"key": label, "error": repr(ex)[:200]})
    for k in cfg: bn.lib.tune(k, None)
    print(json.dumps(rows[-1]), flush=True)
base = [x for x in rows if x["key"].startswith("none") and "ms" in x]
if base:
    b = min(base[0]["ms"])
    print("# ms gained over the plain sections per GiB of tables beyond them (plain: %.2f GiB, %.3f ms):" % (plain_gib, b))
    for x in rows:
        if "ms" in x and x["resident_GiB"] > plain_gib + 0.01:
            print("#   %-40s %.3f ms  %5.2f GiB  -> %.3f ms / GiB" % (x["key"], min(x["ms"]), x["resident_GiB"], (b - min(x["ms"])) / (x["resident_GiB"] - plain_gib)))
