#!/usr/bin/env python3
"""Stress for the staging ring's slot events (csrc/context.hip: upload_pipelined): contexts are created and destroyed over and over
(groups of two on device 0), and the FIRST upload of every new context goes through a fresh ring -- slot events that have never been
recorded.  Before round 6's fix those events were queried / waited for, and ROCm 7.2 answered, now and then, "operation not permitted
on an event last recorded in a capturing stream" (one full GPU suite in three; internal.h: pin_ev_rec).
    python tools/ring_event_stress.py [rounds, default 40] [path/to/other/libwsnark.so]     -> one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import wasmsnark_amd
from wasmsnark_amd import synth, bn128, _lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if len(sys.argv) > 2:
    class Other(_lib.Lib):          # (a tool-side subclass: the product's binding takes no path)
        SO = os.path.abspath(sys.argv[2])
    bn = bn128.Bn128(lib=Other())
else:
    bn = wasmsnark_amd.build(device=0)
r, s = bytes(range(32)), bytes(range(32, 64))
circ = synth.NativeCircuit(bn.lib, 16, n_public=5, seed=3)
sec, _ = circ.build_sections(); wit = circ.witness_bin()
want = circ.expected_proof(r, s)
fails, errs, t0 = 0, {}, time.time()
for i in range(rounds):
    g = None
    try:
        g = bn128.Group(lib=bn.lib, devices=[0, 0])
        gk = g.load_key(sections=sec, wait_tables=False)          # host sections: every context stages its share through its own new ring
        ok = g.groth16GenProof(wit, gk, r=r, s=s) == want
        gk.free()
        if not ok:
            fails += 1; errs["wrong proof"] = errs.get("wrong proof", 0) + 1
    except Exception as e:  # noqa: BLE001
        fails += 1; k = str(e)[:160]; errs[k] = errs.get(k, 0) + 1
    finally:
        if g is not None:
            try:
                g.terminate()
            except Exception:  # noqa: BLE001
                pass
    # a shard load on the default context in between (its ring is old: recorded events)
    try:
        sh = bn.load_key(sections=sec, shard=(i % 4, 4), wait_tables=False); sh.free()
    except Exception as e:  # noqa: BLE001
        fails += 1; k = "default ctx: " + str(e)[:140]; errs[k] = errs.get(k, 0) + 1
print(json.dumps({"lib": os.path.basename(bn.lib.path), "rounds": rounds, "failures": fails, "errors": errs, "seconds": round(time.time() - t0, 1)}), flush=True)
