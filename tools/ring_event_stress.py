#!/usr/bin/env python3
"""Stress around the staging ring's slot events (csrc/context.hip: upload_pipelined): groups of two contexts on device 0 are created, load a
key from host sections (the transposition of matrix B runs on a helper thread: prove.hip), prove and are destroyed, with loads on the
default context in between.  Before round 6's fix the helper thread staged matrix B through the DEFAULT context's ring on the GROUP
context's queue, which left default-ring events pointing at a queue that the group's end destroyed; ROCm 7.2's hipEventQuery reads the
queue through the event ("operation not permitted on an event last recorded in a capturing stream", one GPU suite in five).  Whether the
stale read trips depends on what the freed memory holds: this loop did NOT trip it in 240 rounds on the old library -- it is kept as the
closest thing to a regression run; the deterministic guard is upload_pipelined's queue_of_context check.
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
