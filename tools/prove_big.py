"""Large single-GPU proofs through the section-based key loader (keys beyond proving_key.bin's 4 GiB):
    python tools/prove_big.py 22 24
builds a synthetic valid circuit of 2^L constraints, loads the key as sections, proves with the witness on
the device and checks the proof against the toxic-waste closed form.  Prints one JSON line per size.
    python tools/prove_big.py --node 24      (round 6) writes the key as a WSNARK64 FILE and proves from it through the Node drop-in
                                             (loadKey(path): the 7.8 GB key of 2^24 never exists as a JS buffer); WSNARK_BIG_DIR = where"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wasmsnark_amd
from wasmsnark_amd import synth

bn = wasmsnark_amd.build(device=0)
NODE = "--node" in sys.argv
if NODE:
    import subprocess, tempfile
    from wasmsnark_amd import formats
    for L in [int(a) for a in sys.argv[1:] if a.isdigit()] or [22]:
        circ = synth.NativeCircuit(bn.lib, L, n_public=5, seed=1)
        sec, _ = circ.build_sections()
        d = tempfile.mkdtemp(prefix="wsnark-big-", dir=os.environ.get("WSNARK_BIG_DIR"))
        kp, wp, jp = os.path.join(d, "key.wsnark64"), os.path.join(d, "witness.bin"), os.path.join(d, "want.json")
        t0 = time.perf_counter()
        nbytes = formats.write_key_container(sec, kp)
        t1 = time.perf_counter()
        del sec
        open(wp, "wb").write(circ.witness_bin())
        r32, s32 = bytes(range(32)), bytes(range(32, 64))
        json.dump(circ.expected_proof(r32, s32), open(jp, "w"))
        out = subprocess.run(["node", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "node_key_file_check.js"), kp, wp, jp, r32.hex(), s32.hex()]
                             + ([os.environ["NODE_BENCH_DEVICES"]] if os.environ.get("NODE_BENCH_DEVICES") else []), capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "{}"
        print(json.dumps({"log_domain": L, "key_file_bytes": nbytes, "write_s": round(t1 - t0, 1), "node": json.loads(line) if line.startswith("{") else line,
                          "rc": out.returncode, "stderr": out.stderr[-400:]}), flush=True)
        for f in (kp, wp, jp):
            os.remove(f)
        os.rmdir(d)
        circ.free()
    sys.exit(0)
for L in [int(a) for a in sys.argv[1:]] or [22]:
    t0 = time.perf_counter()
    circ = synth.make_circuit(L, n_public=5, seed=1)
    S = synth.setup(circ, seed=2)
    t1 = time.perf_counter()
    sec, _ = synth.build_sections(circ, S, bn.mul_base)
    key_bytes = sum(len(v) for v in sec.values() if isinstance(v, (bytes, bytearray)))
    t2 = time.perf_counter()
    key = bn.load_key(sections=sec)
    t3 = time.perf_counter()
    del sec
    wit = synth.witness_bin(circ)
    d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    r32, s32 = bytes(range(32)), bytes(range(32, 64))
    proof = bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
    ok = proof == synth.expected_proof(circ, S, r32, s32, bn.mul_base)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r32, s=s32)
        ts.append((time.perf_counter() - t) * 1e3)
    free_b, total_b = torch.cuda.mem_get_info()
    print(json.dumps({"log_domain": L, "n_vars": circ.n_vars, "key_bytes": key_bytes, "prove_ms": [round(x, 2) for x in ts],
                      "matches_toxic_waste_closed_form": bool(ok), "circuit_setup_s": round(t1 - t0, 1),
                      "key_points_s": round(t2 - t1, 1), "load_key_s": round(t3 - t2, 2),
                      "hbm_used_GB": round((total_b - free_b) / 1e9, 1)}), flush=True)
    del key, d_w, circ, S
