"""Large single-GPU proofs through the section-based key loader (keys beyond proving_key.bin's 4 GiB):
    python tools/prove_big.py 22 24
builds a synthetic valid circuit of 2^L constraints, loads the key as sections, proves with the witness on
the device and checks the proof against the toxic-waste closed form.  Prints one JSON line per size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wasmsnark_amd
from wasmsnark_amd import synth

bn = wasmsnark_amd.build(device=0)
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
